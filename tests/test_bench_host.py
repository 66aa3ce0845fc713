"""CPU (-m "not gpu"): the host-side pieces of bench.py that must not break between GPU runs -- its arguments, the table of split
products it reports (`dtype`, side legs), and the committed counter / traffic passes it folds into the line (profiles/)."""
import glob
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_arguments_and_defaults(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.config == "cfg2" and a.split_terms is None
    for t in ("6", "16"):
        monkeypatch.setattr(sys, "argv", ["bench.py", "--split-terms", t, "--config", "cfg4", "--no-split3"])
        a = bench.parse_args()
        assert a.split_terms == int(t) and a.config == "cfg4" and a.no_split3
    for t in ("5", "3"):   # 3: the three-term bf16 mode was removed in round 5
        monkeypatch.setattr(sys, "argv", ["bench.py", "--split-terms", t])
        with pytest.raises(SystemExit):
            bench.parse_args()


def test_every_split_product_has_a_name_a_leg_and_a_dtype_text():
    from trackformer_amd import fused
    assert set(bench._ARITH) == {6, 16} and fused.split_terms() in bench._ARITH
    legs = [v[1] for v in bench._ARITH.values()]
    assert len(set(legs)) == 2 and all(leg.endswith("_fps") for leg in legs)
    assert "fp16" in bench._ARITH[16][2] and "six-term" in bench._ARITH[6][2]
    with pytest.raises(ValueError, match="removed in round 5"):
        fused.set_split_terms(3)
    prev = fused.set_split_terms(6)
    try:
        assert fused.split_terms() == 6
    finally:
        fused.set_split_terms(prev)


def test_committed_counter_passes_are_readable():
    """bench.py folds the newest profiles/rNN_mfma_utilisation.json and rNN_msda_fwd_pquad_traffic.json into its line."""
    m = bench.committed_mfma_utilisation()
    assert m is not None and m["terms"] in (3, 6, 16) and "source_commit" in m   # (3: a counter pass committed before round 4)
    assert m["harness"], m            # the harness kernels were found under their current names
    assert all(0.0 <= v <= 1.0 for v in m["harness"].values())
    frame = m["per_kernel_in_an_eager_cfg2_frame"]
    assert frame and max(frame.values()) <= 1.0 and any("ffn_fused" in k for k in frame)
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_msda_fwd_pquad_traffic.json")))
    assert files
    with open(files[-1]) as f:
        t = json.load(f)
    pq = t.get("msda_fwd_f32_pquad2") or t["msda_fwd_f32_pquad"]   # round 5: the second version of the kernel
    algorithmic = 4 * (22223 * 8 * 32 + 3 * 22223 * 8 * 4 * 4 + 22223 * 8 * 32)      # SURVEY 8(d) at the cfg-2 encoder call
    assert algorithmic == 79647232 and algorithmic <= pq["hbm_traffic_bytes_per_launch"] < 2 * algorithmic


def test_every_rank_gets_its_own_library_caches(monkeypatch, tmp_path):
    """bench.per_rank_caches (round 5): MIOpen's user database / kernel cache and the TunableOp results file are per rank -- eight
    ranks of a first run writing one find-db serialise on its lock -- and a path the caller already set is left alone."""
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    for var in ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR", "PYTORCH_TUNABLEOP_FILENAME"):
        monkeypatch.delenv(var, raising=False)
    base3 = bench.per_rank_caches(3)
    got3 = {v: os.environ[v] for v in ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR", "PYTORCH_TUNABLEOP_FILENAME")}
    assert base3.endswith("tf_bench_rank3") and all(p.startswith(base3) for p in got3.values())
    assert os.path.isdir(got3["MIOPEN_USER_DB_PATH"]) and os.path.isdir(got3["MIOPEN_CUSTOM_CACHE_DIR"])
    for var in got3:
        monkeypatch.delenv(var)
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    base5 = bench.per_rank_caches(5)
    assert os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"          # the caller's choice stands
    assert os.environ["MIOPEN_CUSTOM_CACHE_DIR"].startswith(base5) and base5 != base3
