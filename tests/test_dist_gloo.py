"""CPU, world_size 2 over gloo: the N>1 path of bench.py / sequence-sharded tracking.

Two processes shard four synthetic sequences round-robin (sequence i -> rank i % 2, as the reference's
engine.py:289-303), track them with the real Tracker + model modules on CPU (C oracle as the operator,
test only), and merge the results with the object all_gather; the merged result must equal a
single-process run.  Also checks barrier + max-over-ranks timing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_sequences():
    seqs = []
    for s in range(4):
        g = torch.Generator().manual_seed(100 + s)
        base = torch.randn(1, 3, 96, 128, generator=g)
        seq = []
        for _ in range(2):
            base = base + 0.1 * torch.randn(1, 3, 96, 128, generator=g)
            seq.append({'img': base.clone(), 'orig_size': torch.tensor([[192, 256]]),
                        'size': torch.tensor([[96, 128]]), 'dets': torch.zeros(1, 0, 4)})
        seqs.append(seq)
    return seqs


def _make_tracker(device):
    from oracle import msda_oracle
    from tests import util_models as um
    from trackformer_amd import config, factory, msda
    from trackformer_amd.tracker import Tracker
    msda.MSDeformAttnFunction = msda_oracle.make_torch_function()  # CPU checker, tests only
    overlays = ("deformable", "tracking", "mot17")
    args = config.make_args(*overlays, device="cpu", num_queries=40, enc_layers=1, dec_layers=2)
    torch.manual_seed(42)
    model, _, post = factory.build_model(args)
    from tests.util_weights import perturb_state_dict
    perturb_state_dict(model, 3)
    model.tracking()
    return Tracker(model, post, config.tracker_cfg(), False)


def _flatten(results):
    rows = []
    for seq in sorted(results):
        for tid in sorted(results[seq]):
            for f in sorted(results[seq][tid]):
                r = results[seq][tid][f]
                rows.append([seq, tid, f, *r['bbox'].tolist(), float(r['score'])])
    return np.array(rows)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from trackformer_amd import dist_utils as du
    before = sorted(os.sched_getaffinity(0))
    share = du.pin_rank_to_cpus(rank, world, max_threads=2)   # what bench.py does per rank at N > 1
    if len(before) >= world:
        assert share == before[len(before) * rank // world:len(before) * (rank + 1) // world]
        assert sorted(os.sched_getaffinity(0)) == share and torch.get_num_threads() <= 2
    r, lr, w = du.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and du.is_distributed()
    seen = du.ranks_seen()
    assert [e["rank"] for e in seen] == list(range(world)) and len({e["pid"] for e in seen}) == world
    assert all(e["backend"] == "gloo" for e in seen)
    assert du.shard_sequences(list(range(5))) == ([0, 2, 4] if rank == 0 else [1, 3])
    du.barrier()
    slow = du.max_over_ranks(1.0 + rank)          # rank 1 is "slower"
    total = du.sum_over_ranks(10.0)
    merged = du.track_sequences(_make_tracker, _make_sequences(), "cpu", interleave=2)   # two sequences in flight per rank
    du.barrier()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), _flatten(merged))
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%r %r %d" % (slow, total, len(merged)))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sequence_sharding_matches_single_process(tmp_path, monkeypatch):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from trackformer_amd import dist_utils as du
    from trackformer_amd import msda
    monkeypatch.setattr(msda, "MSDeformAttnFunction", msda.MSDeformAttnFunction)  # restored at teardown
    single = _flatten(du.track_sequences(_make_tracker, _make_sequences(), "cpu"))
    for rank in range(2):
        slow, total, n = open(tmp_path / ("rank%d.txt" % rank)).read().split()
        assert float(slow) == 2.0 and float(total) == 20.0 and int(n) == 4
        got = np.load(tmp_path / ("rank%d.npy" % rank))
        assert got.shape == single.shape and got.shape[0] > 0
        np.testing.assert_array_equal(got[:, :3], single[:, :3])       # sequence, track id, frame
        np.testing.assert_allclose(got[:, 3:], single[:, 3:], atol=1e-4)


def _train_worker(rank, world, port, out_dir):
    """cfg-3 path at world size 2: DDP over gloo, each rank its own batch, one engine.train_step."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle import msda_oracle
    from tests import util_models as um
    from trackformer_amd import config, dist_utils as du, engine, factory, msda
    msda.MSDeformAttnFunction = msda_oracle.make_torch_function()  # CPU checker, tests only
    du.init_from_env(backend="gloo")
    model, criterion, args = um.build_train(factory.build_model, config.make_args)   # same seed: same weights
    optimizer, _ = engine.build_optimizer(model, args)
    ddp = engine.wrap_ddp(model, torch.device("cpu"))
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    samples, targets = um.train_batch(seed=9 + rank)        # different data per rank
    ddp.train()
    criterion.train()
    torch.manual_seed(7)
    loss, _ = engine.train_step(ddp, criterion, optimizer, samples, targets,
                                clip_max_norm=args.clip_max_norm)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    np.save(os.path.join(out_dir, "train%d.npy" % rank),
            np.array([float(loss), float(flat.double().sum()), float(flat.double().abs().sum())]))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_ddp_training_step_keeps_replicas_identical(tmp_path):
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(tmp_path / ("train%d.npy" % r)) for r in range(2))
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert a[0] != b[0]                       # different batches -> different local losses
    np.testing.assert_array_equal(a[1:], b[1:])   # averaged gradients -> identical weights after the step


def test_single_process_helpers_are_noops():
    from trackformer_amd import dist_utils as du
    assert not du.is_distributed()
    du.barrier()
    assert du.max_over_ranks(3.5) == 3.5 and du.sum_over_ranks(2.0) == 2.0
    assert du.shard_sequences(list("abcd"), rank=1, world=2) == ["b", "d"]
    assert du.gather_results({"a": 1}) == [{"a": 1}]
