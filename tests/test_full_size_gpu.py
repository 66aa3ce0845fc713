"""GPU (-m gpu): the BASELINE-size configurations end to end against the reference CPU path.

cfg 2 (800x1333, 300 object + 100 track queries) and cfg 4 (hidden 288, 500 + 300 queries, 8 decoder
levels) through deformable_detr.py:124-275, and a 3-frame 800x1333 sequence through
tracker.py:266-550, compared with goldens produced from the reference's own classes on CPU
(tests/golden/make_golden_full.py).  The model runs in the set-ups `bench.py` times: eager with the
library defaults, the tuned runtime (`runtime.configure_inference`: MIOpen find mode + shipped
TunableOp selections) behind `GraphedDetector` (HIP-graph replay, fused MSDeformAttn entry -> the
LDS-window encoder kernel at S = 22 223), and the same with the bf16 split-product linears.

Tolerances are north_star's: boxes / logits within 1e-3 (fp32), track ids bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from tests import util_models as um

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from trackformer_amd import _cabi
    _cabi.lib()
    return torch.device("cuda:0")


def _checksum(model):
    return float(sum(float(v.double().abs().sum()) for v in model.state_dict().values()))


@pytest.fixture(scope="module")
def models(dev):
    """One model per full-size case, shared by the set-ups below (building + perturbing takes seconds)."""
    from trackformer_amd import config, factory
    cache = {}

    def get(case):
        if case not in cache:
            model, post, args = um.build(case, factory.build_model, config.make_args, device=dev)
            model.to(dev).tracking()
            cache[case] = (model, post, args)
        return cache[case]
    return get


SETUPS = ["eager", "graph_tuned", "graph_split_linear", "graph_split6"]
# set-up -> split product (include/tf_fused.h); "graph_split_linear" is the bench set-up: the package's default product (fp16 pieces)
_SPLIT_SETUPS = {"graph_split_linear": 16, "graph_split6": 6}


def _forward(case, models, dev, setup):
    from trackformer_amd import fused, runtime
    from trackformer_amd.graphed import GraphedDetector
    model, post, args = models(case)
    img, prev, target = um.model_inputs(case, args.hidden_dim)
    img, prev, target = img.to(dev), prev.to(dev), um.to_device(target, dev)
    detector = model
    prev_split = fused.set_split_linear(setup in _SPLIT_SETUPS)
    prev_terms = fused.set_split_terms(_SPLIT_SETUPS.get(setup, 6))
    try:
        if setup != "eager":
            runtime.configure_inference(verbose=False)
            detector = GraphedDetector(model)
        with torch.no_grad():
            reps = 1 if setup == "eager" else 3     # call 1 eager, call 2 captures + replays, call 3 replays
            for _ in range(reps):
                prev_features = None
                if args.multi_frame_attention:
                    _, _, prev_features, _, _ = detector(prev, None, None)
                out, _, feats, memory, hs = detector(img, [dict(t) for t in target], prev_features)
            res = post['bbox'](out, torch.tensor([list(um.FULL_ORIG)], device=dev))[0]
        if setup != "eager" and not args.multi_frame_attention:
            assert len(detector._graphs) == 1, "the HIP-graph path was not taken"
    finally:
        fused.set_split_linear(prev_split)
        fused.set_split_terms(prev_terms)
    return model, out, res, feats, memory


def _compare(case, model, out, res, feats, memory, box_tol=1e-3, logit_tol=1e-3, orig=None):
    orig = orig or um.FULL_ORIG
    z = np.load(os.path.join(GOLDEN, "full_%s.npz" % case))
    assert abs(_checksum(model) - float(z["weight_checksum"])) < 1e-6 * float(z["weight_checksum"])
    np.testing.assert_allclose(out['pred_boxes'].cpu().numpy(), z['pred_boxes'], atol=box_tol)
    np.testing.assert_allclose(out['pred_logits'].cpu().numpy(), z['pred_logits'], atol=logit_tol)
    np.testing.assert_allclose(out['hs_embed'].cpu().numpy(), z['hs_embed'], atol=5 * logit_tol)
    aux_b = np.stack([a['pred_boxes'].cpu().numpy() for a in out['aux_outputs']])
    aux_l = np.stack([a['pred_logits'].cpu().numpy() for a in out['aux_outputs']])
    np.testing.assert_allclose(aux_b, z['aux_boxes'], atol=box_tol)
    np.testing.assert_allclose(aux_l, z['aux_logits'], atol=logit_tol)
    np.testing.assert_array_equal(res['labels'].cpu().numpy(), z['labels'])
    np.testing.assert_allclose(res['scores'].cpu().numpy(), z['scores'], atol=logit_tol)
    np.testing.assert_allclose(res['boxes'].cpu().numpy(), z['boxes'], atol=box_tol * max(orig))
    # encoder output (6 layers of the LDS-window kernel at S = 22 223): every 89th token
    mem = (torch.cat([m.flatten(2) for m in memory], 2) if isinstance(memory, (list, tuple)) else memory.flatten(2)).transpose(1, 2)
    assert list(mem.shape) == z['memory_shape'].tolist()
    rows = mem[0, ::um.FULL_MEMORY_ROW_STRIDE].cpu().numpy()
    np.testing.assert_allclose(rows, z['memory_rows'], atol=logit_tol * max(1.0, np.abs(z['memory_rows']).max()))
    f = feats[-1].tensors[0, ::um.FULL_FEAT_CH_STRIDE].cpu().numpy()
    np.testing.assert_allclose(f, z['feat_last'], atol=1e-3 * max(1.0, np.abs(z['feat_last']).max()))
    return float(np.abs(out['pred_boxes'].cpu().numpy() - z['pred_boxes']).max()), \
        float(np.abs(out['pred_logits'].cpu().numpy() - z['pred_logits']).max())


@pytest.mark.parametrize("setup", SETUPS)
@pytest.mark.parametrize("case", um.FULL_DETECTOR_CASES)
def test_full_size_model_matches_reference_cpu_path(dev, models, case, setup):
    model, out, res, feats, memory = _forward(case, models, dev, setup)
    dbox, dlogit = _compare(case, model, out, res, feats, memory)
    print("%s / %s: max |d boxes| %.2e, max |d logits| %.2e" % (case, setup, dbox, dlogit))


def _run_tracker(models, dev, setup):
    from trackformer_amd import config, fused, runtime
    from trackformer_amd.graphed import GraphedDetector
    from trackformer_amd.tracker import Tracker
    model, post, args = models("cfg2_full")
    detector = model
    prev_split = fused.set_split_linear(setup in _SPLIT_SETUPS)
    prev_terms = fused.set_split_terms(_SPLIT_SETUPS.get(setup, 6))
    try:
        if setup != "eager":
            runtime.configure_inference(verbose=False)
            detector = GraphedDetector(model)
        tracker = Tracker(detector, post, config.tracker_cfg(), False)
        tracker.reset()
        active = []
        with torch.no_grad():
            for blob in um.full_tracker_sequence():
                tracker.step(dict(blob, img=blob['img'].to(dev)))
                active.append(len(tracker.tracks))
    finally:
        fused.set_split_linear(prev_split)
        fused.set_split_terms(prev_terms)
    results = tracker.get_results()
    rows = np.array([[tid, f, *results[tid][f]['bbox'].tolist(), float(results[tid][f]['score']),
                      results[tid][f]['obj_ind']]
                     for tid in sorted(results) for f in sorted(results[tid])], dtype=np.float64)
    return tracker, rows, active


@pytest.mark.parametrize("setup", SETUPS)
def test_full_size_tracker_track_ids_bit_exact(dev, models, setup):
    tracker, rows, active = _run_tracker(models, dev, setup)
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg2.npz"))
    assert int(z["num_tracks"]) == tracker.track_num
    assert int(z["num_reids"]) == tracker.num_reids
    assert z["active_per_frame"].tolist() == active
    assert rows.shape == z["rows"].shape
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])   # id, frame, source query
    np.testing.assert_allclose(rows[:, 2:6], z["rows"][:, 2:6], atol=1e-3 * max(um.FULL_ORIG))
    np.testing.assert_allclose(rows[:, 6], z["rows"][:, 6], atol=1e-3)


def fused_default_terms():
    from trackformer_amd import fused
    return fused.split_terms()


def _ids_until_divergence(models, dev, gold, n_frames, split, terms):
    """Tracker.step over the 64-frame sequence in the bench set-up (tuned runtime, HIP graphs with the bucketed track-query
    count) under one arithmetic set-up; -> (frames that agree with the reference from the start, the tracker)."""
    from trackformer_amd import config, fused, runtime
    from trackformer_amd.graphed import GraphedDetector
    from trackformer_amd.tracker import Tracker
    model, post, args = models("cfg2_full")
    runtime.configure_inference(verbose=False)
    prev_split, prev_terms = fused.set_split_linear(split), fused.set_split_terms(terms)
    try:
        tracker = Tracker(GraphedDetector(model), post, config.tracker_cfg(), False)
        tracker.reset()
        agree = 0
        with torch.no_grad():
            for f, blob in enumerate(um.full_tracker_sequence(n_frames=n_frames)):
                tracker.step(dict(blob, img=blob['img'].to(dev)))
                if sorted(t.id for t in tracker.tracks) != gold[f]:
                    break
                agree = f + 1
    finally:
        fused.set_split_linear(prev_split)
        fused.set_split_terms(prev_terms)
    return agree, tracker


def test_full_size_tracker_64_frame_sequence_against_reference(dev, models):
    """the 64-frame 800x1333 sequence through Tracker.step against the reference's own Tracker on CPU
    (tests/golden/make_golden_full.py tracker64: 9 514 track ids, up to 1 980 simultaneous track queries), run until the first
    frame whose set of live ids differs (after that the track queries fed back differ: another sequence).

    What decides ids here: the fixture records (a) that no score of any frame comes within 0.35 of a score threshold and
    (b) PER FRAME the smallest distance of any IoU an NMS pass looked at to its threshold.  With hundreds of near-identical
    random-weight boxes that margin is 1.3e-2 at frame 0, below 1e-3 from frame 4, 9.3e-6 at frame 24 and 5.7e-6 at frame 26.
    Two fp32 implementations that agree on every box to 1.5e-6 (what this suite measures between the MI355X path and the
    reference CPU path) agree on an IoU only to a few 1e-6: decisions with margins of that size are not pinned by ANY
    fp32-accurate implementation -- the reference on another BLAS would not reproduce them either.

    Measured on MI355X (profiles/r04_id_parity_64.txt, r04_pytest_64_frame.txt), frames that agree:
        fp32 libraries (hipBLASLt / MIOpen)          64 (all) in one run, 59 in another (margin 1.4e-5 at frame 59)
        six-term split product                       26 in every run (margin 5.7e-6 at frame 26)
        fp16 split product (the default since r4)    all 64 in both runs of profiles/r04_id_parity_64_with_fp16.txt
        three-term bf16 split product (rounds 2-4)   14 in two runs (margin 6.8e-4 at frame 14), 32 in a third after the
                                                     convolutions changed kernels (margin 2.0e-5) -- REMOVED in round 5
    i.e. products good to 2^-16 flipped a decision whose margin was 7e-4 -- forty times the margin fp32-class arithmetic
    needs -- and whether they did depended on the kernel's summation order; that is why the products the library has are
    fp32-class (the fp16 pieces: 22 + 1 significand bits; six bf16 terms: all 24).  Asserted: the default and the six-term
    product agree on every frame in front of the first one whose NMS margin is below 2e-5 (24 frames); every row of the agreeing
    frames matches in id / frame / source query, boxes and scores.  (The fixture whose decisions ARE pinned -- every margin
    >= 0.08 -- is test_full_size_well_conditioned_64_frames_every_id below: all 64 frames, every set-up.)"""
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg2_64.npz"))
    assert float(z["min_score_margin"]) > 1e-2
    margins = z["nms_iou_margin_per_frame"]
    rows = z["rows"]
    n_run = 34
    gold = [sorted(int(r[0]) for r in rows[rows[:, 1] == f]) for f in range(n_run)]

    def first_below(thr):
        return int(np.argmax(margins < thr)) if (margins < thr).any() else len(margins)
    agree6, tracker = _ids_until_divergence(models, dev, gold, n_run, True, 6)
    agree16, tracker16 = _ids_until_divergence(models, dev, gold, n_run, True, 16)
    print("64-frame fixture: six terms agree on %d frames (NMS margin of the first differing frame %.1e), fp16 pieces on %d (%.1e); "
          "frames in front of the first margin < 2e-5: %d" % (
              agree6, margins[min(agree6, len(margins) - 1)], agree16, margins[min(agree16, len(margins) - 1)], first_below(2e-5)))
    assert agree6 >= first_below(2e-5), (agree6, margins[:n_run].tolist())
    assert agree16 >= first_below(2e-5), (agree16, margins[:n_run].tolist())   # the fp16 product is held to the six-term bar
    if fused_default_terms() == 16:
        agree6, tracker = agree16, tracker16
    results = tracker.get_results()
    got = np.array([[tid, f, *results[tid][f]['bbox'].tolist(), float(results[tid][f]['score']), results[tid][f]['obj_ind']]
                    for tid in sorted(results) for f in sorted(results[tid]) if f < agree6], dtype=np.float64)
    want = rows[rows[:, 1] < agree6]
    assert got.shape == want.shape
    np.testing.assert_array_equal(got[:, [0, 1, 7]], want[:, [0, 1, 7]])   # id, frame, source query
    np.testing.assert_allclose(got[:, 2:6], want[:, 2:6], atol=1e-3 * max(um.FULL_ORIG))
    np.testing.assert_allclose(got[:, 6], want[:, 6], atol=1e-3)


# ------------------------------------------------------------------ sequences whose decisions are well-conditioned (round 5)
_wc_models = {}


def _wc_model(case, dev):
    """The case's seeded model with um.shape_well_conditioned's planted circuit (its own instance: the shaping edits weights)."""
    from trackformer_amd import config, factory
    if case not in _wc_models:
        model, post, args = um.build(case, factory.build_model, config.make_args, device=dev)
        um.shape_well_conditioned(model)
        model.to(dev).tracking()
        _wc_models[case] = (model, post, args)
    return _wc_models[case]


def _run_wc_tracker(case, dev, setup, n_frames):
    from trackformer_amd import config, fused, runtime
    from trackformer_amd.graphed import GraphedDetector
    from trackformer_amd.tracker import Tracker
    model, post, args = _wc_model(case, dev)
    detector = model
    prev_split = fused.set_split_linear(setup in _SPLIT_SETUPS)
    prev_terms = fused.set_split_terms(_SPLIT_SETUPS.get(setup, 6))
    try:
        if setup != "eager":
            runtime.configure_inference(verbose=False)
            detector = GraphedDetector(model)
        tracker = Tracker(detector, post, config.tracker_cfg(), False)
        tracker.reset()
        active = []
        with torch.no_grad():
            for blob in um.full_tracker_sequence(n_frames=n_frames):
                tracker.step(dict(blob, img=blob['img'].to(dev)))
                active.append(len(tracker.tracks))
    finally:
        fused.set_split_linear(prev_split)
        fused.set_split_terms(prev_terms)
    results = tracker.get_results()
    rows = np.array([[tid, f, *results[tid][f]['bbox'].tolist(), float(results[tid][f]['score']), results[tid][f]['obj_ind']]
                     for tid in sorted(results) for f in sorted(results[tid])], dtype=np.float64)
    return tracker, rows, active


def _compare_wc(z, tracker, rows, active):
    # the fixture's own record of how far every decision of every frame was from flipping: asserted WIDE, so that the
    # comparison below covers all frames unconditionally
    assert float(z["min_score_margin_per_frame"].min()) >= 1e-2
    assert float(z["nms_iou_margin_per_frame"].min()) >= 1e-3
    assert float(z["nms_order_margin_per_frame"].min()) >= 1e-2
    assert int(z["num_tracks"]) == tracker.track_num
    assert int(z["num_reids"]) == tracker.num_reids
    assert z["active_per_frame"].tolist() == active
    assert rows.shape == z["rows"].shape
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])   # id, frame, source query
    np.testing.assert_allclose(rows[:, 2:6], z["rows"][:, 2:6], atol=1e-3 * max(um.FULL_ORIG))
    np.testing.assert_allclose(rows[:, 6], z["rows"][:, 6], atol=1e-3)
    n = len(active)
    ids = [set(z["rows"][z["rows"][:, 1] == f][:, 0].astype(int)) for f in range(n)]
    assert all(ids[f] - ids[f - 1] and ids[f - 1] - ids[f] for f in range(1, n))   # births and terminations in every frame
    assert set.intersection(*ids)                                                   # ... and tracks that live through all of it


@pytest.mark.parametrize("setup", ["graph_split_linear", "graph_split6", "graph_tuned", "eager"])
def test_full_size_well_conditioned_64_frames_every_id(dev, setup):
    """VERDICT r04 task 1(b): 64 frames of 800x1333 through Tracker.step against the reference's own Tracker
    (tests/golden/make_golden_full.py tracker_wc64), ALL 64 frames, ids bit-exact, in the package's default arithmetic (fp16
    split product), the six-term product, the fp32 libraries, and eager.  The detector is the seeded cfg-2 model with
    tests/util_models.shape_well_conditioned's planted circuit: every frame has new tracks, terminated tracks, re-detections
    suppressed by the tracks they duplicate (IoU ~0.99 against the 0.9 threshold) and places contested by two queries --
    and no score within 0.08 of a threshold, no IoU within 0.06 of one (recorded per frame in the fixture, asserted here)."""
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg2_wc64.npz"))
    tracker, rows, active = _run_wc_tracker("cfg2_full", dev, setup, len(z["active_per_frame"]))
    _compare_wc(z, tracker, rows, active)


def _run_wc_tracker_pipelined(case, dev, setup, n_frames, host_frames=False, unobserved=False, depth=None):
    """The loop `bench.py` TIMES (run_tracking): step_async(t) -> step_prepare(t + 1, t + 2) -> step_finish(t), the image-only
    halves of the next frames on GraphedDetector's side streams into its rotating buffers while the host associates frame t
    (depth: frames ahead; None = Tracker.look_ahead, the policy of the timed loop).  Frames
    resident in HBM before the sequence starts (`image_ready`, the driver's line) or in pinned host memory (uploaded on the
    side stream inside the step, as the reference's step does: tracker.py:283-284).  -> (tracker, rows, active, prepared)."""
    from trackformer_amd import config, fused, runtime
    from trackformer_amd.graphed import GraphedDetector
    from trackformer_amd.tracker import Tracker
    model, post, args = _wc_model(case, dev)
    prev_split = fused.set_split_linear(setup in _SPLIT_SETUPS)
    prev_terms = fused.set_split_terms(_SPLIT_SETUPS.get(setup, 6))
    try:
        runtime.configure_inference(verbose=False)
        tracker = Tracker(GraphedDetector(model), post, config.tracker_cfg(), False)
        tracker.reset()
        blobs = []
        for blob in um.full_tracker_sequence(n_frames=n_frames):
            blobs.append(dict(blob, img=blob['img'].pin_memory() if host_frames else blob['img'].to(dev)))
        torch.cuda.synchronize(dev)
        active, prepared = [], 0
        with torch.no_grad():
            if unobserved:
                # the reference's own loop (src/track.py:130-134): step() per frame and nothing else -- Tracker.step defers the
                # association of frame t into step(t + 1), after the image-only half of frame t + 1 has been enqueued
                orig_prepare = tracker.step_prepare

                def counting_prepare(blob, **kw):
                    nonlocal prepared
                    ok = orig_prepare(blob, **kw)
                    prepared += bool(ok)
                    return ok
                tracker.step_prepare = counting_prepare
                for blob in blobs:
                    tracker.step(blob)
                    assert tracker.__dict__.get("_deferred_handle") is not None
                active = None
            else:
                prepared = um.pipelined_loop(tracker, blobs, depth=depth, on_finish=lambda i: active.append(len(tracker.tracks)),
                                             image_ready=not host_frames)
        torch.cuda.synchronize(dev)
    finally:
        fused.set_split_linear(prev_split)
        fused.set_split_terms(prev_terms)
    results = tracker.get_results()
    rows = np.array([[tid, f, *results[tid][f]['bbox'].tolist(), float(results[tid][f]['score']), results[tid][f]['obj_ind']]
                     for tid in sorted(results) for f in sorted(results[tid])], dtype=np.float64)
    return tracker, rows, active, prepared


@pytest.mark.parametrize("frames", ["hbm_frames", "host_frames", "hbm_frames_one_ahead"])
def test_full_size_pipelined_tracker_64_frames_every_id(dev, frames):
    """VERDICT r05 task 1(a): the configuration `bench.py`'s `value` is measured in -- 800x1333, graph_split_linear (the fp16
    split product), two graphs, the image-only half of the next frame prepared on the side stream -- over all 64 frames of the
    well-conditioned sequence against the reference's own Tracker (full_tracker_cfg2_wc64.npz): every id of every frame,
    and at least 60 of the 63 possible frames really were prepared (the first ones run before the graphs exist)."""
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg2_wc64.npz"))
    tracker, rows, active, prepared = _run_wc_tracker_pipelined("cfg2_full", dev, "graph_split_linear", len(z["active_per_frame"]),
                                                                host_frames=frames == "host_frames",
                                                                depth=1 if frames.endswith("one_ahead") else None)
    _compare_wc(z, tracker, rows, active)
    assert prepared >= 60, prepared
    assert tracker.look_ahead == 2   # (two frames ahead is the timed loop's policy for this model: round 6)


def test_full_size_unobserved_step_loop_64_frames_every_id(dev):
    """Round 6: what an UNMODIFIED reference caller gets -- `for blob in sequence: tracker.step(blob)` with host frames, as
    src/track.py:130-134 drives it -- runs the pipelined schedule by itself (deferred association, Tracker.step): all 64 frames
    of the well-conditioned sequence at 800x1333, every id, and the frames really were prepared ahead."""
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg2_wc64.npz"))
    n = len(z["active_per_frame"])
    tracker, rows, _, prepared = _run_wc_tracker_pipelined("cfg2_full", dev, "graph_split_linear", n, host_frames=True, unobserved=True)
    _compare_wc(z, tracker, rows, z["active_per_frame"].tolist())
    assert prepared >= 60, prepared


@pytest.mark.parametrize("loop", ["pipelined", "unobserved"])
def test_full_size_multi_frame_tracker_prepared_ahead_matches_reference(dev, loop):
    """Round 6: BASELINE cfg 4 with the image-only half of frame t + 1 (backbone, the encoder over frame t + 1 and over frame t's
    backbone features) prepared ahead on the side stream -- the loop `bench.py --config cfg4` times, and the plain step() loop
    with its deferred association -- 12 frames at 800x1333 against the reference's Tracker."""
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg4.npz"))
    n = len(z["active_per_frame"])
    tracker, rows, active, prepared = _run_wc_tracker_pipelined("cfg4_full", dev, "graph_split_linear", n, host_frames=(loop == "unobserved"),
                                                                unobserved=(loop == "unobserved"))
    _compare_wc(z, tracker, rows, active if active is not None else z["active_per_frame"].tolist())
    assert prepared >= n - 4, prepared   # (the first frame attends to itself; the graphs of a shape exist from its second sight on)


@pytest.mark.parametrize("setup", ["eager", "graph_split_linear", "graph_tuned"])
def test_full_size_multi_frame_tracker_matches_reference(dev, setup):
    """VERDICT r04 task 1(a): BASELINE cfg 4 (hidden 288, 500 object queries, 8 decoder levels) under the Tracker for 12
    frames of 800x1333: the reference carries frame t's backbone features to frame t + 1 through a deque
    (tracker.py:74,306,547); the repo does the same and, under HIP graphs, the features alias GraphedDetector's static
    buffers (trackformer_amd/tracker.py) -- the path `bench.py --config cfg4` times."""
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg4.npz"))
    tracker, rows, active = _run_wc_tracker("cfg4_full", dev, setup, len(z["active_per_frame"]))
    _compare_wc(z, tracker, rows, active)


# ------------------------------------------------------------------ the round-3 routes (defaults since their hardware validation:
# profiles/r03_optin_pytest_optin.txt).  "graph_split_linear" above runs ALL of them at once; below each family is also
# switched off on its own (the off-switches stay honest) and unit-tested against PyTorch.


@pytest.mark.parametrize("routes", [(True, False), (False, True), (False, False)], ids=["conv1x1_only", "conv3x3_only", "library_convolutions"])
@pytest.mark.parametrize("case", um.FULL_DETECTOR_CASES)
def test_conv_split_routes_full_size(dev, models, case, routes):
    """The bottleneck convolutions of the backbone through the split-product kernels with the FrozenBN shift / identity /
    ReLU epilogue (backbone.set_conv1x1_split / set_conv3x3_split): BASELINE-size model against the reference goldens,
    bench set-up."""
    from trackformer_amd import backbone
    prev = backbone.set_conv1x1_split(routes[0])
    prev3 = backbone.set_conv3x3_split(routes[1])
    try:
        model, out, res, feats, memory = _forward(case, models, dev, "graph_split_linear")
        dbox, dlogit = _compare(case, model, out, res, feats, memory)
        print("%s / conv split %s: max |d boxes| %.2e, max |d logits| %.2e" % (case, routes, dbox, dlogit))
    finally:
        backbone.set_conv1x1_split(prev)
        backbone.set_conv3x3_split(prev3)


def test_conv_split_route_tracker_ids(dev, models):
    from trackformer_amd import backbone
    prev = backbone.set_conv1x1_split(True)
    prev3 = backbone.set_conv3x3_split(True)
    try:
        tracker, rows, active = _run_tracker(models, dev, "graph_split_linear")
    finally:
        backbone.set_conv1x1_split(prev)
        backbone.set_conv3x3_split(prev3)
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg2.npz"))
    assert int(z["num_tracks"]) == tracker.track_num and z["active_per_frame"].tolist() == active
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])


def test_conv_split_backbone_layer_outputs(dev):
    """Bottleneck by bottleneck: the split route against the library convolutions on the same weights (1e-3 relative to
    the feature scale; the three-term product is ~2^-16 per layer)."""
    from trackformer_amd import backbone
    torch.manual_seed(0)
    blk = backbone.Bottleneck(256, 64).to(dev).eval()
    for m in blk.modules():
        if isinstance(m, backbone.FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5)
            m.bias.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(1, 256, 50, 84, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        prev, prev3 = backbone.set_conv1x1_split(False), backbone.set_conv3x3_split(False)
        try:
            ref = blk(x)   # library convolutions
        finally:
            backbone.set_conv1x1_split(prev)
            backbone.set_conv3x3_split(prev3)
        for both in (False, True):
            prev = backbone.set_conv1x1_split(True)
            prev3 = backbone.set_conv3x3_split(both)
            try:
                got = blk(x)
            finally:
                backbone.set_conv1x1_split(prev)
                backbone.set_conv3x3_split(prev3)
            assert got.is_contiguous(memory_format=torch.channels_last)
            assert float((got - ref).abs().max()) < 1e-3 * float(ref.abs().max())
        # a strided bottleneck (3 x 3 with stride 2 + a strided projection that stays in the library)
        ds = torch.nn.Sequential(torch.nn.Conv2d(256, 512, 1, stride=2, bias=False), backbone.FrozenBatchNorm2d(512)).to(dev)
        blk2 = backbone.Bottleneck(256, 128, stride=2, downsample=ds).to(dev).eval()
        prev, prev3 = backbone.set_conv1x1_split(False), backbone.set_conv3x3_split(False)
        try:
            ref2 = blk2(x)
        finally:
            backbone.set_conv1x1_split(prev)
            backbone.set_conv3x3_split(prev3)
        prev = backbone.set_conv1x1_split(True)
        prev3 = backbone.set_conv3x3_split(True)
        try:
            got2 = blk2(x)
        finally:
            backbone.set_conv1x1_split(prev)
            backbone.set_conv3x3_split(prev3)
        assert got2.shape == ref2.shape and float((got2 - ref2).abs().max()) < 1e-3 * float(ref2.abs().max())


@pytest.mark.parametrize("case", um.FULL_DETECTOR_CASES)
def test_input_proj_library_route_full_size(dev, models, case):
    """input_proj's 1 x 1 levels as split GEMM + tf_groupnorm_nhwc_f32 is the default (fused.set_input_proj_fused); here
    switched off: library convolution + ATen GroupNorm."""
    from trackformer_amd import fused
    prev = fused.set_input_proj_fused(False)
    try:
        model, out, res, feats, memory = _forward(case, models, dev, "graph_split_linear")
        dbox, dlogit = _compare(case, model, out, res, feats, memory)
        print("%s / library input_proj: max |d boxes| %.2e, max |d logits| %.2e" % (case, dbox, dlogit))
    finally:
        fused.set_input_proj_fused(prev)


@pytest.mark.parametrize("n,c,h,w,groups", [(1, 256, 50, 84, 32), (2, 288, 13, 21, 32), (1, 64, 7, 5, 8)])
def test_groupnorm_nhwc_matches_torch(dev, n, c, h, w, groups):
    from trackformer_amd import fused
    torch.manual_seed(0)
    gn = torch.nn.GroupNorm(groups, c).to(dev)
    with torch.no_grad():
        gn.weight.normal_(1, 0.3)
        gn.bias.normal_(0, 0.3)
    x = (torch.randn(n, c, h, w, device=dev) * 3 + 1).contiguous(memory_format=torch.channels_last)
    ref = gn(x)
    got = fused.groupnorm_nhwc(x.permute(0, 2, 3, 1).reshape(n * h * w, c), n, gn)
    assert got is not None
    assert torch.allclose(got.view(n, h, w, c).permute(0, 3, 1, 2), ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("case", um.FULL_DETECTOR_CASES)
def test_fused_box_refine_full_size(dev, models, case):
    """The decoder's iterative box refinement as one launch per layer (fused.set_box_refine_fused)."""
    from trackformer_amd import fused
    from trackformer_amd.nested import inverse_sigmoid
    g = torch.Generator().manual_seed(0)
    for ref_dim in (2, 4):
        delta = torch.randn(2, 400, 4, generator=g).to(dev)
        ref = torch.rand(2, 400, ref_dim, generator=g).to(dev)
        prev = fused.set_box_refine_fused(True)
        try:
            got = fused.box_refine(delta, ref)
        finally:
            fused.set_box_refine_fused(prev)
        exp = delta.clone()
        exp[..., :ref_dim] += inverse_sigmoid(ref)
        assert got is not None and torch.allclose(got, exp.sigmoid(), atol=1e-6, rtol=1e-5)
    prev = fused.set_box_refine_fused(False)   # the element-wise ATen formulation
    try:
        model, out, res, feats, memory = _forward(case, models, dev, "graph_split_linear")
        _compare(case, model, out, res, feats, memory)
    finally:
        fused.set_box_refine_fused(prev)


@pytest.mark.parametrize("case", um.FULL_DETECTOR_CASES)
def test_separate_ffn_route_full_size(dev, models, case):
    """The feed-forward blocks in one launch each (tf_ffn_fused_f32) and output projection + residual + LayerNorm in one
    launch (tf_linear_res_ln_f32) are the defaults; here switched off (separate linears + tf_add_layernorm_f32), hidden 256
    (cfg 2) and 288 (cfg 4): BASELINE-size model against the reference goldens and, for cfg 2, the tracker's ids."""
    from trackformer_amd import fused
    prev, prev_ln = fused.set_ffn_fused(False), fused.set_linear_ln_fused(False)
    try:
        model, out, res, feats, memory = _forward(case, models, dev, "graph_split_linear")
        dbox, dlogit = _compare(case, model, out, res, feats, memory)
        print("%s / separate ffn + projection norm: max |d boxes| %.2e, max |d logits| %.2e" % (case, dbox, dlogit))
        if "cfg2" not in case:
            return
        tracker, rows, active = _run_tracker(models, dev, "graph_split_linear")
    finally:
        fused.set_ffn_fused(prev)
        fused.set_linear_ln_fused(prev_ln)
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg2.npz"))
    assert int(z["num_tracks"]) == tracker.track_num and z["active_per_frame"].tolist() == active
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])


@pytest.mark.parametrize("shape,cout,stride,ksplit", [((1, 2048, 25, 42), 256, 2, 36), ((1, 512, 25, 42), 512, 1, 5), ((1, 256, 100, 167), 256, 2, 2)])
def test_conv3x3_split_k(dev, shape, cout, stride, ksplit):
    """tf_conv3x3_splitk_f32: the K loop cut into workgroups with a deterministic second pass, against the library
    convolution (1e-3 of the output scale) and run twice (bit-identical: no atomics)."""
    from trackformer_amd import _cabi, fused
    g = torch.Generator().manual_seed(shape[1] + ksplit)
    x = torch.randn(*shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, shape[1], 3, 3, generator=g) / (3 * shape[1] ** 0.5)).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    taps = w.permute(0, 2, 3, 1).reshape(cout, 9 * shape[1]).contiguous()
    hi, mid, lo, wsc = fused._split_weight(taps)
    n, cin, h, wd = shape
    ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
    outs = []
    for _ in range(2):
        y = torch.full((n, ho, wo, cout), float("nan"), device=dev)
        ws = torch.empty((ksplit, n * ho * wo * cout), device=dev)
        rc = _cabi.lib().tf_conv3x3_splitk_f32(x.data_ptr(), hi.data_ptr(), mid.data_ptr(), 0 if lo is None else lo.data_ptr(),
                                               0 if wsc is None else wsc.data_ptr(), b.data_ptr(), y.data_ptr(), ws.data_ptr(), ksplit, n, h, wd, cin, cout, stride, 1, fused._stream(dev))
        _cabi.check(rc, "tf_conv3x3_splitk_f32")
        outs.append(y)
    ref = torch.relu(torch.nn.functional.conv2d(x, w, b, stride=stride, padding=1)).permute(0, 2, 3, 1)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - ref).abs().max()) < 1e-3 * float(ref.abs().max())


@pytest.mark.parametrize("shape,cout,stride", [((1, 1024, 50, 84), 256, 1), ((1, 2048, 25, 42), 512, 1), ((1, 2048, 25, 42), 256, 1),
                                               ((1, 1024, 50, 84), 2048, 2)])
def test_conv1x1_split_k(dev, shape, cout, stride):
    """The few-pixel 1 x 1 convolutions under a long K (layer3 / layer4 conv1, the coarse input projections) take the
    convolution kernel with their K loop cut (tf_conv1x1_splitk_f32): against the uncut kernel (same products, another sum
    order), the library convolution, and run twice (bit-identical: no atomics)."""
    from trackformer_amd import fused
    g = torch.Generator().manual_seed(shape[1] + cout)
    x = torch.randn(*shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, shape[1], generator=g) / shape[1] ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    m = shape[0] * ((shape[2] - 1) // stride + 1) * ((shape[3] - 1) // stride + 1)
    cut = fused._conv_ksplit(m, shape[1], cout)
    y1, y2 = fused.conv3x3(x, w, b, True, stride), fused.conv3x3(x, w, b, True, stride)
    prev = fused.set_conv1x1_splitk(False)
    try:
        y0 = fused.conv3x3(x, w, b, True, stride)
    finally:
        fused.set_conv1x1_splitk(prev)
    ref = torch.relu(torch.nn.functional.conv2d(x, w[:, :, None, None], b, stride=stride))
    scale = float(ref.abs().max())
    assert torch.equal(y1, y2)
    assert float((y1 - y0).abs().max()) < 1e-5 * scale and float((y1 - ref).abs().max()) < 1e-3 * scale
    if cut >= 3:
        assert fused.conv1x1_wants_split_k(m, shape[1], cout)
    else:
        assert torch.equal(y1, y0)                     # two pieces do not pay for a 1 x 1 convolution: left alone


@pytest.mark.parametrize("terms", [6, 16], ids=["six_terms", "fp16_pieces"])
@pytest.mark.parametrize("shape,cout,ks,stride", [((1, 64, 200, 334), 64, 3, 1), ((1, 256, 100, 167), 256, 3, 2), ((1, 512, 25, 42), 512, 3, 1),
                                                  ((2, 128, 37, 53), 160, 3, 1), ((1, 1024, 50, 84), 2048, 1, 2),
                                                  ((3, 32, 100, 167), 16, 3, 1), ((2, 288, 25, 42), 128, 3, 1)])   # (the mask head's lay5 / lay2: halo form only)
def test_conv3x3_split_at_resnet_shapes(dev, shape, cout, ks, stride, terms):
    """split_conv3_kernel (buffer-resource fetches: a tap outside the image reads zeros from beyond num_records; DESIGN.md
    section 4.4) at ResNet-50's shapes of the 800 x 1333 frame -- borders on all four sides, the last row block partial, the
    split-K pieces fused.conv3x3 chooses -- against a float64 convolution: both products at fp32 round-off (the library's fp32
    convolution is no closer)."""
    from trackformer_amd import fused
    g = torch.Generator().manual_seed(shape[1] + cout)
    x = torch.randn(*shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, shape[1], ks, ks, generator=g) / (ks * shape[1] ** 0.5)).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    taps = w.permute(0, 2, 3, 1).reshape(cout, ks * ks * shape[1]).contiguous()
    prev = fused.set_split_terms(terms)
    try:
        prev_stream = fused.set_conv_stream("all")
        prev_halo = fused.set_conv_halo(False)
        try:
            y = fused.conv3x3(x, taps, b, True, stride)          # the stream GEMM (tf_conv_packed_f32), tap-major K
            fused.set_conv_halo(True)
            y_halo = fused.conv3x3(x, taps, b, True, stride)     # (stride-1 3 x 3 only) the halo form: channel-slice-major K
            fused.set_conv_stream(False)
            y_block = fused.conv3x3(x, taps, b, True, stride)   # the LDS-staged block kernel (tf_conv3x3_split_f32)
        finally:
            fused.set_conv_halo(prev_halo)
            fused.set_conv_stream(prev_stream)
        torch.cuda.synchronize()
    finally:
        fused.set_split_terms(prev)
    assert y is not None and y_block is not None and y_halo is not None
    ho, wo = y.shape[2], y.shape[3]
    if fused._conv_ksplit(shape[0] * ho * wo, ks * ks * shape[1], cout) == 1 or (ks == 1 and not fused.conv1x1_wants_split_k(shape[0] * ho * wo, shape[1], cout)):
        assert torch.equal(y, y_block)                       # same products in the same order
    else:
        assert float((y - y_block).abs().max()) < 1e-5 * float(y_block.abs().max())   # the K pieces are cut elsewhere
    if ks == 3 and stride == 1:   # same products, the nine taps of a 32-channel slice before the next slice
        assert float((y_halo - y_block).abs().max()) < 1e-5 * float(y_block.abs().max())
        y = y_halo                # ... and it is the default: the one held against float64 below
    else:
        assert torch.equal(y_halo, y)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=ks // 2))
    lib32 = torch.relu(torch.nn.functional.conv2d(x, w, b, stride=stride, padding=ks // 2))
    scale = float(ref.abs().max())
    err, err_lib = float((y.double() - ref).abs().max()) / scale, float((lib32.double() - ref).abs().max()) / scale
    print("conv %s -> %d, %d x %d / %d, %d terms: max err / max |y| %.2e (library fp32 convolution: %.2e)" % (
        shape, cout, ks, ks, stride, terms, err, err_lib))
    assert err < 2e-6
    assert err < 4 * err_lib + 1e-7


@pytest.mark.parametrize("case", um.FULL_DETECTOR_CASES)
def test_every_round3_route_switched_off_full_size(dev, models, case):
    """Every switch of DESIGN.md section 4.3 OFF at once (the round-2 defaults: library convolutions, separate linears and
    LayerNorms, msda_fwd_f32_buf / msda_bwd_f32_sorted): BASELINE-size model against the
    reference goldens and, for cfg 2, the tracker's ids."""
    from trackformer_amd import _cabi, backbone, fused
    lib = _cabi.lib()
    setters = [backbone.set_conv1x1_split, backbone.set_conv3x3_split, fused.set_input_proj_fused, fused.set_box_refine_fused,
               fused.set_ffn_fused, fused.set_linear_ln_fused, fused.set_stem_pool_fused, fused.set_pos_add_fused,
               fused.set_stem_conv_split, fused.set_heads_split]
    prev = [s(False) for s in setters]
    opts = {b"direct9": 0}
    prev_opts = {k: lib.tf_msda_set_option(k, v) for k, v in opts.items()}
    try:
        model, out, res, feats, memory = _forward(case, models, dev, "graph_split_linear")
        dbox, dlogit = _compare(case, model, out, res, feats, memory)
        print("%s / every round-3 route off: max |d boxes| %.2e, max |d logits| %.2e" % (case, dbox, dlogit))
        if "cfg2" in case:
            tracker, rows, active = _run_tracker(models, dev, "graph_split_linear")
            z = np.load(os.path.join(GOLDEN, "full_tracker_cfg2.npz"))
            assert int(z["num_tracks"]) == tracker.track_num and z["active_per_frame"].tolist() == active
            np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])
    finally:
        for s, p in zip(setters, prev):
            s(p)
        for k, v in prev_opts.items():
            lib.tf_msda_set_option(k, v)


# ------------------------------------------------------------------ BASELINE cfg 1 / cfg 3 / cfg 5 at their quoted sizes (round 4:
# tests/golden/make_golden_full.py cfg1_full / cfg3_full / cfg5_full / tracker_cfg5, from the reference's own classes on CPU)


def run_cfg1(device):
    """Plain DETR (detr.py:62-128): 100 object queries, coco classes, dense encoder / decoder with ffn 2048, one 480 x 640 frame."""
    from trackformer_amd import config, factory
    model, post, args = um.build("cfg1_full", factory.build_model, config.make_args, device=device)
    model.to(device).eval()
    img, _, target = um.model_inputs("cfg1_full", args.hidden_dim)
    with torch.no_grad():
        out, _, feats, memory, hs = model(img.to(device), target)
        res = post['bbox'](out, torch.tensor([list(um.FULL_IMG_CFG1)], device=device))[0]
    return model, out, res, feats, memory


@pytest.mark.parametrize("split", [False, True], ids=["fp32_libraries", "split_product"])
def test_cfg1_plain_detr_480x640_matches_reference(dev, split):
    from trackformer_amd import fused
    prev = fused.set_split_linear(split)
    try:
        model, out, res, feats, memory = run_cfg1(dev)
    finally:
        fused.set_split_linear(prev)
    dbox, dlogit = _compare("cfg1_full", model, out, res, feats, memory, orig=um.FULL_IMG_CFG1)
    print("cfg1_full / %s: max |d boxes| %.2e, max |d logits| %.2e" % ("split" if split else "library", dbox, dlogit))


@pytest.fixture
def no_solver_search():
    """The mask head's convolutions have one batch size per number of masks asked for; with `cudnn.benchmark` left on by an
    earlier `runtime.configure_inference` MIOpen times its solvers for each of them (the three cfg-5 tests took 500 s of the
    GPU suite: profiles/r04_pytest_durations.txt).  Correctness does not depend on the solver."""
    prev = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False
    yield
    torch.backends.cudnn.benchmark = prev


def _cfg5_model(dev):
    from trackformer_amd import config, factory
    model, post, args = um.build("cfg5_full", factory.build_model, config.make_args, device=dev)
    model.to(dev).tracking()
    return model, post, args


def test_cfg5_mask_head_800x1333_matches_reference(dev):
    """BASELINE cfg 5 (detr_segmentation.py:41-71 on the tracking detector): detector outputs, the mask logits of every 50th
    query and the post-processed probabilities of the first three against the reference's classes on CPU."""
    model, post, args = _cfg5_model(dev)
    img, _, target = um.model_inputs("cfg5_full", args.hidden_dim)
    prev_benchmark = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False   # see no_solver_search (this function is also called by the CPU suite: no fixture)
    try:
        with torch.no_grad():
            out, _, feats, memory, hs = model(img.to(dev), um.to_device(target, dev), None)
            sizes = torch.tensor([list(um.FULL_ORIG)], device=dev)
            res = post['bbox'](out, sizes)
            seg = post['segm'](res, out, sizes, torch.tensor([list(um.FULL_IMG)], device=dev), return_probs=True)
    finally:
        torch.backends.cudnn.benchmark = prev_benchmark
    dbox, dlogit = _compare("cfg5_full", model, out, res[0], feats, memory)
    z = np.load(os.path.join(GOLDEN, "full_cfg5_full.npz"))
    assert list(out['pred_masks'].shape) == z['pred_masks_shape'].tolist()
    pm = out['pred_masks'][:, ::um.FULL_MASK_QUERY_STRIDE].cpu().numpy()
    scale = max(1.0, float(np.abs(z['pred_masks']).max()))
    np.testing.assert_allclose(pm, z['pred_masks'], atol=1e-3 * scale)
    np.testing.assert_allclose(seg[0]['masks'][:3, :, ::4, ::4].cpu().numpy(), z['post_masks'], atol=1e-3)
    print("cfg5_full: max |d boxes| %.2e, max |d logits| %.2e, max |d mask logits| %.2e (scale %.1f)" % (
        dbox, dlogit, float(np.abs(pm - z['pred_masks']).max()), scale))


@pytest.mark.parametrize("lazy", [False, True], ids=["masks_for_every_query", "lazy_masks"])
def test_cfg5_tracker_with_masks_800x1333_matches_reference(dev, lazy, no_solver_search):
    """The reference Tracker + mask head (tracker.py:521-547) over three 800 x 1333 frames: ids / frames / source queries exact,
    boxes and scores within tolerance, the mask area every track owns within 2 % of the image (full_tracker_cfg5.npz)."""
    from trackformer_amd import config
    from trackformer_amd.tracker import Tracker
    model, post, args = _cfg5_model(dev)
    tracker = Tracker(model, post, config.tracker_cfg(), False, lazy_masks=lazy)
    tracker.reset()
    with torch.no_grad():
        for blob in um.full_tracker_sequence(n_frames=3):
            tracker.step(dict(blob, img=blob['img'].to(dev)))
    results = tracker.get_results()
    z = np.load(os.path.join(GOLDEN, "full_tracker_cfg5.npz"))
    rows, areas = [], []
    for tid in sorted(results):
        for f in sorted(results[tid]):
            r = results[tid][f]
            rows.append([tid, f, *r['bbox'].tolist(), float(r['score']), r['obj_ind']])
            areas.append(int(np.asarray(r['mask']).sum()))
            assert list(np.asarray(r['mask']).shape) == z["mask_shape"].tolist()
    rows = np.array(rows, dtype=np.float64)
    assert rows.shape == z["rows"].shape and int(z["num_tracks"]) == tracker.track_num
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])
    np.testing.assert_allclose(rows[:, 2:6], z["rows"][:, 2:6], atol=1e-3 * max(um.FULL_ORIG))
    np.testing.assert_allclose(rows[:, 6], z["rows"][:, 6], atol=1e-3)
    n_px = int(np.prod(z["mask_shape"]))
    assert np.abs(np.array(areas) - z["mask_areas"]).max() <= 0.02 * n_px


def test_cfg3_training_step_800x1333_batch2_matches_reference(dev):
    """BASELINE cfg 3: one training step (engine.py:119-158, detr_tracking.py:39-183) of the full model at batch 2
    (800 x 1333 + a padded 768 x 1280, 30 boxes per image, previous-frame pass, track-query augmentation, SetCriterion, backward
    through tf_msda_backward_f32 at S = 22 223) against the reference on CPU: every loss, the weighted total, and the gradient
    norm of EVERY parameter (full_cfg3_full.npz)."""
    from trackformer_amd import config, factory
    model, criterion, args = um.build_train(factory.build_model, config.make_args, device=dev, full=True)
    model.to(dev)
    criterion.to(dev)
    samples, targets = um.train_batch(device=dev, full=True)
    loss_dict, total, grads = um.train_step(model, criterion, samples, targets)
    z = np.load(os.path.join(GOLDEN, "full_cfg3_full.npz"))
    assert abs(_checksum(model) - float(z["weight_checksum"])) < 1e-6 * float(z["weight_checksum"])
    assert sorted(loss_dict) == z["loss_keys"].tolist()
    got = np.array([loss_dict[k] for k in sorted(loss_dict)])
    np.testing.assert_allclose(got, z["loss_vals"], rtol=1e-3, atol=1e-3)
    assert abs(total - float(z["total"])) <= 1e-3 * abs(float(z["total"]))
    assert len(grads) == int(z["num_grads"]) and sorted(grads) == z["grad_keys"].tolist()
    gn = np.array([grads[k] for k in z["grad_keys"].tolist()])
    rel = np.abs(gn - z["grad_norms"]) / np.maximum(np.abs(z["grad_norms"]), 1e-6 * np.abs(z["grad_norms"]).max())
    print("cfg3_full: total %.6f (reference %.6f), %d gradient norms, max relative difference %.2e (%s)" % (
        total, float(z["total"]), len(gn), float(rel.max()), z["grad_keys"][int(rel.argmax())]))
    np.testing.assert_allclose(gn, z["grad_norms"], rtol=5e-3, atol=1e-6 * float(np.abs(z["grad_norms"]).max()))
