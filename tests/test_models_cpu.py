"""CPU (-m "not gpu"): host-side model + tracker logic against the reference goldens.

The nn.Module graph, track-query handling, post-processing and the Tracker run on CPU with the C
oracle standing in for the HIP operator (monkeypatched here, in the test only -- the product code has
no CPU operator).  Goldens: tests/golden/model_*.npz, tracker_*.npz produced from the reference's own
classes by tests/golden/make_golden_models.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import msda_oracle
from tests import util_models as um
from trackformer_amd import config, factory, msda
from trackformer_amd.tracker import Tracker

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def host_op():
    """The models below run on CPU tensors through the PRODUCT's own host operator (libtf_msda.so: tf_msda_*_host_*,
    csrc/msda_host.cpp -- the real CPU path of SURVEY 8(b)); nothing is patched.  The goldens they are compared with come
    from the reference's classes, so these tests pin the host operator at model level; operator level: test_msda_host.py."""
    from trackformer_amd import _cabi
    _cabi.lib()


@pytest.fixture(params=["host_op", "oracle"])
def cpu_operator(request, monkeypatch):
    """host_op: as above.  oracle: the C oracle patched in as the operator (the checker itself against the same goldens)."""
    if request.param == "oracle":
        monkeypatch.setattr(msda, "MSDeformAttnFunction", msda_oracle.make_torch_function())
    return request.param


def _checksum(model):
    return float(sum(float(v.double().abs().sum()) for v in model.state_dict().values()))


def run_case(case, device="cpu"):
    model, post, args = um.build(case, factory.build_model, config.make_args, device=device)
    model.to(device)
    if hasattr(model, "tracking"):
        model.tracking()
    else:
        model.eval()
    img, prev, target = um.model_inputs(case, args.hidden_dim)
    img, prev, target = img.to(device), prev.to(device), um.to_device(target, device)
    with torch.no_grad():
        prev_features = None
        if args.multi_frame_attention:
            _, _, prev_features, _, _ = model(prev, None, None)
        out, _, feats, memory, hs = model(img, target, prev_features)
        res = post['bbox'](out, torch.tensor([[480, 640]], device=device))[0]
        if case in um.MASK_CASES:
            size, orig = um.MASK_SIZES[case]
            seg = post['bbox'](out, torch.tensor([list(orig)], device=device))
            seg = post['segm'](seg, out, torch.tensor([list(orig)], device=device),
                               torch.tensor([list(size)], device=device), return_probs=True)
            res['post_masks'] = seg[0]['masks']
    return model, out, res, feats


def compare_to_golden(case, model, out, res, feats, box_tol, logit_tol):
    z = np.load(os.path.join(GOLDEN, "model_%s.npz" % case))
    assert abs(_checksum(model) - float(z["weight_checksum"])) < 1e-6 * float(z["weight_checksum"])
    np.testing.assert_allclose(out['pred_boxes'].cpu().numpy(), z['pred_boxes'], atol=box_tol)
    np.testing.assert_allclose(out['pred_logits'].cpu().numpy(), z['pred_logits'], atol=logit_tol)
    np.testing.assert_allclose(out['hs_embed'].cpu().numpy(), z['hs_embed'], atol=logit_tol * 5)
    aux = np.stack([a['pred_boxes'].cpu().numpy() for a in out['aux_outputs']])
    np.testing.assert_allclose(aux, z['aux_boxes'], atol=box_tol)
    np.testing.assert_array_equal(res['labels'].cpu().numpy(), z['labels'])
    np.testing.assert_allclose(res['boxes'].cpu().numpy(), z['boxes'], atol=box_tol * 640)
    f = feats[-1].tensors.cpu().numpy()
    np.testing.assert_allclose(f, z['feat_last'], atol=1e-3 * max(1.0, np.abs(z['feat_last']).max()))
    if case in um.MASK_CASES:   # mask logits [B,Q,H/4,W/4] and post-processed probabilities
        scale = max(1.0, float(np.abs(z['pred_masks']).max()))
        np.testing.assert_allclose(out['pred_masks'].cpu().numpy(), z['pred_masks'],
                                   atol=logit_tol * scale)
        assert res['post_masks'].shape[1:] == (1,) + um.MASK_SIZES[case][1]
        np.testing.assert_allclose(res['post_masks'][:3].cpu().numpy(), z['post_masks'],
                                   atol=logit_tol)


@pytest.mark.parametrize("case", list(um.MODEL_CASES))
def test_model_forward_matches_reference(case, cpu_operator):
    model, out, res, feats = run_case(case)
    compare_to_golden(case, model, out, res, feats, box_tol=2e-5, logit_tol=1e-4)


def run_tracker(reid, device="cpu"):
    model, post, args = um.build("cfg2_deformable_tracking", factory.build_model,
                                 config.make_args, device=device)
    model.to(device)
    model.tracking()
    tracker = Tracker(model, post, config.tracker_cfg(reid=reid), False)
    tracker.reset()
    active, inactive = [], []
    with torch.no_grad():
        for blob in um.tracker_sequence():
            tracker.step(blob)
            active.append(len(tracker.tracks))
            inactive.append(len(tracker.inactive_tracks))
    rows = []
    results = tracker.get_results()
    for tid in sorted(results):
        for f in sorted(results[tid]):
            r = results[tid][f]
            assert r['bbox'].shape == (4,) and r['bbox'].dtype == np.float32
            assert isinstance(r['obj_ind'], int)
            rows.append([tid, f, *r['bbox'].tolist(), float(r['score']), r['obj_ind']])
    return tracker, np.array(rows, dtype=np.float64), active, inactive


def compare_tracker_to_golden(reid, tracker, rows, active, inactive, box_tol_px):
    z = np.load(os.path.join(GOLDEN, "tracker_cfg2_%s.npz" % ("reid" if reid else "default")))
    # track-id assignment: bit-exact (ids, frames they are alive in, source object query)
    assert int(z["num_tracks"]) == tracker.track_num
    assert int(z["num_reids"]) == tracker.num_reids
    assert z["active_per_frame"].tolist() == active
    assert z["inactive_per_frame"].tolist() == inactive
    assert rows.shape == z["rows"].shape
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])
    np.testing.assert_allclose(rows[:, 2:6], z["rows"][:, 2:6], atol=box_tol_px)
    np.testing.assert_allclose(rows[:, 6], z["rows"][:, 6], atol=1e-3)


@pytest.mark.parametrize("reid", [False, True], ids=["default", "reid"])
def test_tracker_sequence_matches_reference(reid, host_op):
    tracker, rows, active, inactive = run_tracker(reid)
    compare_tracker_to_golden(reid, tracker, rows, active, inactive, box_tol_px=0.05)


def run_tracker_variant(name, device="cpu"):
    """Tracker configurations of tests/util_models.TRACKER_VARIANTS (public detections come from the
    fixture: they were generated with the reference model)."""
    var = um.TRACKER_VARIANTS[name]
    z = np.load(os.path.join(GOLDEN, "tracker_cfg2_%s.npz" % name))
    model, post, args = um.build("cfg2_deformable_tracking", factory.build_model,
                                 config.make_args, device=device)
    model.to(device)
    model.tracking()
    tracker = Tracker(model, post, config.tracker_cfg(reid=var['reid'], **var['cfg']), False)
    tracker.reset()
    active, inactive = [], []
    with torch.no_grad():
        for i, blob in enumerate(um.tracker_sequence()):
            if var['dets']:
                blob = dict(blob, dets=torch.from_numpy(z["dets_f%d" % i]))
            tracker.step(blob)
            active.append(len(tracker.tracks))
            inactive.append(len(tracker.inactive_tracks))
    results = tracker.get_results()
    rows = np.array([[tid, f, *results[tid][f]['bbox'].tolist(), float(results[tid][f]['score']),
                      results[tid][f]['obj_ind']]
                     for tid in sorted(results) for f in sorted(results[tid])], dtype=np.float64)
    return tracker, rows, active, inactive


def compare_variant_to_golden(name, tracker, rows, active, inactive, box_tol_px):
    z = np.load(os.path.join(GOLDEN, "tracker_cfg2_%s.npz" % name))
    assert int(z["num_tracks"]) == tracker.track_num
    assert int(z["num_reids"]) == tracker.num_reids
    assert z["active_per_frame"].tolist() == active
    assert z["inactive_per_frame"].tolist() == inactive
    assert rows.shape == z["rows"].shape
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])   # id, frame, query index
    np.testing.assert_allclose(rows[:, 2:6], z["rows"][:, 2:6], atol=box_tol_px)
    np.testing.assert_allclose(rows[:, 6], z["rows"][:, 6], atol=1e-3)


@pytest.mark.parametrize("name", list(um.TRACKER_VARIANTS))
def test_tracker_variants_match_reference(name, host_op):
    tracker, rows, active, inactive = run_tracker_variant(name)
    compare_variant_to_golden(name, tracker, rows, active, inactive, box_tol_px=0.05)


def run_wc_tracker(name, device="cpu", wrap=None, n_frames=None, prepare=False, unobserved=False):
    """Tracker.step over a sequence of tests/util_models.WC_TRACKER_CASES: the well-conditioned detector (the same seeded
    weights with um.shape_well_conditioned's planted circuit).  wrap: detector -> detector (e.g. GraphedDetector).
    prepare: the pipelined form -- step_async(t), step_prepare(t + 1 ..), step_finish(t): the image-only half of the next
    frame(s) is enqueued before this frame's association runs (Tracker.step_prepare; True: as many frames ahead as
    Tracker.look_ahead says, an int: that many -- tests/util_models.pipelined_loop)."""
    case, frames, reid = um.WC_TRACKER_CASES[name]
    model, post, args = um.build(case, factory.build_model, config.make_args, device=device)
    um.shape_well_conditioned(model)
    model.to(device)
    model.tracking()
    tracker = Tracker(wrap(model) if wrap is not None else model, post, config.tracker_cfg(reid=reid), False)
    tracker.reset()
    active, inactive = [], []
    blobs = um.tracker_sequence(n_frames=n_frames or frames)
    with torch.no_grad():
        if unobserved:
            # the reference's loop (src/track.py:130-134): step() per frame, nobody looks at the tracker in between -- the
            # deferred association of Tracker.step (round 6) is outstanding after every step and runs inside the next one
            outstanding = prepared = 0
            orig_prepare = tracker.step_prepare

            def counting_prepare(blob, **kw):
                nonlocal prepared
                ok = orig_prepare(blob, **kw)
                prepared += bool(ok)
                return ok
            tracker.step_prepare = counting_prepare
            for blob in blobs:
                tracker.step(blob)
                outstanding += tracker.__dict__.get("_deferred_handle") is not None
            tracker.steps_outstanding, tracker.frames_prepared = outstanding, prepared
        elif prepare:
            def counts(i):
                active.append(len(tracker.tracks))
                inactive.append(len(tracker.inactive_tracks))
            tracker.frames_prepared = um.pipelined_loop(tracker, blobs, depth=None if prepare is True else int(prepare),
                                                        on_finish=counts)
        else:
            for blob in blobs:
                tracker.step(blob)
                active.append(len(tracker.tracks))
                inactive.append(len(tracker.inactive_tracks))
    results = tracker.get_results()
    rows = np.array([[tid, f, *results[tid][f]['bbox'].tolist(), float(results[tid][f]['score']), results[tid][f]['obj_ind']]
                     for tid in sorted(results) for f in sorted(results[tid])], dtype=np.float64)
    return tracker, rows, active, inactive


def compare_wc_to_golden(name, tracker, rows, active, inactive, box_tol_px, n_frames=None):
    """Every decision of every frame: the fixture's recorded margins (smallest |score - threshold|, |IoU - NMS threshold|,
    score gap of a suppressing pair, per frame) are asserted to be WIDE first -- so nothing below is conditional on where a
    near-tie falls -- then ids / frames / source queries bit-exact, boxes and scores within tolerance."""
    z = np.load(os.path.join(GOLDEN, "tracker_%s.npz" % name))
    assert float(z["score_margin_per_frame"].min()) >= 1e-2
    assert float(z["nms_iou_margin_per_frame"].min()) >= 1e-3
    assert float(z["nms_order_margin_per_frame"].min()) >= 1e-2
    n = n_frames or len(z["active_per_frame"])
    want = z["rows"][z["rows"][:, 1] < n]
    assert z["active_per_frame"][:n].tolist() == active
    assert z["inactive_per_frame"][:n].tolist() == inactive
    if n == len(z["active_per_frame"]):
        assert int(z["num_tracks"]) == tracker.track_num
        assert int(z["num_reids"]) == tracker.num_reids
    assert rows.shape == want.shape
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], want[:, [0, 1, 7]])
    np.testing.assert_allclose(rows[:, 2:6], want[:, 2:6], atol=box_tol_px)
    np.testing.assert_allclose(rows[:, 6], want[:, 6], atol=1e-3)
    # the sequence is not static: tracks are born and terminated in every frame, and some live through all of it
    ids_per_frame = [set(want[want[:, 1] == f][:, 0].astype(int)) for f in range(n)]
    assert all(ids_per_frame[f] - ids_per_frame[f - 1] and ids_per_frame[f - 1] - ids_per_frame[f] for f in range(1, n))
    assert set.intersection(*ids_per_frame)


@pytest.mark.parametrize("name,n_frames", [("cfg2_wc", 6), ("cfg2_wc_reid", 5), ("cfg4_wc", 4)])
def test_well_conditioned_tracker_sequences_match_reference(name, n_frames, host_op):
    """The first frames of each sequence on the CPU through the host operator (the GPU suite runs all of them: 64 frames)."""
    tracker, rows, active, inactive = run_wc_tracker(name, n_frames=n_frames)
    compare_wc_to_golden(name, tracker, rows, active, inactive, box_tol_px=0.05, n_frames=n_frames)


def test_image_only_half_of_the_forward_can_run_ahead(host_op):
    """encode_frame() + forward(encoded=...) is forward(): same tensors bit for bit, with and without track queries; and a
    Tracker that runs the next frame's image-only half before it associates the current frame (step_prepare) files exactly
    the tracks of the plain step() loop.  (Round 5: the single-sequence headline no longer waits for the host.)"""
    model, post, args = um.build("cfg2_deformable_tracking", factory.build_model, config.make_args)
    model.tracking()
    img, prev, target = um.model_inputs("cfg2_deformable_tracking", args.hidden_dim)
    with torch.no_grad():
        for tgt in (None, target):
            want = model(img, None if tgt is None else [dict(t) for t in tgt], None)
            enc = model.encode_frame(img, None)
            got = model(img, None if tgt is None else [dict(t) for t in tgt], None, encoded=enc)
            for k in ("pred_logits", "pred_boxes", "hs_embed"):
                assert torch.equal(got[0][k], want[0][k])
            assert torch.equal(got[4], want[4]) and all(torch.equal(a, b) for a, b in zip(got[3], want[3]))
    plain = run_wc_tracker("cfg2_wc", n_frames=5)
    ahead = run_wc_tracker("cfg2_wc", n_frames=5, prepare=True)
    assert ahead[0].frames_prepared == 4
    np.testing.assert_array_equal(plain[1], ahead[1])
    assert plain[2] == ahead[2] and plain[3] == ahead[3]


@pytest.mark.parametrize("name,n_frames", [("cfg2_wc", 5), ("cfg4_wc", 4), ("cfg2_wc_reid", 6)])
def test_unobserved_step_loop_defers_the_association_and_files_the_same_tracks(name, n_frames, host_op):
    """Round 6: `for blob in sequence: tracker.step(blob)` -- the reference's own loop, src/track.py:130-134 -- leaves every
    frame's association outstanding until the next step() (which first enqueues ITS image-only half) or until somebody reads
    the tracker; tracks, frames, boxes and scores are those of the step-by-step loop, bit for bit, and of the reference."""
    plain = run_wc_tracker(name, n_frames=n_frames)
    lazy = run_wc_tracker(name, n_frames=n_frames, unobserved=True)
    assert lazy[0].steps_outstanding == n_frames and lazy[0].frames_prepared == n_frames - 1
    np.testing.assert_array_equal(plain[1], lazy[1])
    assert lazy[0].track_num == plain[0].track_num and lazy[0].num_reids == plain[0].num_reids
    assert lazy[0].frame_index == plain[0].frame_index == n_frames
    # state read or written from outside runs the outstanding association first
    tracker = lazy[0]
    blob = um.tracker_sequence(n_frames=n_frames + 1)[n_frames]
    with torch.no_grad():
        tracker.step(blob)
    assert tracker.__dict__["_deferred_handle"] is not None
    n_tracks = len(tracker.tracks)
    assert tracker.__dict__["_deferred_handle"] is None and tracker.frame_index == n_frames + 1 and n_tracks > 0
    with torch.no_grad():
        tracker.step(blob)
    tracker.tracks = []                      # (bench.py's re-seeding: a write must not be overwritten by the older frame)
    assert tracker.__dict__["_deferred_handle"] is None and tracker.tracks == []
    tracker.deferred = False
    with torch.no_grad():
        tracker.step(blob)
    assert tracker.__dict__.get("_deferred_handle") is None


def test_image_only_half_of_a_multi_frame_model_can_run_ahead(host_op):
    """Round 6: the multi_frame model (cfg 4).  Its first half reads the previous frame's BACKBONE features (deformable_detr.py:
    133-221 of the reference) -- results of the previous frame's first half, known when step_prepare(t + 1) runs between
    step_async(t) and step_finish(t) (Tracker._upcoming_prev_features).  Every frame but the first (which attends to itself)
    is prepared; tracks are those of the plain loop and of the reference's Tracker."""
    plain = run_wc_tracker("cfg4_wc", n_frames=4)
    ahead = run_wc_tracker("cfg4_wc", n_frames=4, prepare=True)
    assert ahead[0].frames_prepared == 3
    np.testing.assert_array_equal(plain[1], ahead[1])
    assert plain[2] == ahead[2] and plain[3] == ahead[3]
    compare_wc_to_golden("cfg4_wc", ahead[0], ahead[1], ahead[2], ahead[3], box_tol_px=0.05, n_frames=4)


def run_mask_tracker(device="cpu", frames=3, lazy_masks=False, wrap=None, prepare=False):
    """cfg-5 path: Tracker.step on the mask-head model; -> {track id: {frame: result dict}}.  wrap: detector -> detector
    (GraphedDetector); prepare: the pipelined loop -- step_async(t), step_prepare(t + 1), step_finish(t)."""
    model, post, args = um.build("cfg5_segm_tracking", factory.build_model, config.make_args,
                                 device=device)
    model.to(device)
    model.tracking()
    tracker = Tracker(wrap(model) if wrap is not None else model, post, config.tracker_cfg(), False, lazy_masks=lazy_masks)
    tracker.reset()
    blobs = um.tracker_sequence(n_frames=frames) if frames > 3 else um.tracker_sequence()[:frames]
    with torch.no_grad():
        if prepare:
            tracker.frames_prepared = um.pipelined_loop(tracker, blobs, depth=None if prepare is True else int(prepare))
        else:
            for blob in blobs:
                tracker.step(blob)
    tracker_results = tracker.get_results()
    run_mask_tracker.last_tracker = tracker
    return tracker_results


def compare_mask_tracker_to_golden(results, box_tol_px=0.05, area_tol=0.02):
    """results of run_mask_tracker against tests/golden/tracker_cfg5_masks.npz (the reference's own Tracker + mask head +
    PostProcessSegm on CPU, tests/golden/make_golden_models.py mask_tracker): ids / frames / source queries exact, boxes and
    scores within tolerance, the number of mask pixels every track owns within `area_tol` of the image (the per-pixel argmax
    over random-weight tracks near 0.5 is compared by area, not pixel by pixel)."""
    z = np.load(os.path.join(GOLDEN, "tracker_cfg5_masks.npz"))
    rows, areas = [], []
    for tid in sorted(results):
        for f in sorted(results[tid]):
            r = results[tid][f]
            rows.append([tid, f, *r['bbox'].tolist(), float(r['score']), r['obj_ind']])
            areas.append(int(np.asarray(r['mask']).sum()))
            assert list(np.asarray(r['mask']).shape) == z["mask_shape"].tolist()
    rows = np.array(rows, dtype=np.float64)
    assert rows.shape == z["rows"].shape
    np.testing.assert_array_equal(rows[:, [0, 1, 7]], z["rows"][:, [0, 1, 7]])
    np.testing.assert_allclose(rows[:, 2:6], z["rows"][:, 2:6], atol=box_tol_px)
    np.testing.assert_allclose(rows[:, 6], z["rows"][:, 6], atol=1e-3)
    n_px = int(np.prod(z["mask_shape"]))
    assert np.abs(np.array(areas) - z["mask_areas"]).max() <= area_tol * n_px, (areas, z["mask_areas"].tolist())


def test_tracker_with_mask_head_matches_reference(host_op):
    compare_mask_tracker_to_golden(run_mask_tracker())


def test_tracker_with_mask_head_produces_per_track_masks(host_op):
    results = run_mask_tracker()
    assert results
    per_frame = {}
    for tid, frames in results.items():
        for f, r in frames.items():
            assert r['mask'].shape == um.TRACKER_ORIG and r['mask'].dtype == np.bool_
            per_frame.setdefault(f, []).append(r['mask'])
    for f, masks in per_frame.items():   # a pixel belongs to at most one track (tracker.py:521-532)
        assert np.stack(masks).sum(0).max() <= 1
    assert any(m.any() for masks in per_frame.values() for m in masks)


def test_state_dict_layout_of_cfg2_model():
    model, _, _ = factory.build_model(config.make_args('deformable', 'tracking', 'mot17',
                                                       device='cpu'))
    keys = list(model.state_dict().keys())
    assert len(keys) == 597
    for k in ("transformer.level_embed", "transformer.reference_points.weight",
              "transformer.encoder.layers.0.self_attn.sampling_offsets.weight",
              "transformer.decoder.layers.5.cross_attn.output_proj.bias",
              "transformer.decoder.layers.0.self_attn.in_proj_weight",
              "class_embed.5.bias", "bbox_embed.0.layers.2.weight", "query_embed.weight",
              "input_proj.3.0.weight", "input_proj.0.1.bias", "backbone.0.body.conv1.weight",
              "backbone.0.body.layer4.2.bn3.running_var",
              "backbone.0.body.layer2.0.downsample.0.weight"):
        assert k in keys, k
    assert model.state_dict()["query_embed.weight"].shape == (300, 512)
    assert model.num_queries == 300 and model.overflow_boxes and model.hidden_dim == 256
    assert sum(p.numel() for p in model.parameters()) == 40740178


def test_product_model_runs_on_cpu_tensors_through_the_host_operator():
    """CPU tensors take the library's host entry points (the reference raises "Not implemented on the CPU" here,
    ms_deform_attn.h:27,48; SURVEY 8(b): a real CPU path instead).  Device tensors never do: tests/test_msda_host.py."""
    model, _, _ = factory.build_model(config.make_args('deformable', 'tracking', 'mot17',
                                                       device='cpu'))
    model.tracking()
    with torch.no_grad():
        out, *_ = model(torch.zeros(1, 3, 64, 64), None, None)
    assert out['pred_boxes'].shape == (1, 300, 4) and bool(torch.isfinite(out['pred_logits']).all())


# ------------------------------------------------------------------ one training step (cfg 3 path)
def run_train_step(device="cpu", masks=False):
    model, criterion, args = um.build_train(factory.build_model, config.make_args, device=device,
                                            masks=masks)
    model.to(device)
    criterion.to(device)
    samples, targets = um.train_batch(device=device, masks=masks)
    return um.train_step(model, criterion, samples, targets)


def compare_train_to_golden(loss_dict, total, grads, rtol, fixture="train_cfg3_small.npz"):
    z = np.load(os.path.join(GOLDEN, fixture))
    assert sorted(loss_dict) == z["loss_keys"].tolist()
    got = np.array([loss_dict[k] for k in sorted(loss_dict)])
    np.testing.assert_allclose(got, z["loss_vals"], rtol=rtol, atol=rtol)
    assert abs(total - float(z["total"])) <= rtol * abs(float(z["total"]))
    assert len(grads) == int(z["num_grads"])
    gn = np.array([grads[k] for k in z["grad_keys"].tolist()])
    np.testing.assert_allclose(gn, z["grad_norms"], rtol=max(rtol, 2e-3), atol=1e-6)


def test_training_step_matches_reference(host_op):
    """forward (padded batch, track-query augmentation, prev-frame pass) + SetCriterion + backward
    through MSDeformAttnFunction.backward; losses and gradient norms vs the reference on CPU."""
    loss_dict, total, grads = run_train_step()
    compare_train_to_golden(loss_dict, total, grads, rtol=2e-4)


def test_training_step_with_mask_head_matches_reference(host_op):
    """cfg-5 training path: + loss_mask (focal) / loss_dice (detr.py:330-358) and the gradients of
    MHAttentionMap / MaskHeadSmallConv (incl. the split first convolution)."""
    loss_dict, total, grads = run_train_step(masks=True)
    assert "loss_mask" in loss_dict and "loss_dice" in loss_dict
    compare_train_to_golden(loss_dict, total, grads, rtol=2e-4, fixture="train_cfg5_masks_small.npz")


def test_criterion_layers_at_once_equals_the_loop_over_the_layers(host_op):
    """Round 6: SetCriterion computes the class / cardinality / box losses of the final + five auxiliary decoder layers in one chain of
    kernels with a leading layer dimension (criterion._layers_at_once) instead of looping (models/detr.py:266-289 of the reference):
    the same values to fp32 round-off, the same gradients."""
    from trackformer_amd import criterion as crit_mod
    model, criterion, args = um.build_train(factory.build_model, config.make_args)
    samples, targets = um.train_batch()
    got = {}
    for on in (True, False):
        prev = crit_mod.set_layers_at_once(on)
        try:
            torch.manual_seed(7)
            model.zero_grad()
            outputs, tg, *_ = model(samples, [dict(t, prev_target=dict(t['prev_target'])) for t in targets])
            loss_dict = criterion(outputs, tg)
            total = sum(loss_dict[k] * criterion.weight_dict[k] for k in loss_dict if k in criterion.weight_dict)
            total.backward()
            got[on] = ({k: float(v) for k, v in loss_dict.items()},
                       {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None})
        finally:
            crit_mod.set_layers_at_once(prev)
    assert sorted(got[True][0]) == sorted(got[False][0]) and len(got[True][0]) >= 13   # (the small model: 3 decoder layers; BASELINE: 6 -> 25)
    for k, v in got[False][0].items():
        assert abs(got[True][0][k] - v) <= 2e-6 * max(1.0, abs(v)), (k, got[True][0][k], v)
    assert sorted(got[True][1]) == sorted(got[False][1])
    for n, v in got[False][1].items():
        assert abs(got[True][1][n] - v) <= 1e-4 * max(abs(v), 1e-6 * max(got[False][1].values())), n


def test_engine_train_step_reproduces_reference_loss_and_updates_weights(host_op):
    """engine.train_step (engine.py:119-158 body) + build_optimizer (train.py:93-120 groups)."""
    from trackformer_amd import engine
    model, criterion, args = um.build_train(factory.build_model, config.make_args)
    optimizer, scheduler = engine.build_optimizer(model, args)
    lrs = [g['lr'] for g in optimizer.param_groups]
    assert lrs == [args.lr, args.lr_backbone, args.lr * args.lr_linear_proj_mult]
    names = dict(model.named_parameters())
    in_proj_group = {id(p) for p in optimizer.param_groups[2]['params']}
    assert id(names['transformer.reference_points.weight']) in in_proj_group
    assert id(names['transformer.encoder.layers.0.self_attn.sampling_offsets.weight']) in in_proj_group
    assert all(id(p) in {id(q) for q in optimizer.param_groups[1]['params']}
               for n, p in names.items() if n.startswith('backbone.0') and p.requires_grad)
    before = names['class_embed.0.weight'].detach().clone()
    samples, targets = um.train_batch()
    model.train()
    criterion.train()
    torch.manual_seed(7)
    loss, loss_dict = engine.train_step(model, criterion, optimizer, samples, targets,
                                        clip_max_norm=args.clip_max_norm)
    z = np.load(os.path.join(GOLDEN, "train_cfg3_small.npz"))
    assert abs(float(loss) - float(z["total"])) < 1e-3 * abs(float(z["total"]))
    assert not torch.equal(before, names['class_embed.0.weight'].detach())


def test_lazy_mask_head_gives_the_same_tracks_and_masks(host_op):
    """Tracker(lazy_masks=True) (opt-in): the mask head runs only for the queries whose masks the step keeps
    (DETRSegmBase.mask_rows on the frame's MaskContext) -- same track ids, boxes and scores, and the same masks up to the
    convolution library's batch-size dependent summation order (probabilities next to 0.5 may flip a pixel)."""
    full = run_mask_tracker()
    lazy = run_mask_tracker(lazy_masks=True)
    assert sorted(full) == sorted(lazy)
    union_full, union_lazy = {}, {}
    for tid in full:
        assert sorted(full[tid]) == sorted(lazy[tid])
        for f in full[tid]:
            a, b = full[tid][f], lazy[tid][f]
            np.testing.assert_array_equal(a['bbox'], b['bbox'])
            np.testing.assert_array_equal(a['score'], b['score'])
            assert a['obj_ind'] == b['obj_ind'] and a['mask'].shape == b['mask'].shape
            union_full[f] = union_full.get(f, 0) | a['mask']
            union_lazy[f] = union_lazy.get(f, 0) | b['mask']
    # the random-init model's tracks produce near-identical probability maps, so WHICH track wins a pixel (the argmax of
    # tracker.py:521-532) flips with the last bits; that a pixel is covered (max probability > 0.5) does not
    n_px = sum(u.size for u in union_full.values())
    n_diff = sum(int((union_full[f] != union_lazy[f]).sum()) for f in union_full)
    assert n_px > 0 and n_diff <= 2e-3 * n_px, (n_diff, n_px)


def test_mask_rows_equal_the_rows_of_the_full_head(host_op):
    """DETRSegmBase.mask_rows(ctx, hs[:, rows]) == forward()'s pred_masks[:, rows]."""
    model, post, args = um.build("cfg5_segm_tracking", factory.build_model, config.make_args)
    model.tracking()
    img, prev, target = um.model_inputs("cfg5_segm_tracking", args.hidden_dim)
    with torch.no_grad():
        out, *_ = model(img, target, None)
        model.lazy_masks = True
        try:
            lazy_out, *_ = model(img, target, None)
        finally:
            model.lazy_masks = False
        assert 'pred_masks' not in lazy_out and 'mask_context' in lazy_out
        assert torch.equal(lazy_out['pred_logits'], out['pred_logits'])
        rows = torch.tensor([0, 3, 7, out['pred_masks'].shape[1] - 1])
        got = model.mask_rows(lazy_out['mask_context'], lazy_out['hs_embed'][:, rows])
    scale = float(out['pred_masks'].abs().max())
    assert torch.allclose(got, out['pred_masks'][:, rows], atol=1e-5 * max(1.0, scale))


def test_mask_postprocess_of_selected_queries_equals_rows_of_the_full_result():
    """Tracker.step resizes only the masks its surviving tracks reference (tracker._resolve_masks); PostProcessSegm
    on a subset of the queries must give exactly the rows it gives on all of them (detr_segmentation.py:297-334)."""
    from trackformer_amd.detr_segmentation import PostProcessSegm
    g = torch.Generator().manual_seed(0)
    outputs = {"pred_masks": torch.randn(1, 12, 24, 32, generator=g)}
    post = PostProcessSegm()
    sizes, orig = torch.tensor([[90, 120]]), torch.tensor([[180, 250]])
    full = post([{}], outputs, orig, sizes, return_probs=True)[0]["masks"]
    keep = torch.zeros(12, dtype=torch.bool)
    keep[[1, 4, 5, 11]] = True
    part = post([{}], outputs, orig, sizes, return_probs=True, results_mask=[keep])[0]["masks"]
    # (the bilinear resize of a 4-channel and a 12-channel tensor may round differently in the last bit)
    assert part.shape[0] == 4 and torch.allclose(part, full[keep], atol=1e-6, rtol=0)


def test_filler_track_queries_do_not_change_the_real_queries(host_op):
    """GraphedDetector rounds the track-query count up with filler queries (`track_query_filler`, masked as keys of the
    decoder's query self-attention) and drops their rows: on the CPU, through nn.MultiheadAttention's key_padding_mask,
    the real rows must equal the unpadded forward."""
    model, post, args = um.build("cfg2_deformable_tracking", factory.build_model, config.make_args)
    model.tracking()
    img, prev, target = um.model_inputs("cfg2_deformable_tracking", args.hidden_dim)
    n = target[0]['track_query_hs_embeds'].shape[0]
    extra = 16 - n
    padded = [dict(target[0],
                   track_query_hs_embeds=torch.cat([target[0]['track_query_hs_embeds'], torch.zeros(extra, args.hidden_dim)]),
                   track_query_boxes=torch.cat([target[0]['track_query_boxes'],
                                                torch.tensor([0.5, 0.5, 0.1, 0.1]).expand(extra, 4)]),
                   track_query_filler=torch.arange(16) >= n)]
    with torch.no_grad():
        ref, *_ = model(img, [dict(target[0])], None)
        out, *_ = model(img, padded, None)
    keep = torch.cat([torch.arange(n), torch.arange(16, 16 + model.num_queries)])
    for k in ('pred_logits', 'pred_boxes', 'hs_embed'):
        assert out[k].shape[1] == 16 + model.num_queries
        assert torch.allclose(out[k][:, keep], ref[k], atol=2e-5, rtol=1e-5), k
