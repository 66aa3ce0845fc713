"""CPU: the host-side structures of the tracker that replaced per-track tensors (trackformer_amd/tracker.py): the embedding
history as (frame embeddings, row) references, the one-launch track-query build, obj_ind as an int with the reference's [1]
tensor on demand, the (positions, scores) arrays carried through the NMS passes."""
import torch

from trackformer_amd.tracker import Track, Tracker, _HsHistory


def test_embedding_history_reads_like_the_reference_list():
    frame_a, frame_b = torch.randn(7, 16), torch.randn(5, 16)
    h = _HsHistory((frame_a, 3))
    h.append(frame_b[1])                # a plain [C] tensor, as reference-style callers append
    h.append_row(frame_b, 4)
    assert len(h) == 3
    assert torch.equal(h[0], frame_a[3]) and torch.equal(h[1], frame_b[1]) and torch.equal(h[-1], frame_b[4])
    assert [tuple(t.shape) for t in h] == [(16,)] * 3
    assert all(torch.equal(a, b) for a, b in zip(h[1:], [frame_b[1], frame_b[4]]))
    t = Track(torch.zeros(4), torch.tensor(0.5), 9, frame_a[2], 5)
    assert torch.equal(t.hs_embed[-1], frame_a[2]) and len(t.hs_embed) == 1


def test_track_queries_from_one_frame_are_one_index_select_and_equal_the_stack():
    frame, other = torch.randn(40, 8), torch.randn(6, 8)
    rows = [5, 0, 39, 17, 5]
    tracks = [Track(torch.zeros(4), torch.tensor(0.9), i, (frame, r), i) for i, r in enumerate(rows)]
    want = torch.stack([frame[r] for r in rows])
    calls = []
    orig = torch.Tensor.index_select

    def counting(self, *a, **k):
        calls.append(1)
        return orig(self, *a, **k)
    torch.Tensor.index_select = counting
    try:
        got = Tracker._track_query_embeds(tracks, torch.device("cpu"))
    finally:
        torch.Tensor.index_select = orig
    assert torch.equal(got, want) and len(calls) == 1
    # a track pointing into another frame, or holding a plain tensor: the stack of the reference, same values
    tracks[2].hs_embed.append_row(other, 3)
    tracks[4].hs_embed.append(other[1].clone())
    want2 = want.clone()
    want2[2], want2[4] = other[3], other[1]
    assert torch.equal(Tracker._track_query_embeds(tracks, torch.device("cpu")), want2)


def test_obj_ind_is_the_reference_tensor_on_demand():
    t = Track(torch.zeros(4), torch.tensor(0.5), 0, torch.zeros(8), 17)
    assert t.obj_index == 17 and t._obj_ind is None            # nothing materialised for the per-frame results
    assert t.obj_ind.dtype == torch.int64 and t.obj_ind.tolist() == [17]
    u = Track(torch.zeros(4), torch.tensor(0.5), 1, torch.zeros(8), torch.tensor([23]))
    assert u.obj_index == 23 and u.obj_ind.tolist() == [23]
    u.obj_ind = torch.tensor([4])
    assert u.obj_index == 4
    u.gt_id = 3
    u.anything_else = "callers may hang attributes on a track"   # __dict__ is still there


def test_keep_after_nms_filters_list_and_arrays_together():
    class _Det(torch.nn.Module):
        num_queries = 3
    tr = Tracker.__new__(Tracker)
    tr._logger = lambda *a: None
    boxes = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [50., 50., 60., 60.], [0., 0., 10.5, 10.]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    tr.tracks = [Track(b, s, i, torch.zeros(2), i) for i, (b, s) in enumerate(zip(boxes, scores))]
    pos, sc = tr._keep_after_nms(boxes, scores, scores, 0.5, "track_nms_thresh")
    assert [t.id for t in tr.tracks] == [2, 3]                   # 0 and 1 overlap the better-scored 3; list order kept
    assert torch.equal(pos, boxes[[2, 3]]) and torch.equal(sc, scores[[2, 3]])
    pos2, sc2 = tr._keep_after_nms(pos, sc, sc, 0.5, "track_nms_thresh")
    assert pos2 is pos and [t.id for t in tr.tracks] == [2, 3]   # nothing suppressed: nothing rebuilt


def test_positions_and_scores_as_row_references_read_like_tensors():
    from trackformer_amd.tracker import _LastPos, _gather_rows
    boxes, scores = torch.rand(6, 4), torch.rand(6)
    t = Track((boxes, 2), (scores, 2), 0, torch.zeros(8), 1)
    assert type(t._pos) is tuple                                   # nothing built yet
    assert torch.equal(t.pos, boxes[2]) and torch.equal(t.score, scores[2]) and t.score.dim() == 0
    assert torch.is_tensor(t._pos)                                 # built once, then kept
    assert isinstance(t.last_pos, _LastPos) and torch.equal(t.last_pos[-1], boxes[2]) and len(t.last_pos) == 1
    t.pos = boxes[4]
    t.last_pos.append(t._pos)
    t.last_pos.append((boxes, 5))
    assert [p.tolist() for p in t.last_pos] == [boxes[2].tolist(), boxes[4].tolist(), boxes[5].tolist()]
    assert torch.equal(t.last_pos.pop(), boxes[5]) and torch.equal(t.last_pos.popleft(), boxes[2])
    t.reset_last_pos()
    assert len(t.last_pos) == 1 and torch.equal(t.last_pos[0], boxes[4])
    assert t.has_positive_area() == bool(boxes[4, 2] > boxes[4, 0] and boxes[4, 3] > boxes[4, 1])


def test_gather_rows_equals_the_stack_for_every_mix_of_sources():
    from trackformer_amd.tracker import _gather_rows
    g = torch.Generator().manual_seed(0)
    srcs = [torch.rand(9, 4, generator=g) for _ in range(6)]
    plain = [torch.rand(4, generator=g) for _ in range(3)]
    cases = {
        "one source": [(srcs[0], r) for r in (3, 0, 8, 3)],
        "two sources": [(srcs[0], 1), (srcs[1], 7), (srcs[0], 2), (srcs[1], 0)],
        "sources and tensors": [plain[0], (srcs[2], 4), plain[1], (srcs[0], 4), (srcs[2], 5), plain[2]],
        "tensors only": plain,
        "many sources": [(s, i) for i, s in enumerate(srcs)],
        "tensor first": [plain[0], (srcs[0], 1), (srcs[0], 2)],
    }
    for name, items in cases.items():
        want = torch.stack([it[0][it[1]] if type(it) is tuple else it for it in items])
        assert torch.equal(_gather_rows(items), want), name
    scores = torch.rand(5, generator=g)
    assert torch.equal(_gather_rows([(scores, 4), (scores, 0)]), scores[[4, 0]])     # 1-d sources: the scores


def test_rank_cpu_shares_follow_the_gpu_numa_map(monkeypatch):
    """dist_utils._cpu_share: ranks whose GPUs share a NUMA node split that node's CPUs by their order among themselves, for any
    GPU -> node mapping (contiguous, interleaved, uneven); an unknown node anywhere makes EVERY rank fall back to the even split,
    so two ranks never end up on the same CPUs."""
    from trackformer_amd import dist_utils as du
    node_cpus = {0: set(range(0, 16)), 1: set(range(16, 32))}
    monkeypatch.setattr(du, "_node_cpus", lambda n: node_cpus.get(n))
    allowed = list(range(32))
    for node_of in ([0, 0, 0, 0, 1, 1, 1, 1], [0, 1, 0, 1, 0, 1, 0, 1], [0, 0, 0, 0, 0, 0, 1, 1], [0, None, 0, 1, 1, 1, 0, 0]):
        shares = [du._cpu_share(r, 8, allowed, node_of) for r in range(8)]
        flat = [c for s in shares for c in s]
        assert len(flat) == len(set(flat)) and all(shares), (node_of, shares)          # disjoint, nobody empty
        if None not in node_of:
            for r, s in enumerate(shares):
                assert set(s) <= node_cpus[node_of[r]], (node_of, r, s)                # on the rank's own node
            assert len(set(flat)) == 32                                                # every CPU of both nodes is used
        else:
            assert shares == [allowed[4 * r:4 * r + 4] for r in range(8)]              # everyone fell back together
    # a cpuset that covers node 0 only (a container): node 1's ranks have no CPU of their own node -- the fallback is decided on the
    # whole table, so EVERY rank takes the even split (deciding per rank let node 0's ranks keep node slices that overlapped it)
    allowed = list(range(16))
    for node_of in ([0, 0, 0, 0, 1, 1, 1, 1], [0, 1, 0, 1, 0, 1, 0, 1]):
        shares = [du._cpu_share(r, 8, allowed, node_of) for r in range(8)]
        assert shares == [allowed[2 * r:2 * r + 2] for r in range(8)], (node_of, shares)
    # node 1 has CPUs, but fewer than ranks: the same
    allowed = list(range(16)) + [16, 17]
    shares = [du._cpu_share(r, 8, allowed, [0, 0, 0, 0, 1, 1, 1, 1]) for r in range(8)]
    flat = [c for s in shares for c in s]
    assert len(flat) == len(set(flat)) and all(shares) and shares[0] == allowed[0:2]


def test_host_nms_orders_nan_scores_first_and_keeps_input_order_among_ties():
    """tf_nms_host_f32: stable descending order with NaN scores in front (torch.sort's placement); ties by input order."""
    import ctypes

    import numpy as np

    from trackformer_amd import _cabi
    try:
        lib = _cabi.lib()
    except RuntimeError:
        import pytest
        pytest.skip("libtf_msda.so not built")
    boxes = np.array([[0, 0, 10, 10], [100, 100, 110, 110], [0, 0, 10, 10.5], [200, 200, 210, 210], [100, 100, 110, 111]], np.float32)
    scores = np.array([0.5, float("nan"), float("inf"), 0.5, float("inf")], np.float32)
    keep = np.zeros(5, np.int64)
    n_keep = ctypes.c_int(0)
    rc = lib.tf_nms_host_f32(boxes.ctypes.data, scores.ctypes.data, 5, ctypes.c_float(0.5), keep.ctypes.data, ctypes.byref(n_keep))
    assert rc == 0
    # order: NaN (1), then the two +inf in input order (2, 4), then the two 0.5 (0, 3); 4 overlaps 1 (IoU 0.91) and 0 overlaps 2
    assert keep[:n_keep.value].tolist() == [1, 2, 3]
