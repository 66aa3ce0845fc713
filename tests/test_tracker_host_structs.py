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
