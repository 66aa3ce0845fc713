"""Round 6: the schedule of the prepared image-only halves (GraphedDetector WIDE / NARROW, Tracker.look_ahead, the look-ahead loop of
dist_utils.track_sequences) and the stream placement (runtime.pool_stream / bind_streams, dist_utils.sequence_stream).  The host
logic runs on CPU; what needs streams is marked gpu."""
import pytest
import torch

from trackformer_amd import dist_utils, runtime
from trackformer_amd.graphed import GraphedDetector


class _Plain:
    multi_frame_attention = False


class _MultiFrame:
    multi_frame_attention = True


class _Masks:
    multi_frame_attention = False

    def lazy_masks_active(self):
        return True


def test_schedule_by_model_family(monkeypatch):
    for k in ("TF_GRAPH_SLOTS", "TF_GRAPH_LOOKAHEAD", "TF_GRAPH_SIDE_STREAMS", "GPU_MAX_HW_QUEUES"):
        monkeypatch.delenv(k, raising=False)
    wide = GraphedDetector(_Plain())
    assert (wide.SLOTS, wide.LOOKAHEAD, wide.SIDE_STREAMS) == GraphedDetector.WIDE == (4, 2, 2)
    for model in (_MultiFrame(), _Masks()):
        narrow = GraphedDetector(model)
        assert (narrow.SLOTS, narrow.LOOKAHEAD, narrow.SIDE_STREAMS) == GraphedDetector.NARROW == (2, 1, 1)
    lanes = GraphedDetector(_Plain(), lanes=3, lane=2)
    assert (lanes.SLOTS, lanes.LOOKAHEAD, lanes.SIDE_STREAMS) == (4, 2, 2) and (lanes._lanes, lanes._lane) == (3, 2)
    lanes.set_lanes(1)
    assert (lanes._lanes, lanes._lane) == (1, 0)
    monkeypatch.setenv("TF_GRAPH_LOOKAHEAD", "3")
    monkeypatch.setenv("TF_GRAPH_SLOTS", "2")
    forced = GraphedDetector(_Plain())
    assert forced.LOOKAHEAD == 3 and forced.SLOTS == 4   # (never fewer slots than frames in flight: decoded + prepared)


def test_schedule_falls_back_when_the_runtime_has_another_number_of_hardware_queues(monkeypatch):
    """The placement tables were measured with the HIP runtime's default of 4 hardware queues: with another GPU_MAX_HW_QUEUES every
    model keeps round 5's schedule (runtime.placement_tuned)."""
    for k in ("TF_GRAPH_SLOTS", "TF_GRAPH_LOOKAHEAD", "TF_GRAPH_SIDE_STREAMS"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert not runtime.placement_tuned()
    det = GraphedDetector(_Plain())
    assert (det.SLOTS, det.LOOKAHEAD, det.SIDE_STREAMS) == GraphedDetector.NARROW
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    assert runtime.placement_tuned()
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    assert runtime.placement_tuned() and GraphedDetector(_Plain()).SLOTS == 4


def test_slots_rotate_past_the_decoded_and_the_prepared_ones():
    det = GraphedDetector(_Plain())
    img = torch.zeros(1, 3, 8, 8)
    order = []
    for _ in range(6):                       # steady state of the pipelined loop: decode one, prepare one
        i = det._next_slot()
        assert i != det._slot and i not in {e[1] for e in det._fifo}
        det._fifo.append((img, i, 0))
        order.append(i)
        if len(det._fifo) > det.LOOKAHEAD:
            det._slot = det._fifo.pop(0)[1]
    assert order == [1, 2, 3, 0, 1, 2]
    narrow = GraphedDetector(_Masks())
    seen = []
    for _ in range(4):                       # round 5's alternation: the slot that is not being decoded
        i = narrow._next_slot()
        assert i != narrow._slot
        seen.append(i)
        narrow._slot = i
    assert seen == [1, 0, 1, 0]


def test_preparations_are_consumed_in_order():
    det = GraphedDetector(_Plain())
    a, b, c = (torch.zeros(1, 3, 8, 8) for _ in range(3))
    key = (tuple(a.shape), a.device)
    det._enc[key] = [{"generation": 0}, {"generation": 1}, {"generation": 2}, {"generation": 7}]
    det._fifo = [(a, 1, 1), (b, 2, 2)]
    assert det._take_prepared(a, None) == 1 and [e[1] for e in det._fifo] == [2]
    assert det._take_prepared(b, None) == 2 and det._fifo == []
    det._fifo = [(a, 1, 1), (b, 2, 2)]
    assert det._take_prepared(b, None) == 2 and det._fifo == []          # the later one first: the earlier one is dropped
    det._fifo = [(a, 1, 1), (b, 2, 2)]
    assert det._take_prepared(c, None) is None and det._fifo == []       # another image: all dropped
    det._fifo = [(a, 3, 6)]
    assert det._take_prepared(a, None) is None                           # the slot has been prepared again since (generation)


class _RecordingTracker:
    """What track_sequences needs of a tracker; files the order of its calls."""
    look_ahead = 2

    def __init__(self, log, accept=lambda blob: True):
        self.log, self.accept = log, accept

    def reset(self):
        self.log.append(("reset",))

    def step_prepare(self, blob):
        ok = self.accept(blob)
        self.log.append(("prepare", blob, ok))
        return ok

    def step_async(self, blob):
        self.log.append(("async", blob))
        return blob

    def step_finish(self, handle):
        self.log.append(("finish", handle))

    def get_results(self):
        return {"frames": [e[1] for e in self.log if e[0] == "finish"]}


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_track_sequences_prepares_look_ahead_frames_once_and_in_order(depth):
    log = []

    def make(device):
        t = _RecordingTracker(log)
        t.look_ahead = depth
        return t
    frames = ["s0f%d" % i for i in range(6)]
    out = dist_utils.track_sequences(make, [frames, ["s1f0", "s1f1"]], "cpu")
    assert out[0]["frames"][:6] == frames
    done_async = [e[1] for e in log if e[0] == "async"]
    assert done_async == frames + ["s1f0", "s1f1"]
    prepared = [e[1] for e in log if e[0] == "prepare"]
    assert len(prepared) == len(set(prepared))                            # nothing is prepared twice
    assert prepared == frames[1:] + ["s1f1"]                              # everything but a sequence's first frame, in order
    for blob in prepared:                                                 # ... before its own step, at most `depth` frames ahead
        at = log.index(("prepare", blob, True))
        assert at < log.index(("async", blob))
        behind = [e for e in log[:at] if e[0] == "async" and e[1].startswith(blob[:2])]
        assert int(blob[3:]) - int(behind[-1][1][3:]) <= depth


def test_track_sequences_retries_a_refused_preparation():
    log, refused = [], {"s0f2"}

    def accept(blob):
        if blob in refused:
            refused.discard(blob)
            return False
        return True
    dist_utils.track_sequences(lambda device: _RecordingTracker(log, accept), [["s0f%d" % i for i in range(5)]], "cpu")
    prepares = [(e[1], e[2]) for e in log if e[0] == "prepare"]
    assert ("s0f2", False) in prepares and ("s0f2", True) in prepares
    assert [b for b, ok in prepares if ok] == ["s0f1", "s0f2", "s0f3", "s0f4"]   # still in order: nothing overtakes a refused frame


@pytest.mark.gpu
def test_pool_streams_are_picked_by_index_and_bound_once():
    dev = torch.device("cuda", 0)
    runtime.bind_streams(dev)
    runtime.bind_streams(dev)
    for priority in (0, -1):
        for index in (0, 1, 5, 31):
            s = runtime.pool_stream(dev, index, priority)
            assert int(s.stream_id) >> 5 == index and s.priority == priority
            assert runtime.pool_stream(dev, index, priority).stream_id == s.stream_id
    one = dist_utils.sequence_stream(dev)
    assert one.priority == -1 and int(one.stream_id) >> 5 == 0
    lanes = [dist_utils.sequence_stream(dev, 3, k) for k in range(3)]
    assert [s.priority for s in lanes] == [0, 0, 0]
    assert [int(s.stream_id) >> 5 for s in lanes] == list(dist_utils.LANE_MAINS[:3])
    det = GraphedDetector(_Plain())
    assert [int(det._side_stream(dev, i).stream_id) >> 5 for i in range(4)] == [1, 5, 1, 5]
    det = GraphedDetector(_Plain(), lanes=3, lane=1)
    first = dist_utils.LANE_SIDES[1]
    assert [int(det._side_stream(dev, i).stream_id) >> 5 for i in range(2)] == [first, (first + 16) % 32]
