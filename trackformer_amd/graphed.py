"""HIP-graph replay of the detector forward for fixed-shape tracking steps.

At batch 1 the per-frame forward is ~900 small-to-medium kernels (53 convolutions, 12 attention
layers of ~15 launches each, heads); launched eagerly from Python the host cannot feed an MI355X
fast enough.  `GraphedDetector` captures `model(img, [track queries], prev_features)` once per
(image shape, number of track queries, previous frame present) into a HIP graph with static
input / output buffers and replays it for every later frame of that shape: one graph launch instead of
~900 kernel launches, same kernels, same results.

It is a drop-in for the `obj_detector` argument of Tracker: attribute access (num_queries,
overflow_boxes, parameters(), ...) is forwarded to the wrapped model.  Calls it cannot replay --
training mode, gradients enabled, host images, batches -- run the wrapped model eagerly.

Multi-frame attention (cfgs/train_multi_frame.yaml; deformable_detr.py:161-221): the previous
frame's backbone features are an INPUT of the forward.  They live in static buffers owned by the graph
entry; the captured graph ends with a copy of the new frame's features into those buffers, and the
buffers are what the call returns as `features` -- so the tracker hands them back on the next frame
and no copy is needed then (a caller that passes anything else gets a copy-in before the replay).

Aliasing contract: `out['pred_logits' | 'pred_boxes' | 'hs_embed']` are cloned (the tracker keeps them
across frames); `features` (see above), `memory`, `hs` and `out['aux_outputs']` alias static buffers and are valid
until the next call with the same key -- or, for the two-graph models, until the SECOND following prepare() (the image-only
half writes `features` / `memory` into one of two alternating slots, whatever the key).

Debug modes and routes of the split product (fused.audit_activation_range, fused.set_check_finite, fused.route_six_terms):
graph replays call none of fused's Python wrappers, so while an audit or the finite check is active the wrapper runs the
model EAGERLY, and graphs captured under another fused.route_epoch() are dropped (the routed kernels are baked into a graph
at capture time: a route added later forces a new capture).

Bucketed track-query count: in real tracking the number of track queries changes from frame to frame, and every
count would be its own graph.  The wrapper therefore rounds the count up to a multiple of `bucket` (16) with FILLER
track queries (zero embedding, a fixed box) that are masked out as keys of the decoder's query self-attention
(`track_query_filler` -> `filler_key_mask`, deformable_transformer.py) -- their attention weight is exactly zero, so
the real queries see what they would see without them (up to the summation order of the attention) -- and drops the
fillers' rows from the outputs.  Everything else in the decoder is per query.  `bucket=1` switches it off.

Two graphs (round 5; with a mask head since round 6).  For the single-frame models the forward is captured in two halves: A = backbone +
input projections + encoder, which depends on the image only, and B = decoder + heads, which also takes the track queries.
`prepare(img)` replays A for a frame whose track queries are not known yet -- Tracker.step_prepare calls it for frame
t + 1 before it associates frame t, so that the GPU never waits for the host in a single sequence -- and the following
call with the tensor it returned replays B alone.  Without prepare() a call replays A and B back to back: same kernels as one
graph.  prepare() runs A on a SIDE stream of the wrapper and into one of TWO sets of static buffers (used alternately), so that
A of frame t + 1 (53 convolutions and the encoder: large launches) shares the GPU with B of frame t (the decoder: ~150 launches of
a few microseconds on 300-400 rows, which leave most of the chip idle) instead of queueing behind it; B waits for its A by an
event, A waits by an event for the B that last read its buffers.  The arithmetic of a frame is unchanged.

Threads: bench.py and INTEGRATION.md run one tracker thread (own HIP stream) per sequence against one
shared model.  Captures are serialised by a process-wide lock and run in `thread_local` capture mode
with a private memory pool, so other threads may keep replaying / running eagerly meanwhile.  The
number of live graphs per wrapper is bounded (`max_graphs`, least-recently-used eviction): in real
tracking the number of track queries changes from frame to frame, every new count is a new key.
"""
import threading
from collections import OrderedDict

import torch

_CAPTURE_LOCK = threading.Lock()


class GraphedDetector:
    # `features` of a multi-frame replay are the entry's static buffers (module docstring): a caller that keeps more than
    # the latest set (Tracker with prev_frame_dist > 1) has to clone them -- Tracker.step checks this attribute
    features_alias_static_buffers = True
    # The schedule of the prepared image-only halves (round 6; instance attributes, chosen in __init__ by the model):
    #   SLOTS         sets of image-only buffers, handed out round-robin (slot i runs -- and was captured -- on side stream i % SIDE_STREAMS)
    #   LOOKAHEAD     frames that may be prepared and not yet decoded (SLOTS >= LOOKAHEAD + 1: the slot being decoded, the prepared
    #                 ones and the next one to fill are distinct)
    #   SIDE_STREAMS  streams the halves run on: streams SIDE_FIRST, SIDE_FIRST + SIDE_SPACING, ... of PyTorch's pool (_side_stream)
    # WIDE (4 / 2 / 2, spaced) for single-frame models without a mask head: the image-only halves of frames t + 1 and t + 2 are in
    # flight together while frame t is decoded -- one such half leaves the chip part idle (small grids at the end of the backbone,
    # ~120 dependent launches), two of them share it: 2.10 instead of 2.67 ms per half (tools/experiments/two_image_halves.py).
    # cfg 2 on MI355X with the sequence on a high-priority stream (dist_utils.sequence_stream): 348 -> 375 frames/s, host frames
    # 326 -> 371, the reference's own step() loop 325 -> 358 (tools/gpu_runs/gpu_r06_48.sh).  WHICH streams decides
    # (runtime.pool_stream): side streams (1, 5) or (0, 4) or (3, 7) of the pool next to high-priority stream 0: 378 frames/s,
    # (2, 6): 234, (1, 2): 350 (tools/experiments/stream_queue_map.py, profiles/r06_stream_queue_map.txt).
    # NARROW (2 / 1 / 1: the schedule of round 5) for the others: a mask-head model loses 10 % with four slots (cfg 5: 118 -> 105
    # frames/s, three slots 110 -- each slot holds the full-resolution backbone features the mask head reads, four of them no longer
    # stay in the 256 MB Infinity Cache; gpu_r06_49.sh); a multi-frame model's next-but-one frame needs features no call has
    # returned yet, and four slots alone cost it 1.5 % (cfg 4: 180.5 -> 177.7).
    WIDE = (4, 2, 2)
    NARROW = (2, 1, 1)

    def __init__(self, model, max_graphs=32, bucket=16, lanes=1, lane=0):
        """lanes / lane: how many sequences the process tracks at once on this device (dist_utils.track_sequences' interleave; each
        with its own wrapper) and which of them this wrapper serves: decides WHICH streams of PyTorch's pool the image-only halves
        run on (_side_stream)."""
        self.model = model
        self.max_graphs = max_graphs
        self.bucket = max(1, int(bucket))
        self._configure(lanes, lane)
        self._graphs = OrderedDict()
        self._seen = {}
        self._enc = {}            # (image shape, device) -> [slot 0 .. slot K-1]: the image-only half as its own graph, K sets of buffers
        self._fifo = []           # outstanding preparations, oldest first: (static image alias, slot index, generation)
        self._slot = 0            # the slot the last forward read
        self._last_alloc = self.SLOTS - 1   # the slot handed out last (prepare() or an unprepared call): the next one is + 1 mod K
        self._side = {}           # device -> the streams prepare() runs the image-only halves on (slot i: stream i % SIDE_STREAMS)
        self._generation = 0      # prepare() calls so far: a static image is only "prepared" for the call that follows ITS prepare()
        self._epoch = None        # fused.route_epoch() the graphs were captured under

    def _configure(self, lanes, lane=0):
        import os
        # (TF_GRAPH_*: A/B switches of the schedule, tools/gpu_runs/gpu_r06_44.sh ... _49.sh)
        self.SLOTS, self.LOOKAHEAD, self.SIDE_STREAMS = self._schedule_for(lanes)
        self._lanes, self._lane = max(1, int(lanes)), int(lane)
        self.SIDE_SPACING = max(1, int(os.environ.get("TF_GRAPH_SIDE_SPACING", "4")))
        self.SIDE_FIRST = int(os.environ.get("TF_GRAPH_SIDE_FIRST", "1"))

    def set_lanes(self, lanes, lane=0):
        """dist_utils.track_sequences: `lanes` sequences are tracked at once, each through its own wrapper (see __init__)."""
        if ((self.SLOTS, self.LOOKAHEAD, self.SIDE_STREAMS) != self._schedule_for(lanes) or (lanes > 1) != (self._lanes > 1)
                or (lanes > 1 and lane != self._lane)):
            if self._graphs or self._enc:
                torch.cuda.synchronize()
                self._graphs.clear()
                self._enc.clear()
                self._seen.clear()
            self._fifo, self._slot, self._side = [], 0, {}
            self._configure(lanes, lane)
            self._last_alloc = self.SLOTS - 1

    def _schedule_for(self, lanes):
        import os
        from .runtime import placement_tuned
        wide = not self._multi_frame() and not hasattr(self.model, "lazy_masks_active") and placement_tuned()
        slots, look_ahead, side_streams = self.WIDE if wide else self.NARROW
        look_ahead = max(1, int(os.environ.get("TF_GRAPH_LOOKAHEAD", look_ahead)))
        return (max(look_ahead + 1, int(os.environ.get("TF_GRAPH_SLOTS", slots))), look_ahead,
                max(1, int(os.environ.get("TF_GRAPH_SIDE_STREAMS", side_streams))))

    def __getattr__(self, name):  # only called for attributes GraphedDetector itself lacks
        return getattr(self.model, name)

    # ------------------------------------------------------------------
    def _sync_epoch(self):
        """Graphs bake the kernels of the fused routes they were captured under: drop them when the routes changed."""
        from . import fused
        epoch = fused.route_epoch()
        if epoch != self._epoch:
            if self._graphs or self._enc:
                torch.cuda.synchronize()
                self._graphs.clear()
                self._enc.clear()
                self._seen.clear()
                self._fifo = []
            self._epoch = epoch

    def _capturable(self, img, target, prev_features):
        from . import fused
        m = self.model
        if torch.is_grad_enabled() or m.training or not getattr(m, "_tracking", True) or fused.debug_checks_active():
            return False
        if not torch.is_tensor(img) or not img.is_cuda or img.dim() != 4 or img.shape[0] != 1:
            return False
        if target is not None and (len(target) != 1
                                   or 'track_query_hs_embeds' not in target[0]):
            return False
        if prev_features is not None and self._multi_frame():
            # the graphs are captured for all-valid padding masks (batch 1: no padding, nested.all_valid_mask); previous-frame
            # features with a real mask (a caller's own, padded) take the eager path
            from .nested import is_all_valid
            if any(f.mask is not None and not is_all_valid(f.mask) for f in prev_features):
                return False
        return True   # prev_features: fed through static buffers (multi-frame models), ignored by the others

    def _multi_frame(self):
        return bool(getattr(self.model, "multi_frame_attention", False))

    @staticmethod
    def _clone_features(features):
        """Owned copies of a frame's features.  An all-valid padding mask (nested.all_valid_mask: a shared constant TAGGED so that
        consumers skip the mask-dependent work -- valid ratios, reference points, masked_fill -- without a device read) is kept as it
        is: a clone would lose the tag, and every replay of a multi-frame graph would recompute 16 linspace / cumsum chains and
        mask-fill the encoder's tensors for the previous frame (round 6: ~1 ms of a 6.5 ms cfg-4 frame)."""
        from .nested import NestedTensor, is_all_valid
        return [NestedTensor(f.tensors.clone(), None if f.mask is None else (f.mask if is_all_valid(f.mask) else f.mask.clone()))
                for f in features]

    # ------------------------------------------------------------------ the forward in two graphs (models without a second
    # frame / mask head): A = backbone + input projections + encoder (the image only), B = decoder + heads (+ the track queries)
    def _splittable(self):
        m = self.model
        # (a mask head reads the features and the encoder memory: both are results of the image-only half, DETRSegmBase.forward
        # takes them through `encoded=` since round 6)
        # (multi-frame models, round 6: the previous frame's BACKBONE features are results of the previous frame's image-only half --
        # the first half of frame t + 1 depends on the images t + 1 and t only; the first frame of a sequence, which attends to
        # itself, keeps the single graph)
        return hasattr(m, "encode_frame")

    def _split_call(self, prev_features):
        return self._splittable() and (not self._multi_frame() or prev_features is not None)

    _PREV_LEVELS = 3   # encode_frame reads prev_features[-3:] (deformable_detr.py:133 of the reference)

    def _prev_tail(self, prev_features):
        return list(prev_features)[-self._PREV_LEVELS:]

    def _side_stream(self, dev, slot=0):
        st = self._side.get(dev)
        if st is None:
            # fixed members of PyTorch's stream pool (runtime.pool_stream: which streams decides the rate): normal-priority streams
            # SIDE_FIRST, SIDE_FIRST + SIDE_SPACING, ... -- (1, 5) next to a sequence on high-priority stream 0
            # (dist_utils.sequence_stream) is the measured-good placement
            # (several lanes: each wrapper its own streams, dist_utils.LANE_SIDES -- lanes on one side stream would run their
            # image-only halves one after the other: three lanes 313 frames/s; a seeded search over layouts, gpu_r06_53.sh:
            # WIDE 391 - 427 frames/s on every layout tried, NARROW 357 - 362 on 34 of 35)
            from .runtime import bind_streams, placement_tuned, pool_stream
            if not placement_tuned():   # (another number of hardware queues than the tables were measured with: round 5's way)
                st = self._side[dev] = tuple(torch.cuda.Stream(dev) for _ in range(self.SIDE_STREAMS))
                return st[slot % self.SIDE_STREAMS]
            bind_streams(dev)
            if self._lanes > 1:
                from .dist_utils import LANE_SIDES, LANE_SIDES_NARROW
                table = LANE_SIDES if self.SIDE_STREAMS > 1 else LANE_SIDES_NARROW
                first = table[self._lane % len(table)]
                st = tuple(pool_stream(dev, (first + 16 * k) % 32) for k in range(self.SIDE_STREAMS))
            else:
                st = tuple(pool_stream(dev, (self.SIDE_FIRST + k * self.SIDE_SPACING) % 32) for k in range(self.SIDE_STREAMS))
            self._side[dev] = st
        return st[slot % self.SIDE_STREAMS]

    def _next_slot(self):
        """Round-robin over the slots, skipping the one the last call decoded (its results may still be read: the lazy mask head,
        a multi-frame model's previous-frame features) and those of outstanding preparations -- at most 1 + LOOKAHEAD < SLOTS."""
        busy = {self._slot} | {e[1] for e in self._fifo}
        i = self._last_alloc
        for _ in range(self.SLOTS):
            i = (i + 1) % self.SLOTS
            if i not in busy:
                break
        self._last_alloc = i
        return i

    def _take_prepared(self, img, prev_features):
        """The outstanding preparation `img` stands for (-> its slot index) or None.  Preparations are consumed in order: a call
        with the tensor of a later one drops the earlier ones, a call with any other image drops them all (a sequence that ended
        early, a caller that changed its mind) -- a dropped static image that is passed in later is an ordinary device image."""
        hit = next((n for n, e in enumerate(self._fifo) if e[0] is img), None)
        if hit is None:
            self._fifo = []
            return None
        alias, slot, generation = self._fifo[hit]
        del self._fifo[:hit + 1]
        slots = self._enc.get((tuple(alias.shape), alias.device))
        if slots is None or slots[slot] is None or slots[slot].get("generation") != generation:
            return None
        if self._multi_frame() and (prev_features is None or slots[slot].get("prev_id") != self._prev_id(prev_features)):
            return None   # prepared against other previous-frame features than this call's: encoded again
        return slot

    def _capture_encoder(self, img, dev, slot=0):
        """One slot of the image-only half: static image, static results, the graph (captured ON the side stream: library
        workspaces that PyTorch keys by capture stream are then not the ones of the decoder graphs it runs next to), and the
        events that order it against the decoder graphs reading its results."""
        side = self._side_stream(dev, slot)
        cur = torch.cuda.current_stream(dev)
        entry = {"done": torch.cuda.Event(), "free": torch.cuda.Event(), "ran": False, "read": False, "stream": side}
        with _CAPTURE_LOCK:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                entry["img"] = img.to(dev, copy=True)
                entry["prev"] = None
                for _ in range(2):
                    warm = self.model.encode_frame(entry["img"], entry["prev"])
                    if self._multi_frame() and entry["prev"] is None:
                        # static copies of the previous frame's features this slot's graph reads: filled before every replay from
                        # wherever they are (the other slot's results, a caller's tensors) -- _feed_prev
                        entry["prev"] = self._clone_features(self._prev_tail(warm["features_all"]))
            cur.wait_stream(side)
            del warm
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                entry["state"] = self.model.encode_frame(entry["img"], entry["prev"])
        entry["graph"] = graph
        return entry

    def prepare(self, img, prev_features=None, device=None, image_ready=False):
        """Enqueue the IMAGE-ONLY half of the forward (backbone, input projections, encoder) for `img` and return the static
        device tensor that now holds the image: calling the detector with THAT tensor then runs the decoder half only.  A
        tracker calls it for frame t + 1 before it associates frame t (Tracker.step_prepare): the GPU works on the next frame
        while the host decides about this one, and -- the half runs on the wrapper's side stream -- next to the decoder
        half of frame t, which on its own leaves most of the chip idle.
        `img`: a host tensor (copied to `device` on the side stream) or a device tensor; the side stream waits for the work
        enqueued so far on the current stream before it reads a device tensor unless `image_ready` says the tensor is
        complete already (frames resident on the device before the sequence started: waiting would queue this frame's
        image-only half behind the previous frame's decoder half).
        None: nothing was enqueued (not a model / input this applies to, or the graph of this image shape does not exist
        yet -- it is captured by the ordinary calls)."""
        from . import fused
        m = self.model
        self._sync_epoch()
        if (not self._splittable() or torch.is_grad_enabled() or m.training or not getattr(m, "_tracking", True)
                or fused.debug_checks_active()):
            return None
        if not torch.is_tensor(img) or img.dim() != 4 or img.shape[0] != 1:
            return None
        dev = img.device if img.is_cuda else (torch.device(device) if device is not None else None)
        if dev is None or dev.type != "cuda":
            return None
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        multi = self._multi_frame()
        if multi and prev_features is None:
            return None   # the first frame of a sequence attends to itself: the single graph of __call__
        slots = self._enc.get((tuple(img.shape), dev))
        if slots is None:
            return None
        # at most LOOKAHEAD frames prepared and not decoded yet (the round-robin of the slots relies on it); a multi-frame model one:
        # frame t + 2's image-only half reads frame t + 1's backbone features, which no call has returned yet
        self._fifo = [e for e in self._fifo if e[0].device == dev and tuple(e[0].shape) == tuple(img.shape)]
        if len(self._fifo) >= (1 if multi else self.LOOKAHEAD):
            return None
        i = self._next_slot()
        if slots[i] is None:
            slots[i] = self._capture_encoder(img, dev, i)
        a = slots[i]
        side, cur = a["stream"], torch.cuda.current_stream(dev)
        # the previous frame's features: results of another slot's run (ordered by that run's `done` event), or somebody else's
        # tensors, produced on the current stream
        src = next((b for b in slots if b is not None and b is not a and self._is_slot_result(b, prev_features)), None) if multi else None
        foreign_prev = multi and src is None
        if (img.is_cuda and not image_ready) or foreign_prev:
            side.wait_stream(cur)
        if a["read"]:
            side.wait_event(a["free"])      # the decoder half (or the lazy mask head) that last read this slot's results is done with them
        if a["ran"]:
            side.wait_event(a["done"])      # (a run into this slot on the CALLER's stream: an unprepared call)
        if src is not None and src["ran"]:
            # that slot's results are this run's previous-frame features: written by its last image-only run -- on a side stream
            # when it was prepared, on the CALLER's stream when it was not (the first frames of a sequence, a bucket's first sight)
            side.wait_event(src["done"])
        with torch.cuda.stream(side):
            a["img"].copy_(img, non_blocking=True)
            if multi:
                self._feed_prev(a, self._prev_tail(prev_features))
            a["graph"].replay()
            a["done"].record(side)
        if multi:
            a["prev_id"] = self._prev_id(prev_features)
        if img.is_cuda:
            img.record_stream(side)
        a["ran"] = True
        self._generation += 1
        a["generation"] = self._generation   # the slot holds THIS preparation until the next one into it
        # a fresh alias per preparation: it stands for "the image-only half of what this tensor holds has run" until the call that
        # decodes it (or a later preparation into the same slot: the generation)
        alias = a["img"].view(a["img"].shape)
        self._fifo.append((alias, i, self._generation))
        return alias

    def _capture(self, img, target, prev_features, slot=0):
        if self._split_call(prev_features):
            return self._capture_decoder(img, target, slot, prev_features)
        entry = {"img": img.clone()}
        static_target = None
        if target is not None:
            entry["boxes"] = target[0]['track_query_boxes'].clone()
            entry["hs"] = target[0]['track_query_hs_embeds'].clone()
            static_target = [{'track_query_boxes': entry["boxes"],
                              'track_query_hs_embeds': entry["hs"],
                              'image_id': target[0].get('image_id')}]
            if 'track_query_filler' in target[0]:
                entry["filler"] = target[0]['track_query_filler'].clone()
                static_target[0]['track_query_filler'] = entry["filler"]
        entry["target"] = static_target
        multi = self._multi_frame()
        entry["prev"] = None
        if multi and prev_features is not None:
            entry["prev"] = self._clone_features(prev_features)
        dev = img.device
        with _CAPTURE_LOCK:
            # warm-up on a side stream (fills every host-side cache: folded conv weights, position
            # encodings, geometry tensors), then capture
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    warm = self.model(entry["img"], static_target, entry["prev"])
            torch.cuda.current_stream(dev).wait_stream(side)
            if multi and entry["prev"] is None:
                # first frame of a sequence: the model attends to the current frame twice; the buffers the
                # NEXT frame will read its previous features from are created here
                entry["prev_out"] = self._clone_features(warm[2])
            del warm
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self.model(entry["img"], static_target, entry["prev"])
                if multi:
                    dst = entry["prev"] if entry["prev"] is not None else entry["prev_out"]
                    from .nested import is_all_valid
                    for d, s in zip(dst, out[2]):
                        d.tensors.copy_(s.tensors)
                        if d.mask is not None and s.mask is not None and not is_all_valid(d.mask):
                            d.mask.copy_(s.mask)
                    out = (out[0], out[1], dst, out[3], out[4])
                entry["out"] = out
        entry["graph"] = graph
        return entry

    @staticmethod
    def _prev_id(prev_features):
        """What identifies a set of previous-frame features for "the image-only half prepare() ran used THESE"."""
        t = list(prev_features)[-1].tensors
        return (t.data_ptr(), t._version)

    @staticmethod
    def _is_slot_result(a, prev_features):
        if a is None or "state" not in a:
            return False
        return list(prev_features)[-1].tensors.data_ptr() == a["state"]["features_all"][-1].tensors.data_ptr()

    def _capture_decoder(self, img, target, slot, prev_features=None):
        akey = (tuple(img.shape), img.device)
        torch.cuda.synchronize(img.device)   # (a prepared image-only half of this slot may be in flight on the side stream)
        slots = self._enc.setdefault(akey, [None] * self.SLOTS)
        if slots[slot] is None:
            slots[slot] = self._capture_encoder(img, img.device, slot)
        a = slots[slot]
        entry = {"enc": a, "prev": None}
        static_target = None
        if target is not None:
            entry["boxes"] = target[0]['track_query_boxes'].clone()
            entry["hs"] = target[0]['track_query_hs_embeds'].clone()
            static_target = [{'track_query_boxes': entry["boxes"], 'track_query_hs_embeds': entry["hs"],
                              'image_id': target[0].get('image_id')}]
            if 'track_query_filler' in target[0]:
                entry["filler"] = target[0]['track_query_filler'].clone()
                static_target[0]['track_query_filler'] = entry["filler"]
        entry["target"] = static_target
        dev = img.device
        with _CAPTURE_LOCK:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                a["img"].copy_(img)
                if a["prev"] is not None:
                    self._feed_prev(a, self._prev_tail(prev_features))
                a["graph"].replay()      # the state the decoder half reads (static buffers of the encoder graph)
                for _ in range(2):
                    warm = self.model(a["img"], static_target, None, encoded=a["state"])
            torch.cuda.current_stream(dev).wait_stream(side)
            del warm
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                entry["out"] = self.model(a["img"], static_target, None, encoded=a["state"])
        entry["graph"] = graph
        return entry

    def _evict_encoders(self, keep=4):
        """The image-only graphs (two slots per image shape, each with its own pool of backbone + encoder activations) live as
        long as a decoder graph references them; beyond that only the `keep` most recently captured shapes stay -- a data set with
        many resolutions must not grow device memory without bound."""
        used = {id(e["enc"]) for e in self._graphs.values() if "enc" in e}
        idle = [k for k, slots in self._enc.items() if not any(a is not None and id(a) in used for a in slots)]
        for k in idle[:max(0, len(idle) - keep)]:
            if any((tuple(e[0].shape), e[0].device) == k for e in self._fifo):
                continue
            torch.cuda.synchronize(k[1])
            del self._enc[k]

    def _feed_prev(self, entry, prev_features):
        """Copy-in of the previous frame's features unless they ARE the entry's static buffers."""
        from .nested import is_all_valid
        for d, s in zip(entry["prev"], prev_features):
            if d.tensors.data_ptr() != s.tensors.data_ptr():
                d.tensors.copy_(s.tensors, non_blocking=True)
                if d.mask is not None and s.mask is not None and not is_all_valid(d.mask):   # (_capturable: all-valid in, all-valid here)
                    d.mask.copy_(s.mask, non_blocking=True)

    def _bucketed(self, target):
        """-> (padded target, number of real track queries, padded number)."""
        n_real = int(target[0]['track_query_hs_embeds'].shape[0])
        n_pad = -(-n_real // self.bucket) * self.bucket
        if self.bucket <= 1 or not getattr(self.model, "track_query_filler_ok", False) or n_real == 0:
            return target, n_real, n_real
        hs, boxes = target[0]['track_query_hs_embeds'], target[0]['track_query_boxes']
        extra = n_pad - n_real
        filler = torch.zeros(n_pad, dtype=torch.bool, device=hs.device)
        if extra:
            hs = torch.cat([hs, hs.new_zeros((extra, hs.shape[1]))])
            boxes = torch.cat([boxes, boxes.new_tensor([0.5, 0.5, 0.1, 0.1]).expand(extra, 4)])
            filler[n_real:] = True
        padded = dict(target[0], track_query_hs_embeds=hs, track_query_boxes=boxes, track_query_filler=filler)
        return [padded], n_real, n_pad

    @staticmethod
    def _strip(out, hs, n_real, n_pad):
        """Drop the filler rows n_real .. n_pad of every per-query output."""
        if n_pad == n_real:
            return out, hs

        def cut(t, dim):
            return torch.cat([t.narrow(dim, 0, n_real), t.narrow(dim, n_pad, t.shape[dim] - n_pad)], dim)
        out = dict(out)
        for k in ('pred_logits', 'pred_boxes', 'hs_embed', 'pred_masks'):
            if k in out:
                out[k] = cut(out[k], 1)
        if 'aux_outputs' in out:
            out['aux_outputs'] = [{k: cut(v, 1) for k, v in a.items()} for a in out['aux_outputs']]
        return out, cut(hs, 2)

    def _wait_static_image(self, img):
        """`img` may be a static image prepare() handed out (now or earlier): the side stream wrote it -- the current stream
        must not read it before that run is done."""
        if not (torch.is_tensor(img) and img.is_cuda):
            return
        for a in self._enc.get((tuple(img.shape), img.device)) or ():
            if a is not None and a["ran"] and a["img"].data_ptr() == img.data_ptr():
                torch.cuda.current_stream(img.device).wait_event(a["done"])

    def state_read(self, device=None):
        """A consumer outside the detector call has just ENQUEUED a read of the last call's aliased results (`features`, `memory`,
        out['mask_context']: static buffers of the slot that call decoded) on the current stream -- the tracker's lazy mask head runs
        after the association.  prepare() must not overwrite that slot before this read: the slot's `free` event moves behind it."""
        for slots in self._enc.values():
            a = slots[self._slot]
            if a is not None and a["read"] and (device is None or a["img"].device == torch.device(device)):
                a["free"].record(torch.cuda.current_stream(a["img"].device))

    def _mark_read(self, img, slot):
        """An eager forward on the current stream read slot `slot`'s static image: prepare() must not overwrite it before."""
        a = self._enc[(tuple(img.shape), img.device)][slot]
        a["free"].record(torch.cuda.current_stream(img.device))
        a["read"] = True
        self._slot = slot

    def __call__(self, img, target=None, prev_features=None):
        self._sync_epoch()
        prepared = self._take_prepared(img, prev_features)   # the slot prepare() filled for this very tensor, or None
        if not self._capturable(img, target, prev_features):
            self._wait_static_image(img)   # (also consumes the preparation: the eager forward encodes the image itself)
            res = self.model(img, target, prev_features)
            if prepared is not None:
                self._mark_read(img, prepared)
            return res
        multi = self._multi_frame()
        n_real = n_pad = 0
        caller_target = target
        if target is not None:
            target, n_real, n_pad = self._bucketed(target)
        n_track = n_pad
        lazy = getattr(self.model, "lazy_masks_active", None)
        # the buffers prepare() filled for this very tensor; an unprepared call decodes slot 0 (no preparation is outstanding then:
        # _take_prepared dropped them)
        slot = prepared if prepared is not None else 0
        self._wait_static_image(img)   # the slot's run, or a static image used after its preparation was forgotten
        key = (tuple(img.shape), n_track, img.device, bool(multi and prev_features is not None),
               bool(lazy()) if lazy is not None else False,   # a graph with and one without the mask head are different graphs
               slot)                                          # the decoder half reads ONE slot's static buffers
        entry = self._graphs.get(key)
        if entry is None:
            # capture a shape the second time it shows up (one-off shapes are not worth a graph)
            seen_key = key[:-1]   # without the slot: a bucket seen on one slot is captured at its first sight on the other
            self._seen[seen_key] = self._seen.get(seen_key, 0) + 1
            if self._seen[seen_key] < 2:
                res = self.model(img, caller_target, prev_features)
                if prepared is not None:   # the eager forward read the slot's static image: prepare() must not overwrite it yet
                    self._mark_read(img, slot)
                return res
            entry = self._capture(img, target, prev_features, slot)
            self._graphs[key] = entry
            while len(self._graphs) > self.max_graphs:
                self._graphs.popitem(last=False)   # least recently used
            self._evict_encoders()
        else:
            self._graphs.move_to_end(key)
        if target is not None:
            entry["boxes"].copy_(target[0]['track_query_boxes'], non_blocking=True)
            entry["hs"].copy_(target[0]['track_query_hs_embeds'], non_blocking=True)
            if "filler" in entry:
                entry["filler"].copy_(target[0]['track_query_filler'], non_blocking=True)
        if "enc" in entry:   # two graphs: the image-only half unless prepare() already ran it for this very tensor
            a = entry["enc"]
            cur = torch.cuda.current_stream(img.device)
            if a["ran"]:
                cur.wait_event(a["done"])   # the side stream's last run into this slot (the one prepare() started, or a stale one)
            if prepared is None:
                a["img"].copy_(img, non_blocking=True)
                if a["prev"] is not None:
                    self._feed_prev(a, self._prev_tail(prev_features))
                a["graph"].replay()
                a["done"].record(cur)       # (this slot's results now come from a run on the caller's stream: prepare() of the
                a["ran"] = True             # other slot, which reads them as previous-frame features, waits for this event)
        else:
            entry["img"].copy_(img, non_blocking=True)
        if entry["prev"] is not None:
            self._feed_prev(entry, prev_features)
        entry["graph"].replay()
        if "enc" in entry:
            a["free"].record(cur)
            a["read"] = True
            self._slot = slot
        out, tgt, features, memory, hs = entry["out"]
        # what the tracker keeps across frames must not alias the static buffers (see the module docstring)
        out = dict(out)
        if n_pad != n_real:
            out, hs = self._strip(out, hs, n_real, n_pad)   # torch.cat: fresh tensors, nothing aliases
        else:
            out['hs_embed'] = out['hs_embed'].clone()
            out['pred_logits'] = out['pred_logits'].clone()
            out['pred_boxes'] = out['pred_boxes'].clone()
        return out, caller_target, features, memory, hs
