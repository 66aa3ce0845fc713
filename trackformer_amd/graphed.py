"""HIP-graph replay of the detector forward for fixed-shape tracking steps.

At batch 1 the per-frame forward is ~900 small-to-medium kernels (53 convolutions, 12 attention
layers of ~15 launches each, heads); launched eagerly from Python the host cannot feed an MI355X
fast enough.  `GraphedDetector` captures `model(img, [track queries], None)` once per
(image shape, number of track queries) into a HIP graph with static input/output buffers and replays
it for every later frame of that shape: one graph launch instead of ~900 kernel launches, same kernels,
same results.

It is a drop-in for the `obj_detector` argument of Tracker: attribute access (num_queries,
overflow_boxes, parameters(), ...) is forwarded to the wrapped model.  Calls it cannot replay --
multi-frame attention (prev_features are per-frame inputs), training mode, gradients enabled -- run
the wrapped model eagerly.
"""
import torch


class GraphedDetector:
    def __init__(self, model, max_graphs=8):
        self.model = model
        self.max_graphs = max_graphs
        self._graphs = {}
        self._seen = {}

    def __getattr__(self, name):  # only called for attributes GraphedDetector itself lacks
        return getattr(self.model, name)

    # ------------------------------------------------------------------
    def _capturable(self, img, target, prev_features):
        m = self.model
        if torch.is_grad_enabled() or m.training or not getattr(m, "_tracking", True):
            return False
        if getattr(m, "multi_frame_attention", False):
            return False
        if not torch.is_tensor(img) or not img.is_cuda or img.dim() != 4:
            return False
        if target is not None and (len(target) != 1
                                   or 'track_query_hs_embeds' not in target[0]):
            return False
        return True

    def _capture(self, img, target):
        entry = {"img": img.clone()}
        static_target = None
        if target is not None:
            entry["boxes"] = target[0]['track_query_boxes'].clone()
            entry["hs"] = target[0]['track_query_hs_embeds'].clone()
            static_target = [{'track_query_boxes': entry["boxes"],
                              'track_query_hs_embeds': entry["hs"],
                              'image_id': target[0].get('image_id')}]
        entry["target"] = static_target
        # warm-up on a side stream (fills every host-side cache: folded conv weights, position
        # encodings, geometry tensors), then capture
        side = torch.cuda.Stream(img.device)
        side.wait_stream(torch.cuda.current_stream(img.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self.model(entry["img"], static_target, None)
        torch.cuda.current_stream(img.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            entry["out"] = self.model(entry["img"], static_target, None)
        entry["graph"] = graph
        return entry

    def __call__(self, img, target=None, prev_features=None):
        if not self._capturable(img, target, prev_features):
            return self.model(img, target, prev_features)
        n_track = 0 if target is None else int(target[0]['track_query_hs_embeds'].shape[0])
        key = (tuple(img.shape), n_track, img.device)
        entry = self._graphs.get(key)
        if entry is None:
            # capture a shape the second time it shows up (one-off shapes are not worth a graph)
            self._seen[key] = self._seen.get(key, 0) + 1
            if self._seen[key] < 2 or len(self._graphs) >= self.max_graphs:
                return self.model(img, target, prev_features)
            entry = self._graphs[key] = self._capture(img, target)
        entry["img"].copy_(img, non_blocking=True)
        if target is not None:
            entry["boxes"].copy_(target[0]['track_query_boxes'], non_blocking=True)
            entry["hs"].copy_(target[0]['track_query_hs_embeds'], non_blocking=True)
        entry["graph"].replay()
        out, tgt, features, memory, hs = entry["out"]
        # everything the caller keeps across frames must not alias the static buffers
        out = dict(out)
        out['hs_embed'] = out['hs_embed'].clone()
        out['pred_logits'] = out['pred_logits'].clone()
        out['pred_boxes'] = out['pred_boxes'].clone()
        return out, target, features, memory, hs
