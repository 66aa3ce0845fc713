"""Online multi-object tracker driving the detector frame by frame.

Same API and decision logic as the reference's models/tracker.py -- Tracker(obj_detector,
obj_detector_post, tracker_cfg, generate_attention_maps, logger, verbose), .reset(hard), .step(blob),
.get_results(), and the Track state object (:557-583) -- so src/track.py can use it unchanged.

What changed is WHERE the association logic runs.  The reference interleaves GPU tensors with Python
control flow and pays one device->host sync per track and per decision (tracker.py:290, :346, :367,
:400-402, :496, :530-541).  Here each frame does exactly ONE device->host copy (boxes, scores and
labels of all queries packed into a [Lq, 6] tensor); every threshold, NMS, public-detection and
bookkeeping decision is then taken on that host copy in the reference's order, and track positions /
scores are kept as CPU tensors.  Only the per-query output embeddings stay on the GPU (they are fed
back as track queries and never inspected on the host, except by the optional embedding re-ID).
"""
import os
from collections import deque

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from .box_ops import box_iou, box_xyxy_to_cxcywh, clip_boxes_to_image, nms


class Tracker:
    """Tracks objects through a sequence by re-feeding their output embeddings as track queries."""

    def __init__(self, obj_detector, obj_detector_post, tracker_cfg, generate_attention_maps,
                 logger=None, verbose=False, lazy_masks=None):
        self.obj_detector = obj_detector
        self.obj_detector_post = obj_detector_post
        # DEFAULT since round 3 (lazy_masks=False / TF_LAZY_MASKS=0 switches it off): with a mask head, run it only for the
        # queries whose masks the step ends up keeping (DETRSegmBase.mask_rows) instead of for all of them inside the
        # detector forward -- cfg 5 on MI355X: 21.3 -> 178.6 frames/s (profiles/r03_optin_summary.txt)
        if lazy_masks is None:
            lazy_masks = os.environ.get("TF_LAZY_MASKS", "1") != "0"
        module = getattr(obj_detector, "model", obj_detector)   # GraphedDetector wraps the nn.Module
        # scoped to this tracker's own detector calls (detr_segmentation.lazy_mask_scope): the shared module is not modified
        self._lazy_masks = bool(lazy_masks) and "segm" in obj_detector_post and hasattr(module, "mask_rows")
        self.detection_obj_score_thresh = tracker_cfg['detection_obj_score_thresh']
        self.track_obj_score_thresh = tracker_cfg['track_obj_score_thresh']
        self.detection_nms_thresh = tracker_cfg['detection_nms_thresh']
        self.track_nms_thresh = tracker_cfg['track_nms_thresh']
        self.public_detections = tracker_cfg['public_detections']
        self.inactive_patience = float(tracker_cfg['inactive_patience'])
        self.reid_sim_threshold = tracker_cfg['reid_sim_threshold']
        self.reid_sim_only = tracker_cfg['reid_sim_only']
        self.generate_attention_maps = generate_attention_maps
        self.reid_score_thresh = tracker_cfg['reid_score_thresh']
        self.reid_greedy_matching = tracker_cfg['reid_greedy_matching']
        self.prev_frame_dist = tracker_cfg['prev_frame_dist']
        self.steps_termination = tracker_cfg['steps_termination']

        if self.generate_attention_maps:
            last = self.obj_detector.transformer.decoder.layers[-1]
            assert hasattr(last, 'multihead_attn'), \
                'Generation of attention maps not possible for deformable DETR.'
            attention_data = {'maps': None, 'conv_features': {}, 'hooks': []}

            def keep_conv_features(module, inputs, output):
                attention_data['conv_features'] = output

            def keep_attention_map(module, inputs, output):
                h, w = attention_data['conv_features']['3'].tensors.shape[-2:]
                attention_data['maps'] = output[1].view(-1, h, w)

            attention_data['hooks'].append(
                self.obj_detector.backbone[-2].register_forward_hook(keep_conv_features))
            attention_data['hooks'].append(
                last.multihead_attn.register_forward_hook(keep_attention_map))
            self.attention_data = attention_data

        self._logger = logger if logger is not None else (lambda *log_strs: None)
        self._verbose = verbose
        # DEFAULT since round 6 (TF_DEFERRED_STEP=0 / tracker.deferred = False: step() returns after its association, as the
        # reference's does): step() returns once the frame's GPU work is ENQUEUED and runs the frame's association at the start of
        # the next step(), after that frame's image-only half has been enqueued (see step()).
        self.deferred = os.environ.get("TF_DEFERRED_STEP", "1") != "0" and not generate_attention_maps   # (the attention hooks
        # file the maps of the LAST forward: the next frame's image-only half must not run before this frame's are read)

    # ------------------------------------------------------------------ state
    @property
    def num_object_queries(self):
        return self.obj_detector.num_queries

    @property
    def device(self):
        return next(self.obj_detector.parameters()).device

    def _image_id(self, device):
        t = getattr(self, "_image_id_cache", None)
        if t is None or t.device != device:
            t = self._image_id_cache = torch.ones(1, dtype=torch.int64, device=device)
        return t

    @staticmethod
    def _track_query_embeds(prev_tracks, device):
        """[n, C]: every track's newest output embedding (tracker.py:290-297 of the reference stacks them).  When all of them
        are rows of one frame's embeddings -- the usual case: the tracks were updated or created by the previous frame -- that
        is one index_select; any other mix falls back to the stack."""
        items = [t.hs_embed._items[-1] for t in prev_tracks]
        first = items[0]
        if type(first) is tuple:
            src = first[0]
            if all(type(it) is tuple and it[0] is src for it in items):
                idx = torch.tensor([it[1] for it in items], dtype=torch.long)
                return src.index_select(0, idx.to(src.device, non_blocking=True))
        return torch.stack([_HsHistory._row(it) for it in items], dim=0)

    def reset(self, hard=True):
        self._flush_deferred()   # (a frame whose association is still outstanding belongs to the results a soft reset keeps)
        self.tracks = []
        self.inactive_tracks = []
        self._prev_features = deque([None], maxlen=self.prev_frame_dist)
        self._inflight_features = None
        self._prepared = None
        if hard:
            self.track_num = 0
            self._results = {}
            self._pending_results = []
            self.frame_index = 0
            self.num_reids = 0

    def _keep_after_nms(self, pos, scores, rank, thresh, name):
        """NMS over self.tracks (pos [n, 4] / scores [n] in list order, suppression order by `rank`): drops the suppressed
        tracks from the list and returns (pos, scores) of the survivors, still in list order."""
        keep = torch.zeros(pos.shape[0], dtype=torch.bool)
        keep[nms(pos, rank, thresh)] = True
        flags = keep.tolist()
        if not all(flags):
            self._logger(f'REMOVE TRACK IDS ({name}={thresh}): '
                         f'{[t.id for t, k in zip(self.tracks, flags) if not k]}')
            self.tracks = [t for t, k in zip(self.tracks, flags) if k]
            return pos[keep], scores[keep]
        return pos, scores

    def tracks_to_inactive(self, tracks):
        self.tracks = [t for t in self.tracks if t not in tracks]
        for track in tracks:
            track.pos = track.last_pos[-1]
        self.inactive_tracks += tracks

    def add_tracks(self, pos, scores, hs_embeds, indices, masks=None, attention_maps=None,
                   aux_results=None, hs_rows=None):
        """Creates Track objects with consecutive ids track_num, track_num+1, ... (tracker.py:93-122).  hs_rows = (frame
        embeddings [Q, C], [row per new track]) may stand in for hs_embeds [n, C]: the tracks then reference their rows
        (_HsHistory) instead of holding n gathered copies."""
        n = len(pos)
        new_track_ids = list(range(self.track_num, self.track_num + n))
        if n:
            # one unbind per tensor instead of four Python-level indexings per track (150 new tracks: 0.8 -> 0.2 ms)
            indices = torch.as_tensor(indices)
            ints = indices.reshape(n, -1)[:, 0].tolist()   # the results of every later frame carry obj_ind as an int
            if hs_rows is not None:
                src, src_rows = hs_rows
                hs_iter = [(src, r) for r in src_rows]
            else:
                hs_iter = hs_embeds.unbind(0)
            append = self.tracks.append
            for i, (hs, ind) in enumerate(zip(hs_iter, ints)):   # rows of `pos` / `scores` as references, obj_ind as an int
                append(Track((pos, i), (scores, i), self.track_num + i, hs, ind,
                             None if masks is None else masks[i],
                             None if attention_maps is None else attention_maps[i]))
        self.track_num += n

        if new_track_ids:
            self._logger(
                f'INIT TRACK IDS (detection_obj_score_thresh={self.detection_obj_score_thresh}): '
                f'{new_track_ids}')
            if aux_results is not None:
                idx = torch.as_tensor(indices).reshape(-1).to(aux_results[0]['scores'].device)
                aux_scores = torch.cat(
                    [a['scores'][-self.num_object_queries:][idx][:, None] for a in aux_results]
                    + [torch.as_tensor(scores).reshape(-1, 1).to(idx.device)], dim=-1)
                for new_track_id, aux_score in zip(new_track_ids, aux_scores):
                    self._logger(
                        f"AUX SCORES ID {new_track_id}: {[f'{s:.2f}' for s in aux_score]}")
        return new_track_ids

    def public_detections_mask(self, new_det_boxes, public_det_boxes):
        """Mask over the new detections that are backed by a provided public detection."""
        n = new_det_boxes.size(0)
        if not self.public_detections:
            return torch.ones(n, dtype=torch.bool)
        if not len(public_det_boxes) or not n:
            return torch.zeros(n, dtype=torch.bool)
        mask = torch.zeros(n, dtype=torch.bool)
        public_det_boxes = public_det_boxes.detach().cpu().float()
        if self.public_detections == 'center_distance':
            item_size = ((new_det_boxes[:, 2] - new_det_boxes[:, 0])
                         * (new_det_boxes[:, 3] - new_det_boxes[:, 1])).numpy().astype(np.float32)
            det_c = box_xyxy_to_cxcywh(new_det_boxes).numpy()[:, :2]
            pub_c = box_xyxy_to_cxcywh(public_det_boxes).numpy()[:, :2]
            dist = ((det_c.reshape(-1, 1, 2) - pub_c.reshape(1, -1, 2)) ** 2).sum(axis=2)
            for j in range(len(public_det_boxes)):
                i = dist[:, j].argmin()
                if dist[i, j] < item_size[i]:
                    dist[i, :] = 1e18
                    mask[i] = True
        elif self.public_detections == 'min_iou_0_5':
            iou = box_iou(new_det_boxes, public_det_boxes)
            for j in range(len(public_det_boxes)):
                i = iou[:, j].argmax()
                if iou[i, j] >= 0.5:
                    iou[i, :] = 0
                    mask[i] = True
        else:
            raise NotImplementedError
        return mask

    def reid(self, new_det_boxes, new_det_scores, new_det_hs_embeds, new_det_masks=None,
             new_det_attention_maps=None):
        """Re-identify inactive tracks among the new detections; returns the mask of detections that
        remain unassigned (tracker.py:167-264)."""
        self.inactive_tracks = [
            t for t in self.inactive_tracks
            if t.has_positive_area() and t.count_inactive <= self.inactive_patience]
        n = new_det_boxes.size(0)
        if not self.inactive_tracks or not n:
            return torch.ones(n, dtype=torch.bool)

        if self.reid_greedy_matching:
            det = box_xyxy_to_cxcywh(new_det_boxes).numpy()
            inact = box_xyxy_to_cxcywh(_gather_rows([t._pos for t in self.inactive_tracks])).numpy()
            dist_mat = ((inact[:, :2].reshape(-1, 1, 2) - det[:, :2].reshape(1, -1, 2)) ** 2
                        ).sum(axis=2)
            track_size = inact[:, 2] * inact[:, 3]
            item_size = det[:, 2] * det[:, 3]
            invalid = ((dist_mat > track_size.reshape(-1, 1)) + (dist_mat > item_size.reshape(1, -1)))
            dist_mat = dist_mat + invalid * 1e18
            matched = []
            if dist_mat.shape[1]:
                for i in range(dist_mat.shape[0]):
                    j = dist_mat[i].argmin()
                    if dist_mat[i][j] < 1e16:
                        dist_mat[:, j] = 1e18
                        dist_mat[i, j] = 0.0
                        matched.append([i, j])
            matched = np.array(matched, np.int32).reshape(-1, 2)
            row_indices, col_indices = matched[:, 0], matched[:, 1]
        else:
            track_sims = torch.stack([t.hs_embed[-1] for t in self.inactive_tracks])  # [I, C]
            # F.pairwise_distance semantics (eps = 1e-6 added to the difference), one D2H for all
            diff = track_sims[:, None, :] - new_det_hs_embeds[None, :, :] + 1e-6
            dist_mat = diff.norm(p=2, dim=-1).cpu().numpy()
            row_indices, col_indices = linear_sum_assignment(dist_mat)

        assigned, revived = [], []
        for row_ind, col_ind in zip(row_indices, col_indices):
            if dist_mat[row_ind, col_ind] <= self.reid_sim_threshold:
                track = self.inactive_tracks[row_ind]
                self._logger(
                    f'REID: track.id={track.id} - count_inactive={track.count_inactive} - '
                    f'to_inactive_frame={self.frame_index - track.count_inactive}')
                track.count_inactive = 0
                track.pos = new_det_boxes[col_ind]
                track.score = new_det_scores[col_ind]
                track.hs_embed.append(new_det_hs_embeds[col_ind])
                track.reset_last_pos()
                if new_det_masks is not None:
                    track.mask = new_det_masks[col_ind]
                if new_det_attention_maps is not None:
                    track.attention_map = new_det_attention_maps[col_ind]
                assigned.append(col_ind)
                revived.append(track)
                self.tracks.append(track)
                self.num_reids += 1
        for track in revived:
            self.inactive_tracks.remove(track)
        reid_mask = torch.ones(n, dtype=torch.bool)
        for ind in assigned:
            reid_mask[ind] = False
        return reid_mask

    # ------------------------------------------------------------------ one frame
    def step(self, blob):
        """Process one frame.  blob: {'img' [1,3,H,W], 'orig_size' [1,2] (h,w), 'size' [1,2],
        'dets' [1,K,4] xyxy public detections} as produced by the reference's sequence datasets.

        Deferred association (round 6, default; `tracker.deferred = False` / TF_DEFERRED_STEP=0: everything before returning).
        The reference's loop is `for frame_data in seq_loader: tracker.step(frame_data)` (src/track.py:130-134): nothing reads the
        tracker between two steps.  step(t) therefore returns when frame t's forward, post-processing and device -> host copy are
        ENQUEUED; step(t + 1) first enqueues the image-only half of frame t + 1 (step_prepare: backbone + encoder need no tracks),
        THEN waits for frame t's copy and associates it on the host -- while the GPU is already on frame t + 1 -- and then
        enqueues frame t + 1's decoder half with the track queries that association produced.  Same operations on the same
        data in the same order per frame as step_finish(step_async(blob)); the schedule is the pipelined loop's
        (step_async(t) -> step_prepare(t + 1) -> step_finish(t)) without the caller knowing the next frame in advance.  Every
        read of the tracker's state from outside (tracks, inactive_tracks, track_num, frame_index, num_reids, results /
        get_results(), reset()) first runs the outstanding association, so an observer never sees the difference."""
        if not self.deferred:
            self.step_finish(self.step_async(blob))
            return
        d = self.__dict__
        d["_busy"] = True
        try:
            handle = d.get("_deferred_handle")
            if handle is not None:
                self.step_prepare(blob)          # frame t + 1's image-only half: next to frame t's decoder half and association
                d["_deferred_handle"] = None
                self.step_finish(handle)
            d["_deferred_handle"] = self.step_async(blob)
        finally:
            d["_busy"] = False

    def _flush_deferred(self):
        """Run the association a deferred step() left outstanding (no-op otherwise)."""
        d = self.__dict__
        handle = d.get("_deferred_handle")
        if handle is None or d.get("_busy"):
            return
        d["_deferred_handle"] = None
        d["_busy"] = True
        try:
            self.step_finish(handle)
        finally:
            d["_busy"] = False

    def step_prepare(self, blob, image_ready=False):
        """Optional, BEFORE step_finish of the previous frame: enqueue the image-only half of `blob`'s forward (backbone,
        input projections, encoder -- GraphedDetector.prepare, or the model's encode_frame) so that the GPU works on this
        frame while the host still associates the previous one; the following step_async(blob) / step(blob) of the SAME
        blob then runs the decoder half only.  Results are those of step(): the image-only half does not depend on the
        tracks.  Multi-frame models (round 6): their first half also reads the previous frame's BACKBONE features -- results of
        the previous frame's first half, so they are known here as well (_upcoming_prev_features).  Returns whether anything was
        enqueued (not for the first frame of a multi-frame sequence, which attends to itself)."""
        det = self.obj_detector
        prep = getattr(det, "prepare", None)
        pending = self.__dict__.get("_prepared") or []
        multi = bool(getattr(det, "multi_frame_attention", False))
        if multi and pending:
            return False   # the frame after a prepared one: its previous-frame features are those of a frame not yet decoded
        prev = self._upcoming_prev_features() if multi else None
        if multi and prev is None:
            self._prepared = []
            return False
        if prep is not None:
            # GraphedDetector: the half runs on one of the wrapper's side streams, NEXT TO the previous frame's decoder half (a host
            # image is uploaded on that stream too); `image_ready`: a device-resident blob['img'] is complete already.  Up to
            # GraphedDetector.LOOKAHEAD frames may be prepared and not yet stepped (round 6: the image-only halves of frames
            # t + 1 and t + 2 share the chip); one more is refused (None)
            img = prep(blob['img'], prev, device=self.device, image_ready=image_ready)
            if img is not None:
                self._prepared = pending + [(blob['img'], img, None)]
                return True
            return False
        img = blob['img'].to(self.device, non_blocking=True)
        if hasattr(det, "encode_frame") and not det.training and not torch.is_grad_enabled():
            self._prepared = [(blob['img'], img, det.encode_frame(img, prev), prev)]   # (without graphs: one frame ahead)
            return True
        return False

    @property
    def look_ahead(self):
        """How many frames step_prepare may run ahead of step_async: GraphedDetector.LOOKAHEAD with graphs and a single-frame
        model, otherwise 1 (a multi-frame model's image-only half needs the previous frame's backbone features)."""
        det = self.obj_detector
        if bool(getattr(det, "multi_frame_attention", False)) or not hasattr(det, "prepare"):
            return 1
        return int(getattr(det, "LOOKAHEAD", 1))

    def _upcoming_prev_features(self):
        """`self._prev_features[0]` as the NEXT step_async will see it: step_prepare runs between step_async(t) and
        step_finish(t), and it is step_finish(t) that files frame t's features (tracker.py:331 of the reference appends them at
        the end of step)."""
        inflight = self.__dict__.get("_inflight_features")
        if inflight is None:
            return self._prev_features[0]
        future = deque(self._prev_features, maxlen=self.prev_frame_dist)
        future.append(inflight)   # (distance 1: the in-flight features themselves, aliases of the detector's buffers as they are
        return future[0]          # in step(); a longer distance reads an older entry, which step_finish made its own)

    def step_async(self, blob):
        """First half of step(): builds the track queries, ENQUEUES the detector forward, the post-processing and the frame's
        single device -> host copy on the current stream, and returns without waiting for any of it.  step_finish(handle)
        waits for the copy and runs the association.  A process that tracks several independent sequences on one GPU
        interleaves them -- finish A, launch A's next frame, finish B, launch B's next frame, ... -- so that one sequence's
        association (host) runs under the other's forward (GPU) in ONE thread: tracker threads of one interpreter serialise
        on the GIL instead (measured on MI355X with ~100 live tracks and ~150 detections per frame: 4 threads 111 frames/s,
        1 thread 145).  Between step_async and step_finish of one tracker nothing else may touch that tracker."""
        self.inactive_tracks = [
            t for t in self.inactive_tracks
            if t.has_positive_area() and t.count_inactive <= self.inactive_patience]

        self._logger(f'FRAME: {self.frame_index + 1}')
        if self.inactive_tracks:
            self._logger(f'INACTIVE TRACK IDS: {[t.id for t in self.inactive_tracks]}')

        for track in self.tracks:   # host tensors are never modified in place: no clone needed; a reference is filed as it is
            track.last_pos.append(track._pos)

        device = self.device
        # the preparation of THIS blob, if any (step_prepare; in order: earlier ones of other blobs are dropped with it)
        pending = self.__dict__.get("_prepared") or []
        hit = next((n for n, e in enumerate(pending) if e[0] is blob['img']), None)
        prepared = pending[hit] if hit is not None else None
        self._prepared = pending[hit + 1:] if hit is not None else []
        encoded = None
        if prepared is not None and prepared[0] is blob['img'] and (len(prepared) < 4 or prepared[3] is self._prev_features[0]):
            img, encoded = prepared[1], prepared[2]     # step_prepare(blob) ran the image-only half for this very frame
        else:
            img = blob['img'].to(device, non_blocking=True)
        orig_size_host = blob['orig_size'].detach().cpu()
        orig_size = None   # (its device copy: made when a post-processor module is called, not for the fused post-processing)
        orig_h, orig_w = int(orig_size_host[0, 0]), int(orig_size_host[0, 1])

        target = None
        prev_tracks = self.tracks + self.inactive_tracks
        num_prev_track = len(prev_tracks)
        if num_prev_track:
            boxes_xyxy = _gather_rows([t._pos for t in prev_tracks])                # host
            track_query_boxes = box_xyxy_to_cxcywh(boxes_xyxy) / torch.tensor(
                [orig_w, orig_h, orig_w, orig_h], dtype=torch.float32)
            target = [{
                'track_query_boxes': track_query_boxes.to(device, non_blocking=True),
                'image_id': self._image_id(device),
                'track_query_hs_embeds': self._track_query_embeds(prev_tracks, device),
            }]

        if self._lazy_masks:
            from .detr_segmentation import lazy_mask_scope
            with lazy_mask_scope():
                if encoded is not None:
                    outputs, _, features, _, _ = self.obj_detector(img, target, self._prev_features[0], encoded=encoded)
                else:
                    outputs, _, features, _, _ = self.obj_detector(img, target, self._prev_features[0])
        elif encoded is not None:
            outputs, _, features, _, _ = self.obj_detector(img, target, self._prev_features[0], encoded=encoded)
        else:
            outputs, _, features, _, _ = self.obj_detector(img, target, self._prev_features[0])
        hs_embeds = outputs['hs_embed'][0]

        packed_dev = None
        post = self.obj_detector_post['bbox']
        if ("segm" not in self.obj_detector_post and type(post).__name__ == "DeformablePostProcess"
                and type(post).__module__.startswith("trackformer_amd.") and outputs['pred_logits'].shape[0] == 1):
            # sigmoid + best class + box scaling (DeformablePostProcess.forward), clip_boxes_to_image and the stacking below in
            # ONE launch (fused.postprocess_pack: the same operations, each rounded on its own); nothing else of the
            # post-processor's result is read for a model without a mask head
            from . import fused
            packed_dev = fused.postprocess_pack(outputs['pred_logits'][0], outputs['pred_boxes'][0], orig_h, orig_w,
                                                clip=not self.obj_detector.overflow_boxes)
        if packed_dev is not None:
            results, result = None, {}
        else:
            orig_size = orig_size_host.to(device, non_blocking=True)
            results = self.obj_detector_post['bbox'](outputs, orig_size)
            result = results[0]
        if "segm" in self.obj_detector_post:
            # The reference resizes the masks of ALL queries to the original image size here (tracker.py:315-319:
            # 400 x 800 x 1333 floats per frame) and then keeps those of the surviving tracks.  A query's mask
            # depends on nothing but that query, and no decision below looks at a mask, so the tracks get
            # references (_MaskRef: the query's row) and only the referenced rows are resized, once, at the end of
            # the step (_resolve_masks) -- same values, a fraction of the work.
            result['masks'] = _MaskRows(len(result['scores']))

        if packed_dev is None:
            boxes_dev = result['boxes']
            if not self.obj_detector.overflow_boxes:
                boxes_dev = clip_boxes_to_image(boxes_dev, (orig_h, orig_w))
            packed_dev = torch.cat([boxes_dev, result['scores'][:, None],
                                    result['labels'][:, None].to(boxes_dev.dtype)], dim=1)
        # packed_dev: the frame's single device -> host transfer: enqueued here, awaited in step_finish
        event = host = None
        if packed_dev.device.type == "cuda":
            host = self._pinned_rows(packed_dev.shape[0])
            # the copy and its event go on the stream of packed_dev's device that the detector just ran on -- NOT on "the
            # current device's current stream": with the model on cuda:1 and no torch.cuda.set_device, a bare
            # event.record() lands on an idle stream of cuda:0, synchronize() returns at once and step_finish reads the
            # pinned buffer before the copy has landed
            with torch.cuda.device(packed_dev.device):
                host[:packed_dev.shape[0]].copy_(packed_dev, non_blocking=True)
                event = torch.cuda.Event()
                event.record(torch.cuda.current_stream(packed_dev.device))
        self._inflight_features = features   # (filed by step_finish; step_prepare of the next frame reads them before that)
        return dict(blob=blob, outputs=outputs, features=features, hs_embeds=hs_embeds, results=results, result=result,
                    orig_size=orig_size, orig_hw=(orig_h, orig_w), num_prev_track=num_prev_track, device=device,
                    packed_dev=packed_dev, host=host, event=event)

    def _pinned_rows(self, n):
        """The tracker's pinned host buffer for a frame's [n, 6] result rows (grown when needed; reused by every frame:
        step_finish clones what it reads).  (Round 6 also had the post-processing kernel write the rows straight into this buffer --
        pinned memory is mapped into the device's address space -- instead of device rows + an asynchronous copy: ids identical,
        no change of any rate, tools/gpu_runs/gpu_r06_45.sh; not kept.)"""
        host = self.__dict__.get("_packed_host")
        if host is None or host.shape[0] < n:
            host = self.__dict__["_packed_host"] = torch.empty((max(1024, 2 * n), 6), dtype=torch.float32, pin_memory=True)
        return host

    def step_finish(self, st):
        """Second half of step(): waits for the frame's device -> host copy and runs the association on the host."""
        blob, outputs, features, hs_embeds = st["blob"], st["outputs"], st["features"], st["hs_embeds"]
        results, result, orig_size, device = st["results"], st["result"], st["orig_size"], st["device"]
        (orig_h, orig_w), num_prev_track = st["orig_hw"], st["num_prev_track"]
        if st["event"] is not None:
            st["event"].synchronize()
            packed = st["host"][:st["packed_dev"].shape[0]].clone()   # the pinned buffer is reused by the next frame
        else:
            packed = st["packed_dev"]
        boxes = packed[:, :4]
        scores = packed[:, 4]
        is_person = packed[:, 5] == 0

        nq = self.num_object_queries
        # `cur`: (positions [n, 4], scores [n]) of self.tracks in list order while that is known from the arrays at hand --
        # the NMS passes and the results then take them from there instead of stacking ~200 per-track tensors three more
        # times per frame; None = not known (any path that reorders / revives tracks), the consumers stack as before.
        cur = None
        # ---------------------------------------------------------------- existing tracks
        if num_prev_track:
            track_scores = scores[:-nq]
            track_boxes = boxes[:-nq]
            if 'masks' in result:
                track_masks = result['masks'][:-nq]
            if self.generate_attention_maps:
                track_attention_maps = self.attention_data['maps'][:-nq]

            track_keep = ((track_scores > self.track_obj_score_thresh) & is_person[:-nq]).tolist()
            # a track's new position / score / embedding are filed as (this frame's array, row) references: no tensor per
            # track and frame (Track.pos / .score build it when somebody reads one)
            to_inactive, from_inactive = [], []
            for i, track in enumerate(self.tracks):
                if track_keep[i]:
                    track._score = (track_scores, i)
                    track.hs_embed.append_row(hs_embeds, i)
                    track._pos = (track_boxes, i)
                    track.count_termination = 0
                    if 'masks' in result:
                        track.mask = track_masks[i]
                    if self.generate_attention_maps:
                        track.attention_map = track_attention_maps[i]
                else:
                    track.count_termination += 1
                    if track.count_termination >= self.steps_termination:
                        to_inactive.append(track)

            reid_keep = ((track_scores > self.reid_score_thresh) & is_person[:-nq]).tolist()
            for i, track in enumerate(self.inactive_tracks, start=len(self.tracks)):
                if reid_keep[i]:
                    track._score = (track_scores, i)
                    track.hs_embed.append_row(hs_embeds, i)
                    track._pos = (track_boxes, i)
                    if 'masks' in result:
                        track.mask = track_masks[i]
                    if self.generate_attention_maps:
                        track.attention_map = track_attention_maps[i]
                    from_inactive.append(track)

            if to_inactive:
                self._logger(
                    f'NEW INACTIVE TRACK IDS (track_obj_score_thresh={self.track_obj_score_thresh}): '
                    f'{[t.id for t in to_inactive]}')

            self.num_reids += len(from_inactive)
            for track in from_inactive:
                self.inactive_tracks.remove(track)
                self.tracks.append(track)
            self.tracks_to_inactive(to_inactive)

            if self.track_nms_thresh and self.tracks:
                pos_all = _gather_rows([t._pos for t in self.tracks])
                score_all = _gather_rows([t._score for t in self.tracks])
                cur = self._keep_after_nms(pos_all, score_all, score_all, self.track_nms_thresh, 'track_nms_thresh')

        # ---------------------------------------------------------------- new detections
        new_det_keep = (scores[-nq:] > self.detection_obj_score_thresh) & is_person[-nq:]
        new_det_indices = new_det_keep.float().nonzero()          # [K, 1] object-query indices
        sel = new_det_indices[:, 0]
        new_det_boxes = boxes[-nq:][sel]
        new_det_scores = scores[-nq:][sel]
        det_hs, det_masks, det_maps = hs_embeds[-nq:], None, None
        if 'masks' in result:
            det_masks = result['masks'][-nq:]
        if self.generate_attention_maps:
            det_maps = self.attention_data['maps'][-nq:]

        def narrow(m):
            return new_det_boxes[m], new_det_scores[m], new_det_indices[m], sel[m]

        # public detections (tracker.py:438-449)
        new_det_boxes, new_det_scores, new_det_indices, sel = narrow(
            self.public_detections_mask(new_det_boxes, blob['dets'][0]))

        # re-ID of inactive tracks (tracker.py:451-466); without inactive tracks reid() returns at once and the gathered
        # embeddings / masks / maps of the detections are never looked at
        if self.inactive_tracks and len(sel):
            gather = sel.to(device, non_blocking=True)
            reid_mask = self.reid(new_det_boxes, new_det_scores, det_hs[gather],
                                  None if det_masks is None else det_masks[gather],
                                  None if det_maps is None else det_maps[gather])
        else:
            reid_mask = self.reid(new_det_boxes, new_det_scores, None)
        new_det_boxes, new_det_scores, new_det_indices, sel = narrow(reid_mask)
        n_before_new = len(self.tracks)
        if cur is not None and cur[0].shape[0] != n_before_new:
            cur = None                                   # re-identification moved tracks back into the active list

        gather = sel.to(device, non_blocking=True) if (det_masks is not None or det_maps is not None) else None
        aux_results = None
        if self._verbose:
            if orig_size is None:
                orig_size = blob['orig_size'].detach().to(device)
            aux_results = [self.obj_detector_post['bbox'](out, orig_size)[0]
                           for out in outputs['aux_outputs']]
        first_obj_row = hs_embeds.shape[0] - nq          # the new tracks reference their rows of this frame's embeddings
        new_track_ids = self.add_tracks(
            new_det_boxes, new_det_scores, None, new_det_indices,
            None if det_masks is None else det_masks[gather],
            None if det_maps is None else det_maps[gather], aux_results,
            hs_rows=(hs_embeds, [first_obj_row + r for r in sel.tolist()]))

        if cur is not None:                               # the new tracks were appended in the detections' order
            cur = (torch.cat([cur[0], new_det_boxes]), torch.cat([cur[1], new_det_scores]))
        elif n_before_new == 0:
            cur = (new_det_boxes, new_det_scores)

        # ---------------------------------------------------------------- NMS new vs. existing
        if self.detection_nms_thresh and self.tracks:
            if cur is None:
                cur = (_gather_rows([t._pos for t in self.tracks]), _gather_rows([t._score for t in self.tracks]))
            rank = cur[1].clone()
            rank[:len(self.tracks) - len(new_track_ids)] = np.inf   # existing tracks (everything in front of the new ones)
            cur = self._keep_after_nms(cur[0], cur[1], rank, self.detection_nms_thresh, 'detection_nms_thresh')   # always win

        # ---------------------------------------------------------------- results
        label = None
        if 'masks' in result:
            label = self._label_map_fused(outputs, blob, orig_h, orig_w)   # (round 6) None: the module chain below
            if label is None:
                self._resolve_masks(outputs, results, orig_size, blob["size"].to(device))
        masks_host = None
        if label is not None and self.tracks:
            index_map = torch.arange(len(self.tracks), device=label.device, dtype=torch.int16)[:, None, None]
            track_masks = label[None] == index_map    # device-side per-track views (Track.mask keeps the reference's meaning)
            for i, track in enumerate(self.tracks):
                track.mask = track_masks[i]
            # the copy to the host is enqueued, not waited for.  (Measured and not kept: the mask head on a stream of its own, so that
            # the next frame's decoder half does not queue behind it -- 110.9 against 119.9 frames/s, three sequences 105.7 against
            # 129.5: the head's large launches next to the decoder half's small ones slow the chain more than the queueing did)
            masks_host = _LabelMap(label)
        elif 'masks' in result and self.tracks:
            # tracker.py:521-532 of the reference: a pixel belongs to the track with the largest probability there, if that
            # probability exceeds 0.5 -- `logical_and(probs > 0.5, index_map == probs.argmax(0))`.  Ownership is exclusive, so
            # ONE label map (owning track or -1 per pixel) holds every track's mask: 2 bytes per pixel reach the host instead
            # of one byte per pixel AND track (3.9 MB instead of ~290 MB per 1080 x 1800 frame with 150 tracks; cfg 5 went
            # from 40.0 to 21.1 ms per step together with the mask head's split-product route, profiles/r03_bench_cfg5.json ->
            # r04_bench_cfg5.json); the per-track boolean masks of the results are
            # rebuilt from it when somebody reads them (`results`), bit-identical.
            probs = torch.stack([t.mask for t in self.tracks])
            best, owner = probs.max(dim=0)            # ties: the first track, as argmax
            label = torch.where(best > 0.5, owner, torch.full_like(owner, -1)).to(torch.int16)
            index_map = torch.arange(probs.size(0), device=probs.device, dtype=torch.int16)[:, None, None]
            track_masks = label[None] == index_map    # device-side per-track views (Track.mask keeps the reference's meaning)
            for i, track in enumerate(self.tracks):
                track.mask = track_masks[i]
            masks_host = _LabelMap(label)

        if self.tracks:   # one stack + one numpy view for all tracks instead of three conversions per track
            if cur is None or cur[0].shape[0] != len(self.tracks):
                cur = (_gather_rows([t._pos for t in self.tracks]), _gather_rows([t._score for t in self.tracks]))
            all_pos = cur[0]
            if not self.obj_detector.overflow_boxes:
                all_pos = clip_boxes_to_image(all_pos, (orig_h, orig_w))
            all_pos = all_pos.numpy()
            all_scores = cur[1].numpy()
            # every live track now points into these two arrays: the next frame's track-query boxes and its NMS inputs are
            # one gather from one source again, whatever mix of updated / revived / new tracks this frame left behind
            pos_arr, score_arr = cur
            for i, t in enumerate(self.tracks):
                t._pos = (pos_arr, i)
                t._score = (score_arr, i)
        if self.tracks:
            # the frame's rows are filed as arrays; the reference's {track id: {frame: {...}}} layout is built from them when
            # somebody reads `results` / get_results() (per track and frame that is a dict and three objects: ~0.3 ms per
            # frame at 150 live tracks, off the per-frame path now)
            extras = None
            if any(t.mask is not None or t.attention_map is not None for t in self.tracks):
                extras = [(masks_host.of(i) if (masks_host is not None and t.mask is not None) else
                           (t.mask.cpu().numpy() if t.mask is not None else None),
                           t.attention_map.cpu().numpy() if t.attention_map is not None else None)
                          for i, t in enumerate(self.tracks)]
            self._pending_results.append((self.frame_index, [t.id for t in self.tracks], all_pos, all_scores,
                                          [t.obj_index for t in self.tracks], extras))

        for t in self.inactive_tracks:
            t.count_inactive += 1
        self.frame_index += 1
        if self.prev_frame_dist != 1 and getattr(self.obj_detector, "features_alias_static_buffers", False):
            # GraphedDetector hands back its static feature buffers (rewritten by every replay): with a distance of 1 they
            # are exactly what the next call wants back; older slots of the deque must own their data
            features = self.obj_detector._clone_features(features)
        self._prev_features.append(features)
        self._inflight_features = None
        if self.reid_sim_only:
            self.tracks_to_inactive(self.tracks)

    def _context_read(self, device):
        """The lazy mask head has just been enqueued: it reads the detector's aliased buffers (GraphedDetector: the slot the frame
        was decoded from) AFTER the detector call returned -- tell the wrapper, or the next-but-one frame's image-only half, prepared
        on the side stream, could overwrite them while the head is still running (nothing on the host waits for the head any more)."""
        fn = getattr(self.obj_detector, "state_read", None)
        if fn is not None:
            fn(device)

    def _label_map_fused(self, outputs, blob, orig_h, orig_w):
        """The frame's mask ownership map (int16 [orig_h, orig_w]: index into self.tracks or -1) straight from the mask head's
        low-resolution logits -- PostProcessSegm's bilinear resize / sigmoid / crop / nearest resize and the reference's per-pixel
        argmax over the tracks (tracker.py:521-532) in ONE launch (fused.mask_label_map) instead of ~10 passes over one full-size
        fp32 map per track.  Lazy mask head on the GPU with the package's own PostProcessSegm only; None otherwise."""
        post = self.obj_detector_post.get('segm')
        if (not self.tracks or 'pred_masks' in outputs or 'mask_context' not in outputs or post is None
                or type(post).__name__ != "PostProcessSegm" or not type(post).__module__.startswith("trackformer_amd.")
                or getattr(post, "threshold", None) != 0.5 or not outputs['hs_embed'].is_cuda
                or not all(isinstance(t.mask, _MaskRef) for t in self.tracks)):
            return None
        from . import fused
        if not fused._postprocess_fused:
            return None
        refs = sorted({t.mask.row for t in self.tracks})
        module = getattr(self.obj_detector, "model", self.obj_detector)
        hs = outputs['hs_embed']
        padded = refs + [refs[-1]] * (-len(refs) % 32)     # (as _resolve_masks: the convolution library tunes per shape)
        pos = {row: k for k, row in enumerate(refs)}
        size = blob["size"].detach().cpu().reshape(-1).tolist()                            # the (un-padded) size of this image
        idx = torch.tensor(padded, dtype=torch.long).to(hs.device, non_blocking=True)
        with torch.no_grad():
            rows = module.mask_rows(outputs['mask_context'], hs.index_select(1, idx))[0]   # [n_padded, H, W]
        self._context_read(hs.device)
        label = fused.mask_label_map(rows.contiguous(), [pos[t.mask.row] for t in self.tracks], (int(size[0]), int(size[1])),
                                     (int(size[0]), int(size[1])), (orig_h, orig_w))
        if label is None:
            return None
        for t in self.inactive_tracks:      # references of tracks that left the active set are never read
            if isinstance(t.mask, _MaskRef):
                t.mask = None
        return label

    def _resolve_masks(self, outputs, results, orig_size, size):
        """Turn the _MaskRef placeholders of this frame into mask probabilities at the original image size:
        PostProcessSegm (detr_segmentation.py:297-334 of the reference) on exactly the referenced queries."""
        refs = sorted({t.mask.row for t in self.tracks if isinstance(t.mask, _MaskRef)})
        if refs and 'pred_masks' not in outputs:
            # lazy mask head: evaluate it for the referenced queries now (the context aliases this frame's buffers)
            module = getattr(self.obj_detector, "model", self.obj_detector)
            hs = outputs['hs_embed']
            # the row count is rounded up to a multiple of 32 (the last row repeated): the convolution library tunes per
            # shape (seconds per new shape in find mode), and the number of live tracks changes from frame to frame
            padded = refs + [refs[-1]] * (-len(refs) % 32)
            idx = torch.tensor(padded, dtype=torch.long, device=hs.device)
            with torch.no_grad():
                rows = module.mask_rows(outputs['mask_context'], hs.index_select(1, idx))[:, :len(refs)]
            self._context_read(hs.device)
            seg = self.obj_detector_post['segm']([{}], {'pred_masks': rows}, orig_size, size,
                                                 return_probs=True)[0]['masks'].squeeze(dim=1)
            by_row = {row: seg[k] for k, row in enumerate(refs)}
            for t in self.tracks:
                if isinstance(t.mask, _MaskRef):
                    t.mask = by_row[t.mask.row]
        elif refs:
            n = outputs['pred_masks'].shape[1]
            keep = torch.zeros(n, dtype=torch.bool)
            keep[refs] = True
            keep = keep.to(outputs['pred_masks'].device)
            seg = self.obj_detector_post['segm']([{}], outputs, orig_size, size, return_probs=True,
                                                 results_mask=[keep])[0]['masks'].squeeze(dim=1)
            by_row = {row: seg[k] for k, row in enumerate(refs)}
            for t in self.tracks:
                if isinstance(t.mask, _MaskRef):
                    t.mask = by_row[t.mask.row]
        for t in self.inactive_tracks:      # references of tracks that left the active set are never read
            if isinstance(t.mask, _MaskRef):
                t.mask = None

    @property
    def results(self):
        """{track_id: {frame_idx: {'bbox': xyxy px, 'score', 'obj_ind', ['mask'], ['attention_map']}}} (tracker.py:523-541 of
        the reference): the frames filed since the last access are merged in here."""
        self._flush_deferred()
        pending, self._pending_results = self._pending_results, []
        for frame, ids, pos, scores, obj_inds, extras in pending:
            for i, tid in enumerate(ids):
                entry = self._results.setdefault(tid, {})[frame] = {}
                entry['bbox'] = pos[i].copy()
                entry['score'] = scores[i:i + 1].reshape(()).copy()   # 0-d array, as tensor.numpy().copy() gives
                entry['obj_ind'] = obj_inds[i]
                if extras is not None:
                    mask, amap = extras[i]
                    if mask is not None:
                        entry['mask'] = mask.materialise() if isinstance(mask, _LabelMask) else mask
                    if amap is not None:
                        entry['attention_map'] = amap
        return self._results

    @results.setter
    def results(self, value):
        self._results = value
        self._pending_results = []

    def get_results(self):
        """{track_id: {frame_idx: {'bbox': xyxy px, 'score', 'obj_ind', ['mask'], ['attention_map']}}}"""
        return self.results


def _guarded_state(name):
    """Tracker state that a deferred step() (see Tracker.step) may not have brought up to date yet: reading it from outside a
    step first runs the outstanding association."""
    key = "_state_" + name

    def get(self):
        d = self.__dict__
        if d.get("_deferred_handle") is not None and not d.get("_busy"):
            self._flush_deferred()
        try:
            return d[key]
        except KeyError:
            raise AttributeError("%r object has no attribute %r (set by reset())" % (type(self).__name__, name)) from None

    def put(self, value):
        d = self.__dict__
        if d.get("_deferred_handle") is not None and not d.get("_busy"):
            self._flush_deferred()   # (a write from outside -- re-seeding the tracks -- must not be overwritten by the older frame)
        d[key] = value
    return property(get, put)


for _name in ("tracks", "inactive_tracks", "track_num", "frame_index", "num_reids"):
    setattr(Tracker, _name, _guarded_state(_name))
del _name


class _LabelMap:
    """One frame's mask ownership on the host: label[y, x] = index (into that frame's track list) of the track that owns the
    pixel, -1 for none.  `of(i)` is a handle that turns into track i's boolean mask when the results are read.

    Round 6: built from the DEVICE map without waiting for it -- the copy into pinned host memory is enqueued behind the mask head
    on the tracker's stream and an event is recorded; `label` waits for that event when somebody first reads it (the results, at
    the end of the sequence in the reference's loop).  step_finish used to block on this copy: with ~100 live tracks the host sat
    idle for the whole mask head (~4 ms of a 9 ms cfg-5 frame) before it could enqueue the next frame's decoder half.  At most
    `_MAX_PINNED` frames keep their pinned buffer; older ones are moved to ordinary memory (their copies finished long ago)."""
    __slots__ = ("_label", "_pinned", "_event")
    _MAX_PINNED = 8
    _outstanding = None   # deque of maps that still hold a pinned buffer (process-wide: pinned memory is a shared resource)

    def __init__(self, label):
        self._pinned = self._event = None
        if torch.is_tensor(label) and label.is_cuda:
            with torch.cuda.device(label.device):
                self._pinned = torch.empty(label.shape, dtype=label.dtype, pin_memory=True)
                self._pinned.copy_(label, non_blocking=True)
                self._event = torch.cuda.Event()
                self._event.record(torch.cuda.current_stream(label.device))
            self._label = None
            if _LabelMap._outstanding is None:
                _LabelMap._outstanding = deque()
            q = _LabelMap._outstanding
            q.append(self)
            while len(q) > self._MAX_PINNED:
                q.popleft()._settle()
        else:
            self._label = label.cpu().numpy() if torch.is_tensor(label) else label

    def _settle(self):
        if self._label is None:
            self._event.synchronize()
            self._label = self._pinned.numpy().copy()
            self._pinned = self._event = None

    @property
    def label(self):
        if self._label is None:
            self._settle()
            try:
                _LabelMap._outstanding.remove(self)
            except (ValueError, AttributeError):
                pass
        return self._label

    def of(self, i):
        return _LabelMask(self, i)


class _LabelMask:
    __slots__ = ("map", "index")

    def __init__(self, label_map, index):
        self.map, self.index = label_map, index

    def materialise(self):
        return self.map.label == self.index     # bool [H, W], what `track_masks[i].cpu().numpy()` of the reference holds


class _MaskRef:
    """Placeholder for "the mask of query row `row` of the current frame" (see Tracker.step)."""
    __slots__ = ("row",)

    def __init__(self, row):
        self.row = int(row)


class _MaskRows:
    """Stands in for the [Q, H, W] mask tensor of a frame: indexing yields _MaskRef placeholders, with the
    slicing / index-tensor forms Tracker.step uses on the real tensor."""

    def __init__(self, n, rows=None):
        self.rows = list(range(n)) if rows is None else rows

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return _MaskRows(0, self.rows[idx])
        if torch.is_tensor(idx):
            if idx.dim() == 0:
                return _MaskRef(self.rows[int(idx)])
            return [_MaskRef(self.rows[i]) for i in idx.reshape(-1).tolist()]
        return _MaskRef(self.rows[int(idx)])


def _row(item):
    """A (frame array, row) reference or a tensor -> the tensor."""
    return item[0][item[1]] if type(item) is tuple else item


def _gather_rows(items):
    """[n, ...]: the rows a list of tensors / (frame array, row) references stands for, in list order -- torch.stack of the
    reference's per-track tensors.  References into ONE array are one gather; a few arrays (the tracks the previous frame
    updated, the ones it created, an inactive one from earlier) are one gather each and a permutation."""
    first = items[0]
    if type(first) is tuple:
        src, rows = first[0], []
        for it in items:
            if type(it) is not tuple or it[0] is not src:
                break
            rows.append(it[1])
        else:
            return src[rows]
    groups, plain, plain_at = {}, [], []
    for k, it in enumerate(items):
        if type(it) is tuple:
            g = groups.get(id(it[0]))
            if g is None:
                g = groups[id(it[0])] = (it[0], [], [])
            g[1].append(k)
            g[2].append(it[1])
        else:
            plain.append(it)
            plain_at.append(k)
    if len(groups) > 4:                         # many sources: nothing to gain over the plain stack
        return torch.stack([_row(it) for it in items])
    parts, at = [], []
    for src, ks, rows in groups.values():
        parts.append(src[rows])
        at += ks
    if plain:
        parts.append(torch.stack(plain))
        at += plain_at
    out = torch.cat(parts)
    inverse = [0] * len(at)
    for j, k in enumerate(at):
        inverse[k] = j
    return out[inverse]


class _LastPos(deque):
    """Track.last_pos: the deque of past positions of the reference (tracker.py:557-583).  The tracker files an entry per
    track and frame; it files the track's (frame boxes, row) reference as it is, and the [4] tensor is built when an entry
    is read."""

    def __getitem__(self, i):
        return _row(deque.__getitem__(self, i))

    def __iter__(self):
        return (_row(it) for it in deque.__iter__(self))

    def pop(self):
        return _row(deque.pop(self))

    def popleft(self):
        return _row(deque.popleft(self))


class _HsHistory(object):
    """Track.hs_embed: the output embeddings a track has had, newest last (a list of [C] device tensors in the reference,
    models/tracker.py:557-583; the newest one is the next frame's track query).  An entry filed by the tracker is a
    (frame embeddings [Q, C], row) reference: no per-track device view is created per frame (they were ~300 tensor objects
    per frame at 100 tracks + 100 detections), and the next frame's queries are ONE index_select when every track points
    into the same frame (Tracker._track_query_embeds).  Reading an entry gives the [C] row, as the reference's list does."""
    __slots__ = ("_items",)

    def __init__(self, first):
        self._items = [first]

    def append(self, row_tensor):
        self._items.append(row_tensor)

    def append_row(self, frame_embeds, row):
        self._items.append((frame_embeds, row))

    def __len__(self):
        return len(self._items)

    @staticmethod
    def _row(item):
        return item[0][item[1]] if type(item) is tuple else item

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._row(it) for it in self._items[i]]
        return self._row(self._items[i])

    def __iter__(self):
        return (self._row(it) for it in self._items)


class Track(object):
    """State of one track: last position / score (host tensors), the history of its output
    embeddings (device tensors, the newest one is the next frame's track query) and counters."""
    # slots for the fields every frame touches (100 tracks are created and ~200 updated per frame in bench.py's association
    # leg); __dict__ stays available for whatever a caller wants to hang on a track
    __slots__ = ("id", "_pos", "last_pos", "_score", "ims", "count_inactive", "count_termination", "gt_id", "hs_embed", "mask",
                 "attention_map", "_obj_ind", "_obj_index", "__dict__")

    def __init__(self, pos, score, track_id, hs_embed, obj_ind, mask=None, attention_map=None):
        self.id = track_id
        self._pos = pos              # a [4] tensor, or a (frame boxes [n, 4], row) reference: `pos` builds the tensor when read
        self.last_pos = _LastPos([pos])
        self._score = score          # a 0-d tensor, or a (frame scores [n], row) reference
        self.ims = deque([])
        self.count_inactive = 0
        self.count_termination = 0
        self.gt_id = None
        self.hs_embed = _HsHistory(hs_embed)   # a [C] tensor or a (frame embeddings, row) reference
        self.mask = mask
        self.attention_map = attention_map
        if type(obj_ind) is int:                 # the tracker's own tracks: no per-track index tensor until somebody asks
            self._obj_ind, self._obj_index = None, obj_ind
        else:
            self._obj_ind, self._obj_index = obj_ind, None

    @property
    def pos(self):
        """Last position, xyxy in pixels of the original image: the [4] host tensor of the reference."""
        v = self._pos
        if type(v) is tuple:
            v = self._pos = v[0][v[1]]
        return v

    @pos.setter
    def pos(self, value):
        self._pos = value

    @property
    def score(self):
        v = self._score
        if type(v) is tuple:
            v = self._score = v[0][v[1]]
        return v

    @score.setter
    def score(self, value):
        self._score = value

    @property
    def obj_ind(self):
        """Index of the object query that started the track: the [1] int64 tensor the reference keeps (tracker.py:557-583)."""
        if self._obj_ind is None:
            self._obj_ind = torch.tensor([self._obj_index], dtype=torch.int64)
        return self._obj_ind

    @obj_ind.setter
    def obj_ind(self, value):
        self._obj_ind, self._obj_index = value, None

    @property
    def obj_index(self) -> int:
        """obj_ind as a Python int (computed once; the results of every frame carry it)."""
        v = self._obj_index
        if v is None:
            v = self._obj_index = int(torch.as_tensor(self._obj_ind).reshape(-1)[0])
        return v

    def has_positive_area(self) -> bool:
        x0, y0, x1, y1 = self.pos.tolist()
        return x1 > x0 and y1 > y0

    def reset_last_pos(self) -> None:
        self.last_pos.clear()
        self.last_pos.append(self._pos)
