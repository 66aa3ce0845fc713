"""Training-time track-query augmentation (host side).

Restates DETRTrackingBase.add_track_queries_to_targets of the reference
(models/detr_tracking.py:39-183): from the previous frame's detections and their Hungarian
assignment it picks a random subset of true tracks (false negatives are dropped), adds random false
positive queries sampled near real ones, and writes into each target dict
    track_query_match_ids, track_query_hs_embeds, track_query_boxes,
    track_queries_mask, track_queries_fal_pos_mask.
The order of torch RNG calls (randint, randint, randperm per sample, randperm + multinomial/randperm
per false positive) is the reference's, so a seeded run draws the same queries.
"""
import math

import torch


def add_track_queries_to_targets(targets, prev_indices, prev_out, add_false_pos=True, *,
                                 false_positive_prob=0.0, false_negative_prob=0.0, num_queries):
    device = prev_out['pred_boxes'].device
    num_prev_queries = prev_out['pred_boxes'].shape[1]

    # one subset size shared by the whole batch so that track queries stack (torch.stack in
    # deformable_transformer.py:211-212 needs equal counts)
    min_prev_target_ind = min(len(prev_ind[1]) for prev_ind in prev_indices)
    num_prev_target_ind = 0
    if min_prev_target_ind:
        num_prev_target_ind = torch.randint(0, min_prev_target_ind + 1, (1,)).item()
    num_prev_target_ind_for_fps = 0
    if num_prev_target_ind:
        num_prev_target_ind_for_fps = torch.randint(
            int(math.ceil(false_positive_prob * num_prev_target_ind)) + 1, (1,)).item()

    for i, (target, (prev_out_ind, prev_target_ind)) in enumerate(zip(targets, prev_indices)):
        if false_negative_prob:
            subset = torch.randperm(len(prev_target_ind))[:num_prev_target_ind]
            prev_out_ind = prev_out_ind[subset]
            prev_target_ind = prev_target_ind[subset]

        # which of the kept previous-frame tracks are still present in the current frame
        prev_track_ids = target['prev_target']['track_ids'][prev_target_ind]
        match_matrix = prev_track_ids.unsqueeze(dim=1).eq(target['track_ids'])
        target_ind_matching = match_matrix.any(dim=1)
        target['track_query_match_ids'] = match_matrix.nonzero()[:, 1]

        if add_false_pos:
            # (the matcher's indices live on the host, the id comparison on the device: PyTorch >= 1.8
            # no longer mixes them implicitly as the reference's line 97 relies on)
            prev_boxes_matched = prev_out['pred_boxes'][
                i, prev_out_ind[target_ind_matching.to(prev_out_ind.device)]]
            free = torch.ones(num_prev_queries, dtype=torch.bool)
            free[prev_out_ind.cpu()] = False
            not_prev_out_ind = free.nonzero()[:, 0].tolist()

            random_false_out_ind = []
            for j in torch.randperm(num_prev_target_ind)[:num_prev_target_ind_for_fps]:
                prev_boxes_unmatched = prev_out['pred_boxes'][i, not_prev_out_ind]
                if len(prev_boxes_matched) > j:
                    d = prev_boxes_matched[j].unsqueeze(dim=0)[:, :2] - prev_boxes_unmatched[:, :2]
                    # NB: the reference squares the x distance twice (detr_tracking.py:124); kept
                    box_weights = torch.sqrt(d[:, 0] ** 2 + d[:, 0] ** 2)
                    pick = torch.multinomial(box_weights.cpu(), 1).item()
                else:
                    pick = torch.randperm(len(not_prev_out_ind))[0]
                random_false_out_ind.append(not_prev_out_ind.pop(pick))

            prev_out_ind = torch.tensor(prev_out_ind.tolist() + random_false_out_ind).long()
            target_ind_matching = torch.cat([
                target_ind_matching,
                torch.zeros(len(random_false_out_ind), dtype=torch.bool, device=device)])

        track_queries_mask = torch.ones_like(target_ind_matching).bool()
        track_queries_fal_pos_mask = ~target_ind_matching

        target['track_query_hs_embeds'] = prev_out['hs_embed'][i, prev_out_ind]
        target['track_query_boxes'] = prev_out['pred_boxes'][i, prev_out_ind].detach()
        no_obj = torch.zeros(num_queries, dtype=torch.bool, device=device)
        target['track_queries_mask'] = torch.cat([track_queries_mask, no_obj]).bool()
        target['track_queries_fal_pos_mask'] = torch.cat([track_queries_fal_pos_mask, no_obj]).bool()
