"""Deformable-DETR encoder/decoder with TrackFormer's track-query prepending.

Same classes, constructor arguments, parameter names and forward contracts as the reference's
models/deformable_transformer.py (DeformableTransformer :21-255, EncoderLayer :258-297,
Encoder :300-327, DecoderLayer :330-383, Decoder :386-431, build_deforamble_transformer :434-454).

What is different is host-side only: per-frame constants that depend just on the pyramid geometry
(the `spatial_shapes` device tensor, valid ratios of an unpadded batch, encoder reference points)
are cached instead of being rebuilt with dozens of tiny kernels every frame, and the host copy of
the level shapes travels with the `spatial_shapes` tensor so the MSDeformAttn operator never has to
read it back from the device.
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

from . import fused
from .msda import MSDeformAttn, attach_host_shapes
from .nested import inverse_sigmoid, is_all_valid


def _inference(module) -> bool:
    """eval mode with gradients disabled: the fused single-pass kernels may replace ATen chains."""
    return (not module.training) and (not torch.is_grad_enabled())


def _ffn_hidden(linear, activation, x, inference):
    """activation(linear(x)).  Inference on the GPU with ReLU: bias + ReLU run in the GEMM epilogue
    (hipBLASLt) instead of a separate pass over the [tokens, d_ffn] activation."""
    if inference and activation is F.relu and x.is_cuda and x.dtype == torch.float32:
        if fused.split_linear_enabled():   # opt-in: bf16 split product on the matrix cores, ReLU in its epilogue
            y = fused.linear(x, linear.weight, linear.bias, relu=True)
            if y is not None:
                return y
        x2 = x.reshape(-1, x.shape[-1])
        y = torch._addmm_activation(linear.bias, x2, linear.weight.t(), use_gelu=False)
        return y.view(*x.shape[:-1], y.shape[-1])
    return activation(linear(x))


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _get_activation_fn(activation):
    """String -> activation function (models/transformer.py:300-308)."""
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


class _GeometryCache:
    """Device constants keyed by (level shapes, batch, device): spatial_shapes tensor, all-ones valid
    ratios and the encoder's reference points for an unpadded batch."""

    def __init__(self):
        self._shapes = {}
        self._ones = {}
        self._enc_ref = {}

    def spatial_shapes(self, shapes, device):
        key = (tuple(shapes), device)
        t = self._shapes.get(key)
        if t is None:
            t = torch.as_tensor(list(shapes), dtype=torch.long, device=device)
            self._shapes[key] = attach_host_shapes(t, shapes)
        return t

    def unit_valid_ratios(self, bs, n_levels, device):
        key = (bs, n_levels, device)
        t = self._ones.get(key)
        if t is None:
            t = self._ones[key] = torch.ones(bs, n_levels, 2, dtype=torch.float32, device=device)
            t._tf_unit = True   # (consumers skip the multiplication by exactly 1.0: DeformableTransformerDecoder.forward)
        return t

    def encoder_reference_points(self, shapes, valid_ratios, unit_ratios):
        if not unit_ratios or torch.is_grad_enabled():
            return DeformableTransformerEncoder.get_reference_points(shapes, valid_ratios,
                                                                    valid_ratios.device)
        key = (tuple(shapes), valid_ratios.shape[0], valid_ratios.device)
        t = self._enc_ref.get(key)
        if t is None:
            t = self._enc_ref[key] = DeformableTransformerEncoder.get_reference_points(
                shapes, valid_ratios, valid_ratios.device)
        return t


_GEOMETRY = _GeometryCache()

# per-module inference caches (_ref_points_cache, _lvl_pos_cache): a HIP graph captured after a cache hit has the cached
# tensor's raw address baked in, so an entry must never be freed while such a graph can still be replayed.  The caches
# are therefore keyed (one entry per geometry / parameter version) and never evict: once this many entries exist, new
# keys are recomputed per call (inside a capture that lands in the graph's own pool) instead of displacing an old one.
_KEEP_CACHE_ENTRIES = 64


def _host_shapes(spatial_shapes):
    """(H, W) python ints of a spatial_shapes tensor (attached by DeformableTransformer.forward;
    falls back to one device read for foreign callers)."""
    hs = getattr(spatial_shapes, "_tf_msda_host_shapes", None)
    if hs is None:
        hs = tuple((int(h), int(w)) for h, w in spatial_shapes.tolist())
        attach_host_shapes(spatial_shapes, hs)
    return hs


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=1024, dropout=0.1, activation="relu",
                 return_intermediate_dec=False, num_feature_levels=4, dec_n_points=4,
                 enc_n_points=4, two_stage=False, two_stage_num_proposals=300,
                 multi_frame_attention_separate_encoder=False):
        super().__init__()
        self.d_model = d_model
        self.nhead = nhead
        self.two_stage = two_stage
        self.two_stage_num_proposals = two_stage_num_proposals
        self.num_feature_levels = num_feature_levels
        self.multi_frame_attention_separate_encoder = multi_frame_attention_separate_encoder

        enc_levels = num_feature_levels // 2 if multi_frame_attention_separate_encoder \
            else num_feature_levels
        encoder_layer = DeformableTransformerEncoderLayer(
            d_model, dim_feedforward, dropout, activation, enc_levels, nhead, enc_n_points)
        self.encoder = DeformableTransformerEncoder(encoder_layer, num_encoder_layers)
        decoder_layer = DeformableTransformerDecoderLayer(
            d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, dec_n_points)
        self.decoder = DeformableTransformerDecoder(decoder_layer, num_decoder_layers,
                                                    return_intermediate_dec)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        if two_stage:
            self.enc_output = nn.Linear(d_model, d_model)
            self.enc_output_norm = nn.LayerNorm(d_model)
            self.pos_trans = nn.Linear(d_model * 2, d_model * 2)
            self.pos_trans_norm = nn.LayerNorm(d_model * 2)
        else:
            self.reference_points = nn.Linear(d_model, 2)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        if not self.two_stage:
            xavier_uniform_(self.reference_points.weight.data, gain=1.0)
            constant_(self.reference_points.bias.data, 0.)
        normal_(self.level_embed)

    # ------------------------------------------------------------------ two-stage helpers
    def get_proposal_pos_embed(self, proposals):
        """sine embedding of (logit) proposal boxes: [N, K, 4] -> [N, K, 512]."""
        num_pos_feats, temperature, scale = 128, 10000, 2 * math.pi
        i = torch.arange(num_pos_feats, dtype=torch.float32, device=proposals.device)
        dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
        pos = (proposals.sigmoid() * scale)[:, :, :, None] / dim_t
        return torch.stack((pos[..., 0::2].sin(), pos[..., 1::2].cos()), dim=4).flatten(2)

    def gen_encoder_output_proposals(self, memory, memory_padding_mask, spatial_shapes):
        """Per-token box proposals (logit space) + gated/normalised memory for the two-stage variant."""
        n = memory.shape[0]
        proposals = []
        cur = 0
        for lvl, (h, w) in enumerate(_host_shapes(spatial_shapes)):
            m = memory_padding_mask[:, cur:cur + h * w].view(n, h, w, 1)
            valid_h = torch.sum(~m[:, :, 0, 0], 1)
            valid_w = torch.sum(~m[:, 0, :, 0], 1)
            gy, gx = torch.meshgrid(
                torch.linspace(0, h - 1, h, dtype=torch.float32, device=memory.device),
                torch.linspace(0, w - 1, w, dtype=torch.float32, device=memory.device),
                indexing="ij")
            grid = torch.stack([gx, gy], -1)
            scale = torch.stack([valid_w, valid_h], 1).view(n, 1, 1, 2)
            grid = (grid[None].expand(n, -1, -1, -1) + 0.5) / scale
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
            proposals.append(torch.cat((grid, wh), -1).view(n, -1, 4))
            cur += h * w
        out_prop = torch.cat(proposals, 1)
        valid = ((out_prop > 0.01) & (out_prop < 0.99)).all(-1, keepdim=True)
        out_prop = torch.log(out_prop / (1 - out_prop))
        out_prop = out_prop.masked_fill(memory_padding_mask.unsqueeze(-1), float('inf'))
        out_prop = out_prop.masked_fill(~valid, float('inf'))
        out_mem = memory.masked_fill(memory_padding_mask.unsqueeze(-1), float(0))
        out_mem = out_mem.masked_fill(~valid, float(0))
        return self.enc_output_norm(self.enc_output(out_mem)), out_prop

    def get_valid_ratio(self, mask):
        """fraction of un-padded (width, height) per sample: [N, 2]."""
        _, H, W = mask.shape
        valid_h = torch.sum(~mask[:, :, 0], 1)
        valid_w = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_w.float() / W, valid_h.float() / H], -1)

    # ------------------------------------------------------------------ forward
    def _object_reference_points(self, query_param, query_embed, bs):
        """sigmoid(reference_points(query_embed)) of the object queries (deformable_transformer.py:199 of the reference).  In
        inference it depends on parameters only: kept until one of them changes."""
        if self.training or torch.is_grad_enabled():
            return self.reference_points(query_embed).sigmoid()
        lin = self.reference_points
        key = (id(query_param), query_param._version, query_param.data_ptr(), lin.weight._version, lin.weight.data_ptr(),
               lin.bias._version, bs, query_embed.device)
        cache = self.__dict__.setdefault("_ref_points_cache", {})
        hit = cache.get(key)
        if hit is None:
            value = lin(query_embed).sigmoid()
            if (query_embed.is_cuda and torch.cuda.is_current_stream_capturing()) or len(cache) >= _KEEP_CACHE_ENTRIES:
                return value   # never keep a buffer of a graph's memory pool; a full cache recomputes, it never evicts
            hit = (value, query_param)
            if value.is_cuda:
                fused._publish_barrier(value.device)   # built on this stream, read by every lane's stream
            cache[key] = hit
        return hit[0]

    def _level_position_embedding(self, pos_embeds):
        """cat_l(pos_l flattened + level_embed[l]) (deformable_transformer.py:139-156 of the reference).  In inference the
        position encodings of an unpadded frame are cached tensors (position_encoding.py), so the result is a per-geometry
        constant: it is kept as long as the very same tensors (and the same level_embed contents) come in."""
        def compute():
            return torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1)
                              for lvl, p in enumerate(pos_embeds)], 1)
        if self.training or torch.is_grad_enabled():
            return compute()
        # Only the position-encoding module's own CACHED tensors (all-valid masks: one tensor per geometry, position_encoding.py
        # marks them) are worth keying on: a padded batch gets fresh tensors on every call, each of which would pin its
        # [N, S, C] result and its inputs for ever without being hit again (~90 MB per entry at batch 2, 800 x 1333).
        if not all(getattr(p, "_tf_cached_geometry", False) for p in pos_embeds):
            return compute()
        key = tuple((id(p), p._version) for p in pos_embeds) + (self.level_embed._version, self.level_embed.data_ptr())
        cache = self.__dict__.setdefault("_lvl_pos_cache", {})
        hit = cache.get(key)
        if hit is None:
            if (pos_embeds[0].is_cuda and torch.cuda.is_current_stream_capturing()) or len(cache) >= _KEEP_CACHE_ENTRIES:
                return compute()   # never keep a buffer of a graph's memory pool; a full cache recomputes, it never evicts
            # the inputs are kept alive next to the result: an id() in the key can then not be reused by another tensor
            hit = (compute(), list(pos_embeds))
            if pos_embeds[0].is_cuda:
                fused._publish_barrier(pos_embeds[0].device)   # built on this stream, read by every lane's stream
            cache[key] = hit
        return hit[0]

    def forward(self, srcs, masks, pos_embeds, query_embed=None, targets=None):
        return self.decode(self.encode(srcs, masks, pos_embeds), query_embed, targets)

    def encode(self, srcs, masks, pos_embeds):
        """The part of forward() that depends on the IMAGE only (deformable_transformer.py:133-173 of the reference: flatten,
        level embedding, encoder) -> the state decode() needs.  Tracker.step_prepare runs it for frame t + 1 while the host
        still associates frame t: the track queries enter in decode()."""
        shapes = tuple((int(s.shape[2]), int(s.shape[3])) for s in srcs)
        device = srcs[0].device
        src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        lvl_pos_embed_flatten = self._level_position_embedding(pos_embeds)
        unpadded = all(is_all_valid(m) for m in masks)
        if unpadded:
            # nothing is padded: the flattened mask is all False -> skip it entirely (masked_fill with
            # an all-False mask is the identity) and the valid ratios are exactly 1
            mask_flatten = None
            valid_ratios = _GEOMETRY.unit_valid_ratios(src_flatten.shape[0], len(shapes), device)
        else:
            mask_flatten = torch.cat([m.flatten(1) for m in masks], 1)
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
        spatial_shapes = _GEOMETRY.spatial_shapes(shapes, device)

        # encoder
        if self.multi_frame_attention_separate_encoder:
            half_t = src_flatten.shape[1] // 2
            half_l = self.num_feature_levels // 2

            def run(sl_t, sl_l):
                return self.encoder(
                    src_flatten[:, sl_t], _GEOMETRY.spatial_shapes(shapes[sl_l], device),
                    valid_ratios[:, sl_l], lvl_pos_embed_flatten[:, sl_t],
                    None if mask_flatten is None else mask_flatten[:, sl_t], unit_ratios=unpadded)
            prev_memory = run(slice(0, half_t), slice(0, half_l))
            memory = run(slice(half_t, None), slice(half_l, None))
            memory = torch.cat([memory, prev_memory], 1)
        else:
            memory = self.encoder(src_flatten, spatial_shapes, valid_ratios, lvl_pos_embed_flatten,
                                  mask_flatten, unit_ratios=unpadded)

        return {"memory": memory, "spatial_shapes": spatial_shapes, "valid_ratios": valid_ratios,
                "mask_flatten": mask_flatten, "device": device}

    def decode(self, enc, query_embed=None, targets=None):
        """The rest of forward() (deformable_transformer.py:175-246 of the reference): object + track queries, decoder."""
        assert self.two_stage or query_embed is not None
        memory, spatial_shapes, valid_ratios = enc["memory"], enc["spatial_shapes"], enc["valid_ratios"]
        mask_flatten, device = enc["mask_flatten"], enc["device"]
        # decoder inputs
        bs, _, c = memory.shape
        query_attn_mask = None
        filler_key_mask = None
        enc_outputs_class = enc_outputs_coord_unact = None
        if self.two_stage:
            pad = mask_flatten if mask_flatten is not None else \
                torch.zeros(memory.shape[:2], dtype=torch.bool, device=device)
            output_memory, output_proposals = self.gen_encoder_output_proposals(
                memory, pad, spatial_shapes)
            enc_outputs_class = self.decoder.class_embed[self.decoder.num_layers](output_memory)
            enc_outputs_coord_unact = \
                self.decoder.bbox_embed[self.decoder.num_layers](output_memory) + output_proposals
            topk = self.two_stage_num_proposals
            topk_proposals = torch.topk(enc_outputs_class[..., 0], topk, dim=1)[1]
            topk_coords_unact = torch.gather(
                enc_outputs_coord_unact, 1, topk_proposals.unsqueeze(-1).repeat(1, 1, 4)).detach()
            reference_points = topk_coords_unact.sigmoid()
            init_reference_out = reference_points
            pos_trans_out = self.pos_trans_norm(
                self.pos_trans(self.get_proposal_pos_embed(topk_coords_unact)))
            query_embed, tgt = torch.split(pos_trans_out, c, dim=2)
        else:
            query_param = query_embed
            query_embed, tgt = torch.split(query_embed, c, dim=1)
            query_embed = query_embed.unsqueeze(0).expand(bs, -1, -1)
            tgt = tgt.unsqueeze(0).expand(bs, -1, -1)
            reference_points = self._object_reference_points(query_param, query_embed, bs)

            if targets is not None and 'track_query_hs_embeds' in targets[0]:
                # TrackFormer: track queries go FIRST; their content is the previous frame's output
                # embedding, their positional query is zero and their reference point is the centre
                # of the previous box (deformable_transformer.py:202-225).
                prev_hs_embed = torch.stack([t['track_query_hs_embeds'] for t in targets])
                prev_boxes = torch.stack([t['track_query_boxes'] for t in targets])
                query_embed = torch.cat([torch.zeros_like(prev_hs_embed), query_embed], dim=1)
                tgt = torch.cat([prev_hs_embed, tgt], dim=1)
                reference_points = torch.cat([prev_boxes[..., :2], reference_points], dim=1)
                if 'track_query_filler' in targets[0]:   # see DeformableTransformerDecoderLayer.forward
                    filler = torch.stack([t['track_query_filler'] for t in targets])
                    filler_key_mask = torch.cat(
                        [filler, torch.zeros((bs, tgt.shape[1] - filler.shape[1]), dtype=torch.bool,
                                             device=filler.device)], dim=1)
            init_reference_out = reference_points

        hs, inter_references = self.decoder(tgt, reference_points, memory, spatial_shapes,
                                            valid_ratios, query_embed, mask_flatten,
                                            query_attn_mask, filler_key_mask)
        if self.two_stage:
            return (hs, memory, init_reference_out, inter_references, enc_outputs_class,
                    enc_outputs_coord_unact)
        return hs, memory, init_reference_out, inter_references, None, None


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4,
                 n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src):
        inf = _inference(self)
        if inf and fused.ffn_fused_enabled() and self.activation is F.relu:   # opt-in: the whole block in one launch
            out = fused.ffn(src, self.linear1, self.linear2, self.norm2, residual=src)
            if out is not None:
                return out
        src2 = fused.module_linear(self.linear2, self.dropout2(_ffn_hidden(self.linear1, self.activation, src, inf)), inf)
        return fused.residual_norm(src, self.dropout3(src2), self.norm2, inf)

    def forward(self, src, pos, reference_points, spatial_shapes, padding_mask=None):
        inf = _inference(self)
        # opt-in (fused.set_pos_add_fused): the positional add rides in the attention's projection GEMM
        q, q_pos = (src, pos) if (inf and pos is not None and fused.pos_add_fused_enabled()) else (self.with_pos_embed(src, pos), None)
        if inf and fused.linear_ln_fused_enabled():   # opt-in: output projection + add + norm1 in one launch
            src = self.self_attn(q, reference_points, src, spatial_shapes, padding_mask,
                                 residual_norm=(src, self.norm1), query_pos=q_pos)
            return self.forward_ffn(src)
        src2 = self.self_attn(q, reference_points, src, spatial_shapes, padding_mask, query_pos=q_pos)
        src = fused.residual_norm(src, self.dropout1(src2), self.norm1, _inference(self))
        return self.forward_ffn(src)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel centres of every level in normalised valid-image coordinates, broadcast over the
        levels they will be sampled in: [N, S, L, 2] (deformable_transformer.py:307-319)."""
        if torch.is_tensor(spatial_shapes):
            spatial_shapes = _host_shapes(spatial_shapes)
        per_level = []
        for lvl, (h, w) in enumerate(spatial_shapes):
            ys = torch.linspace(0.5, h - 0.5, h, dtype=torch.float32, device=device)
            xs = torch.linspace(0.5, w - 0.5, w, dtype=torch.float32, device=device)
            ref_y, ref_x = torch.meshgrid(ys, xs, indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
            per_level.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(per_level, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, valid_ratios, pos=None, padding_mask=None,
                unit_ratios=False):
        reference_points = _GEOMETRY.encoder_reference_points(
            _host_shapes(spatial_shapes), valid_ratios, unit_ratios)
        output = src
        for layer in self.layers:
            output = layer(output, pos, reference_points, spatial_shapes, padding_mask)
        return output


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4,
                 n_heads=8, n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        inf = _inference(self)
        if inf and fused.ffn_fused_enabled() and self.activation is F.relu:   # opt-in; only with many rows (fused.ffn)
            out = fused.ffn(tgt, self.linear1, self.linear2, self.norm3, residual=tgt)
            if out is not None:
                return out
        tgt2 = fused.module_linear(self.linear2, self.dropout3(_ffn_hidden(self.linear1, self.activation, tgt, inf)), inf)
        return fused.residual_norm(tgt, self.dropout4(tgt2), self.norm3, inf)

    def _self_attention_inference(self, qk_in, v_in, key_padding_mask, residual_norm=None, qk_pos=None):
        """nn.MultiheadAttention(q=k=qk_in, v=v_in) for batch-first inputs without materialising the
        attention weights: one GEMM for the shared q/k input, one for v, fused SDPA, out_proj.
        residual_norm = (residual, norm): returns (norm(residual + attention), True) when the opt-in one-launch
        projection + add + LayerNorm applied, else (attention, False)."""
        mha = self.self_attn
        E, H = mha.embed_dim, mha.num_heads
        w, b = mha.in_proj_weight, mha.in_proj_bias
        n, lq, _ = qk_in.shape
        qk = None
        if qk_pos is not None:   # opt-in: q = k = qk_in + qk_pos, the add inside the GEMM
            qk = fused.linear_add(qk_in, qk_pos, w, b[:2 * E], rows=(0, 2 * E))
            if qk is None:
                qk_in = qk_in + qk_pos
        if qk is None:
            qk = fused.linear(qk_in, w, b[:2 * E], rows=(0, 2 * E))    # split product on the matrix cores when enabled
        if qk is None:
            qk = F.linear(qk_in, w[:2 * E], b[:2 * E])
        v = fused.linear(v_in, w, b[2 * E:], rows=(2 * E, 3 * E))
        if v is None:
            v = F.linear(v_in, w[2 * E:], b[2 * E:])
        o = fused.mha_core(qk, v, H, key_padding_mask)      # one fp32 launch: scores, softmax, P V
        if o is not None:
            if residual_norm is not None and fused.linear_ln_fused_enabled():
                y = fused.linear_residual_norm(o, mha.out_proj, residual_norm[0], residual_norm[1])
                if y is not None:
                    return y, True
            return fused.module_linear(mha.out_proj, o, True), False
        qk = qk.view(n, lq, 2, H, E // H)
        v = v.view(n, lq, H, E // H)
        q, k = qk[:, :, 0].transpose(1, 2), qk[:, :, 1].transpose(1, 2)      # [n, H, lq, d]
        mask = None
        if key_padding_mask is not None:
            mask = ~key_padding_mask[:, None, None, :]                           # True = attend
        o = F.scaled_dot_product_attention(q, k, v.transpose(1, 2), attn_mask=mask)
        return mha.out_proj(o.transpose(1, 2).reshape(n, lq, E)), False

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes,
                src_padding_mask=None, query_attn_mask=None, filler_key_mask=None):
        """filler_key_mask [N, Lq] (True = ignore as a KEY of the self-attention only): marks filler track queries
        that GraphedDetector appends to reach a bucketed query count; their own outputs are dropped by the caller, so
        nothing else needs to know about them (the cross-attention is per query)."""
        # self attention among the (track + object) queries
        key_mask = query_attn_mask if query_attn_mask is not None else filler_key_mask
        if _inference(self) and tgt.is_cuda and self.self_attn.in_proj_weight is not None:
            if query_pos is not None and fused.pos_add_fused_enabled():
                tgt2, normed = self._self_attention_inference(tgt, tgt, key_mask, residual_norm=(tgt, self.norm2), qk_pos=query_pos)
            else:
                tgt2, normed = self._self_attention_inference(self.with_pos_embed(tgt, query_pos), tgt, key_mask,
                                                              residual_norm=(tgt, self.norm2))
        else:
            normed = False
            q = k = self.with_pos_embed(tgt, query_pos)
            tgt2 = self.self_attn(q.transpose(0, 1), k.transpose(0, 1), tgt.transpose(0, 1),
                                  key_padding_mask=key_mask)[0].transpose(0, 1)
        tgt = tgt2 if normed else fused.residual_norm(tgt, self.dropout2(tgt2), self.norm2, _inference(self))
        # deformable cross attention into the encoder memory
        inf = _inference(self)
        cq, cq_pos = (tgt, query_pos) if (inf and query_pos is not None and fused.pos_add_fused_enabled()) \
            else (self.with_pos_embed(tgt, query_pos), None)
        if inf and fused.linear_ln_fused_enabled():   # opt-in, as in the encoder layer
            tgt = self.cross_attn(cq, reference_points, src, src_spatial_shapes, src_padding_mask, query_attn_mask,
                                  residual_norm=(tgt, self.norm1), query_pos=cq_pos)
            return self.forward_ffn(tgt)
        tgt2 = self.cross_attn(cq, reference_points, src, src_spatial_shapes, src_padding_mask, query_attn_mask,
                               query_pos=cq_pos)
        tgt = fused.residual_norm(tgt, self.dropout1(tgt2), self.norm1, _inference(self))
        return self.forward_ffn(tgt)


class DeformableTransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        # set by DeformableDETR for iterative box refinement / two-stage
        self.bbox_embed = None
        self.class_embed = None

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_valid_ratios,
                query_pos=None, src_padding_mask=None, query_attn_mask=None, filler_key_mask=None):
        output = tgt
        intermediate = []
        intermediate_reference_points = []
        unit = getattr(src_valid_ratios, "_tf_unit", False)   # all-valid masks: the ratios are the cached tensor of ones
        for lid, layer in enumerate(self.layers):
            if unit:
                # x * 1.0 is x: the reference's concatenation + multiplication (deformable_transformer.py:343-348) are a broadcast view
                reference_points_input = reference_points[:, :, None].expand(-1, -1, src_valid_ratios.shape[1], -1)
            elif reference_points.shape[-1] == 4:
                reference_points_input = reference_points[:, :, None] \
                    * torch.cat([src_valid_ratios, src_valid_ratios], -1)[:, None]
            else:
                assert reference_points.shape[-1] == 2
                reference_points_input = reference_points[:, :, None] * src_valid_ratios[:, None]
            output = layer(output, query_pos, reference_points_input, src, src_spatial_shapes,
                           src_padding_mask, query_attn_mask, filler_key_mask)

            if self.bbox_embed is not None:  # iterative bounding box refinement
                tmp = self.bbox_embed[lid](output)
                fused_ref = (fused.box_refine(tmp, reference_points)   # opt-in; None = not applicable / off
                             if not self.training and not torch.is_grad_enabled() and tmp.is_cuda else None)
                if fused_ref is not None:
                    new_reference_points = fused_ref
                elif reference_points.shape[-1] == 4:
                    new_reference_points = (tmp + inverse_sigmoid(reference_points)).sigmoid()
                else:
                    assert reference_points.shape[-1] == 2
                    new_reference_points = tmp
                    new_reference_points[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points)
                    new_reference_points = new_reference_points.sigmoid()
                reference_points = new_reference_points.detach()

            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)

        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(intermediate_reference_points)
        return output, reference_points


def build_deforamble_transformer(args):  # [sic] -- the reference's spelling, kept for drop-in use
    num_feature_levels = args.num_feature_levels
    if args.multi_frame_attention:
        num_feature_levels *= 2
    return DeformableTransformer(
        d_model=args.hidden_dim,
        nhead=args.nheads,
        num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers,
        dim_feedforward=args.dim_feedforward,
        dropout=args.dropout,
        activation="relu",
        return_intermediate_dec=True,
        num_feature_levels=num_feature_levels,
        dec_n_points=args.dec_n_points,
        enc_n_points=args.enc_n_points,
        two_stage=args.two_stage,
        two_stage_num_proposals=args.num_queries,
        multi_frame_attention_separate_encoder=(args.multi_frame_attention
                                                and args.multi_frame_attention_separate_encoder))


build_deformable_transformer = build_deforamble_transformer
