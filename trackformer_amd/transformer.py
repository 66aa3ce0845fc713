"""Dense (vanilla DETR) transformer used by the plain-DETR configuration (BASELINE cfg 1).

Same classes, constructor arguments, parameter names and outputs as the reference's
models/transformer.py (Transformer :18-80, TransformerEncoder :83-106, TransformerDecoder :109-163,
TransformerEncoderLayer :166-226, TransformerDecoderLayer :229-308, build_transformer :325-337).
Pre-/post-norm variants share one code path here (the norm position is the only difference).
"""
import copy
from typing import Optional

import torch
from torch import Tensor, nn

from .deformable_transformer import _get_activation_fn


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _with_pos(tensor, pos: Optional[Tensor]):
    return tensor if pos is None else tensor + pos


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before

    def with_pos_embed(self, tensor, pos: Optional[Tensor]):
        return _with_pos(tensor, pos)

    def forward(self, src, src_mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None):
        pre = self.normalize_before
        x = self.norm1(src) if pre else src
        q = k = _with_pos(x, pos)
        attn = self.self_attn(q, k, value=x, attn_mask=src_mask,
                              key_padding_mask=src_key_padding_mask)[0]
        src = src + self.dropout1(attn)
        if not pre:
            src = self.norm1(src)
        x = self.norm2(src) if pre else src
        src = src + self.dropout2(self.linear2(self.dropout(self.activation(self.linear1(x)))))
        return src if pre else self.norm2(src)


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before

    def with_pos_embed(self, tensor, pos: Optional[Tensor]):
        return _with_pos(tensor, pos)

    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None,
                memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                query_pos: Optional[Tensor] = None):
        pre = self.normalize_before
        x = self.norm1(tgt) if pre else tgt
        q = k = _with_pos(x, query_pos)
        attn = self.self_attn(q, k, value=x, attn_mask=tgt_mask,
                              key_padding_mask=tgt_key_padding_mask)[0]
        tgt = tgt + self.dropout1(attn)
        if not pre:
            tgt = self.norm1(tgt)
        x = self.norm2(tgt) if pre else tgt
        attn = self.multihead_attn(query=_with_pos(x, query_pos), key=_with_pos(memory, pos),
                                   value=memory, attn_mask=memory_mask,
                                   key_padding_mask=memory_key_padding_mask)[0]
        tgt = tgt + self.dropout2(attn)
        if not pre:
            tgt = self.norm2(tgt)
        x = self.norm3(tgt) if pre else tgt
        tgt = tgt + self.dropout3(self.linear2(self.dropout(self.activation(self.linear1(x)))))
        return tgt if pre else self.norm3(tgt)


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None):
        output = src
        for layer in self.layers:
            output = layer(output, src_mask=mask, src_key_padding_mask=src_key_padding_mask,
                           pos=pos)
        return output if self.norm is None else self.norm(output)


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, encoder_layer, num_layers, norm=None,
                 return_intermediate=False, track_attention=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate
        self.track_attention = track_attention
        if self.track_attention:
            self.layers_track_attention = _get_clones(encoder_layer, num_layers)

    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None,
                memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                query_pos: Optional[Tensor] = None, prev_frame: Optional[dict] = None):
        """-> (normed stack of layer outputs, un-normed stack)."""
        output = tgt
        intermediate = []
        if self.track_attention:  # extra self-attention among the track queries (all but the last 100)
            track_query_pos = query_pos[:-100].clone()
            query_pos[:-100] = 0.0
        for i, layer in enumerate(self.layers):
            if self.track_attention:
                track_output = self.layers_track_attention[i](
                    output[:-100].clone(), src_mask=tgt_mask,
                    src_key_padding_mask=tgt_key_padding_mask, pos=track_query_pos)
                output = torch.cat([track_output, output[-100:]])
            output = layer(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                           tgt_key_padding_mask=tgt_key_padding_mask,
                           memory_key_padding_mask=memory_key_padding_mask, pos=pos,
                           query_pos=query_pos)
            if self.return_intermediate:
                intermediate.append(output)
        if self.return_intermediate:
            output = torch.stack(intermediate)
        if self.norm is not None:
            return self.norm(output), output
        return output, output


class Transformer(nn.Module):
    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False,
                 return_intermediate_dec=False, track_attention=False):
        super().__init__()
        encoder_layer = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout,
                                                activation, normalize_before)
        self.encoder = TransformerEncoder(encoder_layer, num_encoder_layers,
                                          nn.LayerNorm(d_model) if normalize_before else None)
        decoder_layer = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout,
                                                activation, normalize_before)
        self.decoder = TransformerDecoder(decoder_layer, encoder_layer, num_decoder_layers,
                                          nn.LayerNorm(d_model),
                                          return_intermediate=return_intermediate_dec,
                                          track_attention=track_attention)
        self._reset_parameters()
        self.d_model = d_model
        self.nhead = nhead

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, mask, query_embed, pos_embed, tgt=None, prev_frame=None):
        """src [B,C,H,W], mask [B,H,W], query_embed [Q,B,C], pos_embed [B,C,H,W]
        -> (hs [layers,B,Q,C], hs_without_norm, memory [B,C,H,W])."""
        bs, c, h, w = src.shape
        src = src.flatten(2).permute(2, 0, 1)
        pos_embed = pos_embed.flatten(2).permute(2, 0, 1)
        mask = mask.flatten(1)
        if tgt is None:
            tgt = torch.zeros_like(query_embed)
        memory = self.encoder(src, src_key_padding_mask=mask, pos=pos_embed)
        if prev_frame is not None:
            prev_pos = prev_frame['pos'].flatten(2).permute(2, 0, 1)
            prev_mask = prev_frame['mask'].flatten(1)
            prev_frame['memory'] = self.encoder(prev_frame['src'].flatten(2).permute(2, 0, 1),
                                                src_key_padding_mask=prev_mask, pos=prev_pos)
            prev_frame['memory_key_padding_mask'] = prev_mask
            prev_frame['pos'] = prev_pos
        hs, hs_without_norm = self.decoder(tgt, memory, memory_key_padding_mask=mask,
                                           pos=pos_embed, query_pos=query_embed,
                                           prev_frame=prev_frame)
        return (hs.transpose(1, 2), hs_without_norm.transpose(1, 2),
                memory.permute(1, 2, 0).view(bs, c, h, w))


def build_transformer(args):
    return Transformer(d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads,
                       dim_feedforward=args.dim_feedforward, num_encoder_layers=args.enc_layers,
                       num_decoder_layers=args.dec_layers, normalize_before=args.pre_norm,
                       return_intermediate_dec=True, track_attention=args.track_attention)
