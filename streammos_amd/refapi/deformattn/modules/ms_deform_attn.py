"""``MSDeformAttn`` module (mirror of deformattn/modules/ms_deform_attn.py:30-116): same constructor,
parameter names (sampling_offsets, attention_weights, value_proj, output_proj), initialisation and
forward contract; the sampler itself is the HIP kernel behind ``MSDeformAttnFunction``."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from ..functions import MSDeformAttnFunction


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.im2col_step = 256
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        # ms_deform_attn.py:62-76: zero offset weights, offset bias = unit directions on a square ring scaled
        # by the point index, zero attention logits, xavier projections
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            ang = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
            ring = torch.stack((ang.cos(), ang.sin()), -1)
            ring = ring / ring.abs().max(-1, keepdim=True)[0]
            ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
            ring = ring * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
            self.sampling_offsets.bias.copy_(ring.reshape(-1))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        n, len_q, _ = query.shape
        len_in = input_flatten.shape[1]
        assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == len_in
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
            query = query.masked_fill(input_padding_mask[..., None], 0.0)
        heads, levels, points = self.n_heads, self.n_levels, self.n_points
        value = value.view(n, len_in, heads, self.d_model // heads)
        offsets = self.sampling_offsets(query).view(n, len_q, heads, levels, points, 2)
        weights = F.softmax(self.attention_weights(query).view(n, len_q, heads, levels * points), -1)
        weights = weights.view(n, len_q, heads, levels, points)
        if reference_points.shape[-1] == 2:
            wh = torch.stack((input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]), -1)
            locations = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead."
                             .format(reference_points.shape[-1]))
        sampled = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                             locations.contiguous(), weights.contiguous(), self.im2col_step)
        return self.output_proj(sampled)
