"""Deformable attention: the autograd wrapper around the HIP sampler and the attention module built on it.

Interface parity with the reference (names, argument order, parameter names, initial values):
``MSDeformAttnFunction.apply(value, shapes, level_start, locations, weights, im2col_step)``
(deformattn/functions/ms_deform_attn_func.py:21-38) and ``MSDeformAttn(d_model, n_levels, n_heads, n_points)``
with ``forward(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
input_padding_mask)`` (deformattn/modules/ms_deform_attn.py:30-116).
"""
import math

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import MultiScaleDeformableAttention as _ext


class MSDeformAttnFunction(Function):
    """Sampler with gradients for value, sampling locations and attention weights (all through libsmos_hip.so)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.step = im2col_step
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return _ext.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                           attention_weights, im2col_step)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        saved = ctx.saved_tensors
        d_value, d_loc, d_attn = _ext.ms_deform_attn_backward(*saved, grad_output.contiguous(), ctx.step)
        return d_value, None, None, d_loc, d_attn, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Debug-only formulation on torch ops (the reference keeps one for its tests): every level is sampled with
    grid_sample(align_corners=False) at 2*loc-1 and the samples are blended with the attention weights."""
    from torch.nn.functional import grid_sample
    batch, _, heads, dim = value.shape
    n_query, n_points = sampling_locations.shape[1], sampling_locations.shape[4]
    acc = value.new_zeros((batch * heads, dim, n_query))
    start = 0
    for level, hw in enumerate(value_spatial_shapes):
        h, w = int(hw[0]), int(hw[1])
        plane = value[:, start:start + h * w].permute(0, 2, 3, 1).reshape(batch * heads, dim, h, w)
        start += h * w
        where = (2 * sampling_locations[:, :, :, level] - 1).permute(0, 2, 1, 3, 4).reshape(batch * heads, n_query, n_points, 2)
        taken = grid_sample(plane, where, mode="bilinear", padding_mode="zeros", align_corners=False)
        blend = attention_weights[:, :, :, level].permute(0, 2, 1, 3).reshape(batch * heads, 1, n_query, n_points)
        acc = acc + (taken * blend).sum(-1)
    return acc.view(batch, heads * dim, n_query).transpose(1, 2).contiguous()


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.im2col_step = 256
        samples = n_heads * n_levels * n_points
        # registration order = checkpoint key order of the reference
        self.sampling_offsets = nn.Linear(d_model, 2 * samples)
        self.attention_weights = nn.Linear(d_model, samples)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    @torch.no_grad()
    def _reset_parameters(self):
        """Offsets start as a fixed star pattern (head h looks along direction 2*pi*h/n_heads, normalised to the unit
        square's boundary, point p at radius p+1) with zero weights; attention logits start uniform; the two projections
        use Xavier-uniform weights and zero bias."""
        angles = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        direction = torch.stack((torch.cos(angles), torch.sin(angles)), dim=-1)
        direction = direction / direction.abs().amax(dim=-1, keepdim=True)
        radius = torch.arange(1, self.n_points + 1, dtype=torch.float32)
        star = direction[:, None, None, :] * radius[None, None, :, None]              # (heads, 1, points, 2)
        star = star.expand(self.n_heads, self.n_levels, self.n_points, 2)
        self.sampling_offsets.weight.zero_()
        self.sampling_offsets.bias.copy_(star.reshape(-1))
        for lin in (self.attention_weights,):
            lin.weight.zero_()
            lin.bias.zero_()
        for lin in (self.value_proj, self.output_proj):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.zero_()

    def _locations(self, reference_points, offsets, shapes):
        """Sampling positions in [0,1]^2: reference point + offset measured in cells of each level (2-d references) or in
        fractions of the reference box (4-d references)."""
        if reference_points.shape[-1] == 2:
            cells_wh = shapes.flip(-1)                                                 # (levels, 2) as (W, H)
            return reference_points[:, :, None, :, None, :] + offsets / cells_wh[None, None, None, :, None, :]
        if reference_points.shape[-1] == 4:
            centre, size = reference_points[..., :2], reference_points[..., 2:]
            return centre[:, :, None, :, None, :] + offsets / self.n_points * size[:, :, None, :, None, :] * 0.5
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        batch, n_query = query.shape[0], query.shape[1]
        n_keys = input_flatten.shape[1]
        assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == n_keys
        keys = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            pad = input_padding_mask[..., None]
            keys = keys.masked_fill(pad, 0.0)
            query = query.masked_fill(pad, 0.0)
        h, l, p = self.n_heads, self.n_levels, self.n_points
        keys = keys.view(batch, n_keys, h, self.d_model // h)
        offsets = self.sampling_offsets(query).view(batch, n_query, h, l, p, 2)
        logits = self.attention_weights(query).view(batch, n_query, h, l * p)
        weights = torch.softmax(logits, dim=-1).view(batch, n_query, h, l, p)
        where = self._locations(reference_points, offsets, input_spatial_shapes)
        mixed = MSDeformAttnFunction.apply(keys, input_spatial_shapes, input_level_start_index, where.contiguous(),
                                           weights.contiguous(), self.im2col_step)
        return self.output_proj(mixed)
