"""``MSDeformAttnFunction`` (mirror of deformattn/functions/ms_deform_attn_func.py:21-38)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ... import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        out = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                          attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        g_value, g_loc, g_attn = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output.contiguous(),
                                                               ctx.im2col_step)
        return g_value, None, None, g_loc, g_attn, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Debug-only torch formulation (the reference keeps one at ms_deform_attn_func.py:41-61): per level,
    grid_sample(align_corners=False) of the value map at 2*loc-1, weighted sum over levels and points."""
    n, _, m, d = value.shape
    lq, p = sampling_locations.shape[1], sampling_locations.shape[4]
    sizes = [int(h) * int(w) for h, w in value_spatial_shapes]
    total = value.new_zeros((n * m, d, lq))
    for lvl, (chunk, hw) in enumerate(zip(value.split(sizes, dim=1), value_spatial_shapes)):
        h, w = int(hw[0]), int(hw[1])
        img = chunk.permute(0, 2, 3, 1).reshape(n * m, d, h, w)
        grid = (2 * sampling_locations[:, :, :, lvl] - 1).permute(0, 2, 1, 3, 4).reshape(n * m, lq, p, 2)
        sampled = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        wts = attention_weights[:, :, :, lvl].permute(0, 2, 1, 3).reshape(n * m, 1, lq, p)
        total = total + (sampled * wts).sum(-1)
    return total.view(n, m * d, lq).transpose(1, 2).contiguous()
