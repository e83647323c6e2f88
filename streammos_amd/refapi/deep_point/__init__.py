"""``deep_point``: the point -> grid max-pool operator (mirror of deep_point/__init__.py:15-65).

``VoxelMaxPool(pcds_feat, pcds_ind, output_size, scale_rate)`` keeps the reference's signature,
shape checks and autograd behaviour.  GPU tensors run the fused HIP kernel of libsmos_hip.so, CPU
tensors the host twin in libsmos_cpu.so -- the same placement dispatch as the reference
(deep_point/__init__.py:34-40).  Differences that do not change results: sizes, strides and scales are
passed as host values (the reference uploads four tiny meta tensors per call, :29-32) and the int64
``voxel_max_idx`` scratch is not materialised, because the kernels recompute the cell from the
coordinates.
"""
import torch
from torch.autograd import Function

from ... import ops
from ..point_deep import cpu_kernel

__smos_refapi__ = True


def _check(pcds_feat, pcds_ind, output_size, scale_rate):
    # deep_point/__init__.py:18-23
    assert pcds_feat.dtype == pcds_ind.dtype
    assert pcds_feat.dim() == 4 and pcds_ind.dim() == 4
    assert pcds_feat.size(2) == pcds_ind.size(1)
    assert pcds_ind.size(2) == len(output_size) == len(scale_rate)


class VoxelMaxPoolFunction(Function):
    @staticmethod
    def forward(ctx, pcds_feat, pcds_ind, output_size, scale_rate):
        _check(pcds_feat, pcds_ind, output_size, scale_rate)
        output_size = tuple(int(s) for s in output_size)
        scale_rate = tuple(float(s) for s in scale_rate)
        pcds_ind = pcds_ind.contiguous()
        pcds_feat = pcds_feat.contiguous()
        voxel_out = pcds_feat.new_zeros((pcds_feat.size(0), pcds_feat.size(1)) + output_size)
        if pcds_feat.is_cuda:
            ops.voxel_maxpool_fwd(pcds_feat, pcds_ind, voxel_out, output_size, scale_rate)
        else:
            cpu_kernel.voxel_maxpooling_cpu_forward(pcds_feat, pcds_ind, voxel_out, None, None, None, None,
                                                    torch.tensor(scale_rate, dtype=torch.float32))
        ctx.geometry = (output_size, scale_rate)
        ctx.save_for_backward(pcds_feat, pcds_ind, voxel_out)
        return voxel_out

    @staticmethod
    def backward(ctx, grad_voxel_out):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        pcds_feat, pcds_ind, voxel_out = ctx.saved_tensors
        output_size, scale_rate = ctx.geometry
        grad_voxel_out = grad_voxel_out.contiguous()
        grad_feat = torch.zeros(pcds_feat.shape, dtype=pcds_feat.dtype, device=pcds_feat.device)
        if pcds_feat.is_cuda:
            ops.voxel_maxpool_bwd(pcds_feat, pcds_ind, voxel_out, grad_voxel_out, grad_feat, output_size, scale_rate)
        else:
            cpu_kernel.voxel_maxpooling_cpu_backward(pcds_feat, pcds_ind, voxel_out, None, grad_feat, grad_voxel_out,
                                                     None, None, None, torch.tensor(scale_rate, dtype=torch.float32))
        return grad_feat, None, None, None


def VoxelMaxPool(pcds_feat, pcds_ind, output_size, scale_rate):
    return VoxelMaxPoolFunction.apply(pcds_feat, pcds_ind, output_size, scale_rate)
