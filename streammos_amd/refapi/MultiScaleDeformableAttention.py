"""Drop-in for the pybind module ``MultiScaleDeformableAttention`` (deformattn/src/vision.cpp:13-16).

``ms_deform_attn_forward`` runs the HIP sampler of libsmos_hip.so.  Like the reference's extension
there is no CPU implementation: a non-GPU tensor raises (deformattn/src/ms_deform_attn.h:38).
"""
from .. import ops

__smos_refapi__ = True


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if step > 0 and batch % step != 0:     # ms_deform_attn_cuda.cu:50-52
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (batch, step))
    return ops.msda_fwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]  (deformattn/src/cuda/ms_deform_attn_cuda.cu:83-153)"""
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    return list(ops.msda_bwd(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output.contiguous()))
