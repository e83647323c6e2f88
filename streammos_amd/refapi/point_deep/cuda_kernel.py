"""Drop-in for the pybind module ``point_deep.cuda_kernel`` (deep_point/src/point_deep_cuda.cpp:59-62).

Same two functions, same argument lists, same in-place contract: the caller owns every buffer
(``voxel_out`` zero-filled, ``voxel_max_idx`` filled with -1, ``grad_pcds_feat`` zero-filled).  The four
small meta tensors live on the device in the reference (deep_point/__init__.py:29-32); sizes and strides
are taken from ``voxel_out`` itself here, only ``scale_rate`` has to be read back (one tiny D2H copy --
callers that care use ``streammos_amd.ops`` / ``deep_point.VoxelMaxPool``, which pass host values).
"""
from ... import ops

__smos_refapi__ = True


def _check_inputs(*tensors):
    # CHECK_INPUT of point_deep_cuda.cpp:11-13
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("point_deep.cuda_kernel: tensor must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError("point_deep.cuda_kernel: tensor must be contiguous")


def voxel_maxpooling_forward(pcds_feat, pcds_ind, voxel_out, voxel_max_idx, voxel_out_size, voxel_out_stride,
                             output_size, scale_rate):
    _check_inputs(pcds_feat, pcds_ind, voxel_out, voxel_max_idx, voxel_out_size, voxel_out_stride, output_size, scale_rate)
    ops.voxel_maxpool_fwd(pcds_feat, pcds_ind, voxel_out, tuple(voxel_out.shape[2:]),
                          [float(s) for s in scale_rate.detach().cpu().tolist()], voxel_max_idx=voxel_max_idx)


def voxel_maxpooling_backward(pcds_feat, pcds_ind, voxel_out, voxel_max_idx, grad_pcds_feat, grad_voxel_out,
                              voxel_out_size, voxel_out_stride, output_size, scale_rate):
    _check_inputs(pcds_feat, pcds_ind, voxel_out, voxel_max_idx, grad_pcds_feat, grad_voxel_out, voxel_out_size,
                  voxel_out_stride, output_size, scale_rate)
    ops.voxel_maxpool_bwd(pcds_feat, pcds_ind, voxel_out, grad_voxel_out, grad_pcds_feat, tuple(voxel_out.shape[2:]),
                          [float(s) for s in scale_rate.detach().cpu().tolist()])
