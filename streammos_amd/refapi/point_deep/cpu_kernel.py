"""Drop-in for the pybind module ``point_deep.cpu_kernel`` (deep_point/src/point_deep.cpp:183-186).

Called on CPU tensors only -- e.g. from forked DataLoader workers that rasterise labels into BEV
(datasets/data_StreamMOS.py:284-290,536-542) -- so it binds libsmos_cpu.so (plain C++, no HIP; see
include/smos_cpu.h) and never touches the GPU.
"""
import ctypes
import os

import torch

from ... import _lib

__smos_refapi__ = True

_CPU_LIB = os.path.join(os.path.dirname(_lib.LIB_PATH), "libsmos_cpu.so")
_cpu = None
_CODE = {torch.float32: 0, torch.float64: 2}


def _load():
    global _cpu
    if _cpu is None:
        if not os.path.isfile(_CPU_LIB):
            raise RuntimeError("point_deep.cpu_kernel: %s is missing; run `python -m streammos_amd.build`" % _CPU_LIB)
        lib = ctypes.CDLL(_CPU_LIB)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        lib.smos_cpu_voxel_maxpool_fwd.argtypes = [vp, _lib.c_i64p, vp, vp, _lib.c_i64p, vp, i64, i64, i64, i32,
                                                   _lib.c_i64p, _lib.c_f32p, i32]
        lib.smos_cpu_voxel_maxpool_bwd.argtypes = [vp, _lib.c_i64p, vp, vp, vp, _lib.c_i64p, vp, i64, i64, i64, i32,
                                                   _lib.c_i64p, _lib.c_f32p, i32]
        _cpu = lib
    return _cpu


def _common(pcds_feat, pcds_ind, voxel_out, scale_rate):
    for t in (pcds_feat, pcds_ind, voxel_out):
        if t.is_cuda:
            raise RuntimeError("point_deep.cpu_kernel: got a CUDA tensor")
    code = _CODE.get(pcds_feat.dtype)
    if code is None or pcds_ind.dtype != pcds_feat.dtype or voxel_out.dtype != pcds_feat.dtype:
        raise RuntimeError("point_deep.cpu_kernel: float32/float64 tensors of one dtype expected, got %s/%s/%s"
                           % (pcds_feat.dtype, pcds_ind.dtype, voxel_out.dtype))
    if not pcds_ind.is_contiguous():
        raise RuntimeError("point_deep.cpu_kernel: pcds_ind must be contiguous")
    d = pcds_ind.shape[2]
    return (code, d, _lib.i64_array(pcds_feat.stride()[:3]), _lib.i64_array(voxel_out.stride()),
            _lib.i64_array(voxel_out.shape[2:]), _lib.f32_array([float(s) for s in scale_rate.tolist()]))


def voxel_maxpooling_cpu_forward(pcds_feat, pcds_ind, voxel_out, voxel_max_idx, voxel_out_size, voxel_out_stride,
                                 output_size, scale_rate):
    code, d, fs, os_, size, scale = _common(pcds_feat, pcds_ind, voxel_out, scale_rate)
    rc = _load().smos_cpu_voxel_maxpool_fwd(pcds_feat.data_ptr(), fs, pcds_ind.data_ptr(), voxel_out.data_ptr(), os_,
                                            voxel_max_idx.data_ptr() if voxel_max_idx is not None else None,
                                            pcds_feat.shape[0], pcds_feat.shape[1], pcds_feat.shape[2], d, size, scale, code)
    if rc != 0:
        raise RuntimeError("point_deep.cpu_kernel.voxel_maxpooling_cpu_forward failed with status %d" % rc)


def voxel_maxpooling_cpu_backward(pcds_feat, pcds_ind, voxel_out, voxel_max_idx, grad_pcds_feat, grad_voxel_out,
                                  voxel_out_size, voxel_out_stride, output_size, scale_rate):
    code, d, fs, os_, size, scale = _common(pcds_feat, pcds_ind, voxel_out, scale_rate)
    # strides of size-1 dims are arbitrary, so layouts are compared through contiguity, not stride tuples
    if not (grad_voxel_out.is_contiguous() and voxel_out.is_contiguous() and grad_pcds_feat.is_contiguous()
            and pcds_feat.is_contiguous()):
        raise RuntimeError("point_deep.cpu_kernel: backward expects contiguous tensors")
    rc = _load().smos_cpu_voxel_maxpool_bwd(pcds_feat.data_ptr(), fs, pcds_ind.data_ptr(), voxel_out.data_ptr(),
                                            grad_voxel_out.data_ptr(), os_, grad_pcds_feat.data_ptr(), pcds_feat.shape[0],
                                            pcds_feat.shape[1], pcds_feat.shape[2], d, size, scale, code)
    if rc != 0:
        raise RuntimeError("point_deep.cpu_kernel.voxel_maxpooling_cpu_backward failed with status %d" % rc)
