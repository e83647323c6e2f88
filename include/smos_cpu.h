/*
 * smos_cpu.h -- C ABI of libsmos_cpu.so: the host twin of the point -> grid max-pool.
 *
 * The reference ships a CPU extension next to the CUDA one (pybind module point_deep.cpu_kernel,
 * deep_point/src/point_deep.cpp:183-186) because its DataLoader workers call VoxelMaxPool on CPU
 * tensors to rasterise labels into BEV (datasets/data_StreamMOS.py:284-290,536-542).  This library is
 * that twin: plain C++, no HIP, fork-safe, never touches the GPU.  It is dispatched on tensor
 * placement exactly like the reference (deep_point/__init__.py:34-40) -- it is NOT a fallback for
 * GPU tensors, which always go to libsmos_hip.so.
 */
#ifndef SMOS_CPU_H_
#define SMOS_CPU_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same semantics and argument meaning as smos_voxel_maxpool_fwd/bwd in smos.h, host pointers,
 * dtype: 0 = float32, 2 = float64.  Returns 0 on success, 1 on bad arguments, 2 on unsupported dtype. */
int smos_cpu_voxel_maxpool_fwd(const void* feat, const int64_t* feat_stride, const void* ind, void* out,
                               const int64_t* out_stride, int64_t* voxel_max_idx, int64_t BS, int64_t C, int64_t N,
                               int32_t D, const int64_t* out_size, const float* scale, int32_t dtype);
int smos_cpu_voxel_maxpool_bwd(const void* feat, const int64_t* feat_stride, const void* ind, const void* out,
                               const void* grad_out, const int64_t* out_stride, void* grad_feat, int64_t BS,
                               int64_t C, int64_t N, int32_t D, const int64_t* out_size, const float* scale,
                               int32_t dtype);
#ifdef __cplusplus
}
#endif
#endif
