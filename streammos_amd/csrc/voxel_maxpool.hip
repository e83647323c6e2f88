// Point -> grid max-pool scatter for gfx950 (replaces deep_point's three CUDA kernels,
// reference: deep_point/src/point_deep_cuda_kernel.cu:24-99,109-132).
//
// Design (see DESIGN.md "VoxelMaxPool"):
//  * ONE fused launch: the cell of a point is recomputed from its coordinates in registers, so the
//    reference's int64 [BS,N] index scratch is only written when the caller asks for it (backward).
//  * The caller hands over a zero-filled output (that is the reference's contract,
//    deep_point/__init__.py:26).  For values > 0 the IEEE bit pattern is monotone as a signed integer
//    and 0.0f is the identity, so the max is a single native `global_atomic_smax` -- no CAS loop
//    (reference: atomics.cuh:76-88) and no separate "init" pass.  Zeros are skipped entirely (about
//    half of a post-ReLU feature map).
//  * Negative values cannot use that trick (an all-negative cell must end with its largest negative,
//    not 0).  The fast kernel raises a device flag when it meets one; two follow-up launches that
//    return immediately while the flag is clear then redo the exact reference algorithm (racing
//    member store, then signed-max / unsigned-min on the bit pattern).  The model never takes it: all
//    five call sites scatter post-ReLU features (models/StreamMOS.py:102, multi_view_encoder.py:396-419).
//  * Two lane mappings, chosen from the strides on the host:
//      "points": one lane per point, loop over channels  -> coalesced reads of channel-major [C,N]
//                features (the reference layout), scattered 4-byte atomics;
//      "rows"  : one lane per channel, a group of lanes per point -> point-major features and a
//                channels-last grid: every wave instruction is one or two contiguous 128/256-byte rows,
//                the shape HBM-side atomics run at full rate on MI355X.
#include <hip/hip_fp16.h>

#include "smos_common.h"

namespace smos {

struct VmpGeom {
  int32_t D;
  int64_t size[4];
  int64_t sstride[4];  // spatial strides of out (elements)
  float scale[4];
  int64_t out_b, out_c;      // batch / channel stride of out
  int64_t fs_b, fs_c, fs_n;  // feature strides
};

// Flat spatial offset of a point, or -1 when it is dropped.
// reference: point_deep_cuda_kernel.cu:39-47 -- int64(float(coord) * scale), kept iff 0 <= cell < size.
// trunc(p) >= 0  <=>  p > -1 and trunc(p) < size  <=>  p < size (size is an integer).
__device__ __forceinline__ int64_t cell_offset(const float* __restrict__ ind_row, const VmpGeom& g) {
  int64_t off = 0;
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    if (d < g.D) {
      float p = __fmul_rn(ind_row[d], g.scale[d]);
      bool in = (p > -1.0f) && (p < (float)g.size[d]);
      ok = ok && in;
      off += (int64_t)(in ? (int)p : 0) * g.sstride[d];
    }
  }
  return ok ? off : (int64_t)-1;
}

// ---------------------------------------------------------------------------------------------
// mapping "points": lane = point, loop over a chunk of channels
// ---------------------------------------------------------------------------------------------
template <bool kWriteIdx>
__global__ __launch_bounds__(kBlock) void vmp_fwd_points(const float* __restrict__ feat,
                                                         const float* __restrict__ ind, float* __restrict__ out,
                                                         int64_t* __restrict__ vidx, int* flag, VmpGeom g,
                                                         int64_t BS, int64_t C, int64_t N, int c_chunk) {
  const int64_t total = BS * N;
  const int c0 = blockIdx.y * c_chunk;
  const int c1 = (int)min((int64_t)(c0 + c_chunk), C);
  int saw_neg = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / N;
    const int64_t n = i - b * N;
    int64_t off = cell_offset(ind + i * g.D, g);
    if (off < 0) continue;
    off += b * g.out_b;
    if (kWriteIdx && blockIdx.y == 0) vidx[i] = off;
    const float* __restrict__ f = feat + b * g.fs_b + n * g.fs_n;
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = f[(int64_t)(c + k) * g.fs_c];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (v[k] > 0.0f)
          atomicMax(reinterpret_cast<int*>(out + off + (int64_t)(c + k) * g.out_c), __float_as_int(v[k]));
        else if (v[k] < 0.0f)
          saw_neg = 1;
      }
    }
    for (; c < c1; ++c) {
      float v = f[(int64_t)c * g.fs_c];
      if (v > 0.0f)
        atomicMax(reinterpret_cast<int*>(out + off + (int64_t)c * g.out_c), __float_as_int(v));
      else if (v < 0.0f)
        saw_neg = 1;
    }
  }
  if (saw_neg) *flag = 1;
}

// ---------------------------------------------------------------------------------------------
// mapping "rows": a group of kG lanes (kG = 8..64) owns one point, lane = channel (+ kG per step)
// requires fs_c == 1 and out_c == 1
// ---------------------------------------------------------------------------------------------
template <int kG, bool kWriteIdx>
__global__ __launch_bounds__(kBlock) void vmp_fwd_rows(const float* __restrict__ feat, const float* __restrict__ ind,
                                                       float* __restrict__ out, int64_t* __restrict__ vidx,
                                                       int* flag, VmpGeom g, int64_t BS, int64_t C, int64_t N) {
  constexpr int kGroups = kBlock / kG;
  const int lane = threadIdx.x % kG;
  const int grp = threadIdx.x / kG;
  const int64_t total = BS * N;
  int saw_neg = 0;
  for (int64_t i = (int64_t)blockIdx.x * kGroups + grp; i < total; i += (int64_t)gridDim.x * kGroups) {
    const int64_t b = i / N;
    const int64_t n = i - b * N;
    int64_t off = cell_offset(ind + i * g.D, g);
    if (off < 0) continue;
    off += b * g.out_b;
    if (kWriteIdx && lane == 0) vidx[i] = off;
    const float* __restrict__ f = feat + b * g.fs_b + n * g.fs_n;
    for (int c = lane; c < C; c += kG) {
      float v = f[c];
      if (v > 0.0f)
        atomicMax(reinterpret_cast<int*>(out + off + c), __float_as_int(v));
      else if (v < 0.0f)
        saw_neg = 1;
    }
  }
  if (saw_neg) *flag = 1;
}

// ---------------------------------------------------------------------------------------------
// exact path for negative values (reference algorithm), gated by the flag
// ---------------------------------------------------------------------------------------------
template <int kPass>  // 0: store one member per occupied cell, 1: max over all members
__global__ __launch_bounds__(kBlock) void vmp_fwd_signed(const float* __restrict__ feat, const float* __restrict__ ind,
                                                         float* __restrict__ out, const int* flag, VmpGeom g,
                                                         int64_t BS, int64_t C, int64_t N) {
  if (*reinterpret_cast<const volatile int*>(flag) == 0) return;
  const int64_t total = BS * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / N;
    const int64_t n = i - b * N;
    int64_t off = cell_offset(ind + i * g.D, g);
    if (off < 0) continue;
    off += b * g.out_b;
    const float* __restrict__ f = feat + b * g.fs_b + n * g.fs_n;
    for (int64_t c = 0; c < C; ++c) {
      const float v = f[c * g.fs_c];
      float* p = out + off + c * g.out_c;
      if (kPass == 0) {
        *p = v;  // racing plain store, like VoxelMaxPoolUpdateOutputInit (cuda_kernel.cu:72-74)
      } else if (v == v) {
        // the cell now holds a real member: signed max for v >= 0 beats any negative pattern, unsigned
        // min among negative patterns picks the smallest magnitude and never displaces a non-negative
        if (v >= 0.0f)
          atomicMax(reinterpret_cast<int*>(p), __float_as_int(v));
        else
          atomicMin(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void vmp_bwd_points(const float* __restrict__ feat, const float* __restrict__ ind,
                                                         const float* __restrict__ out,
                                                         const float* __restrict__ grad_out,
                                                         float* __restrict__ grad_feat, VmpGeom g, int64_t BS,
                                                         int64_t C, int64_t N) {
  const int64_t total = BS * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / N;
    const int64_t n = i - b * N;
    int64_t off = cell_offset(ind + i * g.D, g);
    if (off < 0) continue;
    off += b * g.out_b;
    const int64_t fo = b * g.fs_b + n * g.fs_n;
    for (int64_t c = 0; c < C; ++c) {
      const float v = feat[fo + c * g.fs_c];
      const int64_t o = off + c * g.out_c;
      if (out[o] == v) grad_feat[fo + c * g.fs_c] = grad_out[o];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// float16 / float64 (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, point_deep_cuda_kernel.cu:147,168; the
// model itself only ever calls float32, models/StreamMOS.py:18).  Correctness path, not tuned: the reference's own
// algorithm -- a racing member store per occupied cell (kPass 0, VoxelMaxPoolUpdateOutputInit :60-79), then an atomic max over
// all members (kPass 1, :82-99).  float64: the signed-max / unsigned-min pair on the 64-bit pattern (no CAS loop); float16:
// compare-and-swap on the containing 32-bit word, as atomics.cuh:225-242 does.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct VmpNum;
template <>
struct VmpNum<double> {
  static __device__ __forceinline__ float to_float(double v) { return (float)v; }
  static __device__ __forceinline__ bool is_nan(double v) { return v != v; }
  static __device__ __forceinline__ void atom_max(double* p, double v) {
    if (v >= 0.0)
      atomicMax(reinterpret_cast<long long*>(p), __double_as_longlong(v));
    else
      atomicMin(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
  }
  static __device__ __forceinline__ bool equal(double a, double b) { return a == b; }
};
template <>
struct VmpNum<__half> {
  static __device__ __forceinline__ float to_float(__half v) { return __half2float(v); }
  static __device__ __forceinline__ bool is_nan(__half v) { return __hisnan(v); }
  static __device__ __forceinline__ void atom_max(__half* p, __half v) {
    unsigned int* word = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(p) - (reinterpret_cast<size_t>(p) & 2));
    const bool hi = (reinterpret_cast<size_t>(p) & 2) != 0;
    const unsigned short vb = __half_as_ushort(v);
    unsigned int old = *word, assumed;
    do {
      assumed = old;
      const unsigned short cur = hi ? (unsigned short)(old >> 16) : (unsigned short)(old & 0xffffu);
      if (!(__half2float(v) > __half2float(__ushort_as_half(cur)))) return;        // the cell already holds a value >= v
      const unsigned int want = hi ? ((old & 0xffffu) | ((unsigned int)vb << 16)) : ((old & 0xffff0000u) | vb);
      old = atomicCAS(word, assumed, want);
    } while (assumed != old);
  }
  static __device__ __forceinline__ bool equal(__half a, __half b) { return __half2float(a) == __half2float(b); }
};

template <typename T>
__device__ __forceinline__ int64_t cell_offset_t(const T* __restrict__ ind_row, const VmpGeom& g) {
  float row[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < 4; ++d)
    if (d < g.D) row[d] = VmpNum<T>::to_float(ind_row[d]);      // static_cast<float>(pcds_ind), cuda_kernel.cu:40
  return cell_offset(row, g);
}

template <typename T, int kPass>
__global__ __launch_bounds__(kBlock) void vmp_fwd_generic(const T* __restrict__ feat, const T* __restrict__ ind, T* __restrict__ out,
                                                          int64_t* __restrict__ vidx, VmpGeom g, int64_t BS, int64_t C, int64_t N) {
  const int64_t total = BS * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / N;
    const int64_t n = i - b * N;
    int64_t off = cell_offset_t<T>(ind + i * g.D, g);
    if (off < 0) continue;
    off += b * g.out_b;
    if (kPass == 0 && vidx) vidx[i] = off;
    const T* __restrict__ f = feat + b * g.fs_b + n * g.fs_n;
    for (int64_t c = 0; c < C; ++c) {
      const T v = f[c * g.fs_c];
      T* p = out + off + c * g.out_c;
      if (kPass == 0)
        *p = v;
      else if (!VmpNum<T>::is_nan(v))
        VmpNum<T>::atom_max(p, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void vmp_bwd_generic(const T* __restrict__ feat, const T* __restrict__ ind, const T* __restrict__ out,
                                                          const T* __restrict__ grad_out, T* __restrict__ grad_feat, VmpGeom g,
                                                          int64_t BS, int64_t C, int64_t N) {
  const int64_t total = BS * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / N;
    const int64_t n = i - b * N;
    int64_t off = cell_offset_t<T>(ind + i * g.D, g);
    if (off < 0) continue;
    off += b * g.out_b;
    const int64_t fo = b * g.fs_b + n * g.fs_n;
    for (int64_t c = 0; c < C; ++c) {
      const int64_t o = off + c * g.out_c;
      if (VmpNum<T>::equal(out[o], feat[fo + c * g.fs_c])) grad_feat[fo + c * g.fs_c] = grad_out[o];
    }
  }
}

template <typename T>
static int vmp_fwd_generic_launch(const void* feat, const void* ind, void* out, int64_t* vidx, const VmpGeom& g, int64_t BS,
                                  int64_t C, int64_t N, hipStream_t s) {
  dim3 grid(grid_for(BS * N));
  hipLaunchKernelGGL((vmp_fwd_generic<T, 0>), grid, dim3(kBlock), 0, s, (const T*)feat, (const T*)ind, (T*)out, vidx, g, BS, C, N);
  hipLaunchKernelGGL((vmp_fwd_generic<T, 1>), grid, dim3(kBlock), 0, s, (const T*)feat, (const T*)ind, (T*)out, vidx, g, BS, C, N);
  return check_launch("voxel_maxpool_fwd");
}

static int fill_geom(VmpGeom& g, const int64_t* feat_stride, const int64_t* out_stride, int32_t D,
                     const int64_t* out_size, const float* scale) {
  SMOS_REQUIRE(D >= 1 && D <= 4, "voxel_maxpool: D=%d outside 1..4", (int)D);
  SMOS_REQUIRE(feat_stride && out_stride && out_size && scale, "voxel_maxpool: null shape/stride vector");
  g.D = D;
  for (int d = 0; d < 4; ++d) {
    g.size[d] = d < D ? out_size[d] : 1;
    g.sstride[d] = d < D ? out_stride[2 + d] : 0;
    g.scale[d] = d < D ? scale[d] : 0.f;
    if (d < D) SMOS_REQUIRE(out_size[d] > 0 && out_size[d] < (1 << 24), "voxel_maxpool: out_size[%d]=%lld", d, (long long)out_size[d]);
  }
  g.out_b = out_stride[0];
  g.out_c = out_stride[1];
  g.fs_b = feat_stride[0];
  g.fs_c = feat_stride[1];
  g.fs_n = feat_stride[2];
  return SMOS_OK;
}

}  // namespace smos

using namespace smos;

extern "C" int smos_voxel_maxpool_fwd(const void* feat, const int64_t* feat_stride, const void* ind, void* out,
                                      const int64_t* out_stride, int64_t* voxel_max_idx, int64_t BS, int64_t C,
                                      int64_t N, int32_t D, const int64_t* out_size, const float* scale,
                                      int32_t dtype, int32_t* flag_ws, smos_stream_t stream) {
  if (dtype != SMOS_F32 && dtype != SMOS_F16 && dtype != SMOS_F64) {
    set_error("voxel_maxpool_fwd: unknown dtype code %d", (int)dtype);
    return SMOS_ERR_UNSUPPORTED;
  }
  SMOS_REQUIRE(BS >= 0 && C >= 0 && N >= 0, "voxel_maxpool_fwd: negative size");
  if (BS == 0 || C == 0 || N == 0) return SMOS_OK;
  SMOS_REQUIRE(feat && ind && out && flag_ws, "voxel_maxpool_fwd: null device pointer");
  VmpGeom g;
  if (int rc = fill_geom(g, feat_stride, out_stride, D, out_size, scale)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SMOS_F64) return vmp_fwd_generic_launch<double>(feat, ind, out, voxel_max_idx, g, BS, C, N, s);
  if (dtype == SMOS_F16) return vmp_fwd_generic_launch<__half>(feat, ind, out, voxel_max_idx, g, BS, C, N, s);
  const float* f = (const float*)feat;
  const float* in = (const float*)ind;
  float* o = (float*)out;
  if (hipMemsetAsync(flag_ws, 0, sizeof(int32_t), s) != hipSuccess) return check_launch("voxel_maxpool_fwd memset");

  const int64_t pts = BS * N;
  if (g.fs_c == 1 && g.out_c == 1 && C >= 8) {
    int G = 8;
    while (G < 64 && G < C) G <<= 1;
    const int64_t groups_per_block = kBlock / G;
    dim3 grid(grid_for(pts * G, kBlock, 256 * 16));
    (void)groups_per_block;
#define SMOS_ROWS(GG)                                                                                         \
  if (voxel_max_idx)                                                                                          \
    hipLaunchKernelGGL((vmp_fwd_rows<GG, true>), grid, dim3(kBlock), 0, s, f, in, o, voxel_max_idx, flag_ws, g, BS, C, N); \
  else                                                                                                        \
    hipLaunchKernelGGL((vmp_fwd_rows<GG, false>), grid, dim3(kBlock), 0, s, f, in, o, voxel_max_idx, flag_ws, g, BS, C, N)
    switch (G) {
      case 8: SMOS_ROWS(8); break;
      case 16: SMOS_ROWS(16); break;
      case 32: SMOS_ROWS(32); break;
      default: SMOS_ROWS(64); break;
    }
#undef SMOS_ROWS
  } else {
    // enough lanes to fill 256 CUs: split the channel loop over blockIdx.y when there are few points
    int chunks = 1;
    if (pts < (1 << 20) && C >= 16) {
      chunks = (int)(((int64_t)(1 << 20) + pts - 1) / pts);
      if (chunks > C / 8) chunks = (int)(C / 8);
      if (chunks < 1) chunks = 1;
    }
    int c_chunk = (int)((C + chunks - 1) / chunks);
    c_chunk = (c_chunk + 7) / 8 * 8;
    chunks = (int)((C + c_chunk - 1) / c_chunk);
    dim3 grid(grid_for(pts, kBlock, 256 * 16), chunks);
    if (voxel_max_idx)
      hipLaunchKernelGGL((vmp_fwd_points<true>), grid, dim3(kBlock), 0, s, f, in, o, voxel_max_idx, flag_ws, g, BS, C, N, c_chunk);
    else
      hipLaunchKernelGGL((vmp_fwd_points<false>), grid, dim3(kBlock), 0, s, f, in, o, voxel_max_idx, flag_ws, g, BS, C, N, c_chunk);
  }
  dim3 sgrid(grid_for(pts));
  hipLaunchKernelGGL((vmp_fwd_signed<0>), sgrid, dim3(kBlock), 0, s, f, in, o, (const int*)flag_ws, g, BS, C, N);
  hipLaunchKernelGGL((vmp_fwd_signed<1>), sgrid, dim3(kBlock), 0, s, f, in, o, (const int*)flag_ws, g, BS, C, N);
  return check_launch("voxel_maxpool_fwd");
}

extern "C" int smos_voxel_maxpool_bwd(const void* feat, const int64_t* feat_stride, const void* ind, const void* out,
                                      const void* grad_out, const int64_t* out_stride, void* grad_feat, int64_t BS,
                                      int64_t C, int64_t N, int32_t D, const int64_t* out_size, const float* scale,
                                      int32_t dtype, smos_stream_t stream) {
  if (dtype != SMOS_F32 && dtype != SMOS_F16 && dtype != SMOS_F64) {
    set_error("voxel_maxpool_bwd: unknown dtype code %d", (int)dtype);
    return SMOS_ERR_UNSUPPORTED;
  }
  SMOS_REQUIRE(BS >= 0 && C >= 0 && N >= 0, "voxel_maxpool_bwd: negative size");
  if (BS == 0 || C == 0 || N == 0) return SMOS_OK;
  SMOS_REQUIRE(feat && ind && out && grad_out && grad_feat, "voxel_maxpool_bwd: null device pointer");
  VmpGeom g;
  if (int rc = fill_geom(g, feat_stride, out_stride, D, out_size, scale)) return rc;
  if (dtype == SMOS_F64) {
    hipLaunchKernelGGL((vmp_bwd_generic<double>), dim3(grid_for(BS * N)), dim3(kBlock), 0, (hipStream_t)stream, (const double*)feat,
                       (const double*)ind, (const double*)out, (const double*)grad_out, (double*)grad_feat, g, BS, C, N);
    return check_launch("voxel_maxpool_bwd");
  }
  if (dtype == SMOS_F16) {
    hipLaunchKernelGGL((vmp_bwd_generic<__half>), dim3(grid_for(BS * N)), dim3(kBlock), 0, (hipStream_t)stream, (const __half*)feat,
                       (const __half*)ind, (const __half*)out, (const __half*)grad_out, (__half*)grad_feat, g, BS, C, N);
    return check_launch("voxel_maxpool_bwd");
  }
  hipLaunchKernelGGL(vmp_bwd_points, dim3(grid_for(BS * N)), dim3(kBlock), 0, (hipStream_t)stream, (const float*)feat,
                     (const float*)ind, (const float*)out, (const float*)grad_out, (float*)grad_feat, g, BS, C, N);
  return check_launch("voxel_maxpool_bwd");
}
