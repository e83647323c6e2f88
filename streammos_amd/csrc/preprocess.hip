// Device-side validation preprocessing for gfx950.
//
// Restates what DataloadVal does per sample on the host with numpy (datasets/data_StreamMOS.py:397-599,
// datasets/utils.py:98-192) so that only the raw scans (3 x ~1.9 MB) and three 4x4 pose differences cross
// PCIe instead of the 92 MB of derived tensors:
//   transform_mask : pose alignment in float64 -> float32 (utils.py:116-126) + half-open range mask (:107-113)
//   emit           : order-preserving compaction through an inclusive prefix sum of the mask, the TTA sign flips
//                    (data_StreamMOS.py:495-513), BEV / range-view quantisation (utils.py:151-192), the 7-channel
//                    point feature (data_StreamMOS.py:25-50) and the -1000 / -4000 padding tail (:567-574)
//   unpad_labels   : per-point labels of the padded sample scattered back to the raw scan (val_StreamMOS.py:112-118)
// float32 arithmetic is written with the round-to-nearest intrinsics in the operation order of the numpy
// expressions; sqrt is correctly rounded on both sides, asinf / atan2f may differ from numpy's SIMD
// routines in the last ulp (documented tolerance in tests/test_gpu_preprocess.py).
#include "smos_common.h"

namespace smos {

struct PrepGeom {
  float lo[3], hi[3], cell[3];
  float phi_hi, dphi, th_hi, dtheta;
};

struct PrepPose {
  double m[12];
  int identity;
};

__global__ __launch_bounds__(kBlock) void prep_transform_mask(const float* __restrict__ scan, int64_t n, PrepPose pose,
                                                              PrepGeom g, float* __restrict__ moved,
                                                              int32_t* __restrict__ mask) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = reinterpret_cast<const float4*>(scan)[i];
    float x = p.x, y = p.y, z = p.z;
    if (!pose.identity) {
      const double dx = x, dy = y, dz = z;
      x = (float)pose_row_f64(pose.m + 0, dx, dy, dz);
      y = (float)pose_row_f64(pose.m + 4, dx, dy, dz);
      z = (float)pose_row_f64(pose.m + 8, dx, dy, dz);
    }
    reinterpret_cast<float4*>(moved)[i] = make_float4(x, y, z, p.w);
    mask[i] = (x >= g.lo[0] && x < g.hi[0] && y >= g.lo[1] && y < g.hi[1] && z >= g.lo[2] && z < g.hi[2]) ? 1 : 0;
  }
}

struct EmitArgs {
  const float* moved;     // [n, 4]
  const int32_t* mask;    // [n]
  const int32_t* prefix;  // [n] inclusive prefix sum of mask
  float* xyzi;            // [V, T, 7, N]
  float* coord;           // [V, T, N, 3]
  float* sphere;          // [V, T, N, 2]
  int64_t n, N;
  int t, T, V;
  float sx[4], sy[4];
  PrepGeom g;
};

__device__ __forceinline__ void emit_point(const EmitArgs& a, int64_t j, float x0, float y0, float z, float inten) {
#pragma unroll 1
  for (int v = 0; v < a.V; ++v) {
    const float x = __fmul_rn(x0, a.sx[v]), y = __fmul_rn(y0, a.sy[v]);
    const float qx = __fdiv_rn(__fsub_rn(x, a.g.lo[0]), a.g.cell[0]);
    const float qy = __fdiv_rn(__fsub_rn(y, a.g.lo[1]), a.g.cell[1]);
    const float qz = __fdiv_rn(__fsub_rn(z, a.g.lo[2]), a.g.cell[2]);
    // d = sqrt(x^2 + y^2 + z^2) + 1e-12 in float32, summed left to right
    // the float32 square root is taken in float64 and rounded once: correctly rounded like numpy's (v_sqrt_f32 is not)
    const float ss = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
    const float d = __fadd_rn((float)sqrt((double)ss), 1e-12f);
    const float phi_q = __fdiv_rn(__fsub_rn(a.g.phi_hi, atan2f(x, y)), a.g.dphi);
    const float th_q = __fdiv_rn(__fsub_rn(a.g.th_hi, asinf(__fdiv_rn(z, d))), a.g.dtheta);
    const int64_t s = (int64_t)v * a.T + a.t;
    float* f = a.xyzi + s * 7 * a.N + j;
    f[0] = x;
    f[a.N] = y;
    f[2 * a.N] = z;
    f[3 * a.N] = inten;
    f[4 * a.N] = d;
    f[5 * a.N] = __fsub_rn(qx, floorf(qx));
    f[6 * a.N] = __fsub_rn(qy, floorf(qy));
    float* c = a.coord + (s * a.N + j) * 3;
    c[0] = qx;
    c[1] = qy;
    c[2] = qz;
    float* sp = a.sphere + (s * a.N + j) * 2;
    sp[0] = th_q;
    sp[1] = phi_q;
  }
}

__global__ __launch_bounds__(kBlock) void prep_emit(EmitArgs a) {
  const int64_t count = a.n > 0 ? min((int64_t)a.prefix[a.n - 1], a.N) : 0;
  const int64_t span = max(a.n, a.N);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < span; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < a.n && a.mask[i]) {
      const int64_t j = (int64_t)a.prefix[i] - 1;
      if (j < a.N) {
        const float4 p = reinterpret_cast<const float4*>(a.moved)[i];
        emit_point(a, j, p.x, p.y, p.z, p.w);
      }
    }
    if (i >= count && i < a.N) emit_point(a, i, -1000.0f, -1000.0f, -4000.0f, -1000.0f);
  }
}

__global__ __launch_bounds__(kBlock) void prep_unpad_labels(const uint8_t* __restrict__ labels, int64_t N,
                                                            const int32_t* __restrict__ mask,
                                                            const int32_t* __restrict__ prefix, int64_t n,
                                                            uint8_t* __restrict__ raw) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = (int64_t)prefix[i] - 1;
    raw[i] = (mask[i] && j < N) ? labels[j] : (uint8_t)0;
  }
}

static void fill_geom(PrepGeom& g, const double* range6, const int64_t* bev_size3, const double* rv4) {
  // the reference keeps the cell sizes as Python doubles and numpy rounds them to float32 at use
  for (int d = 0; d < 3; ++d) {
    g.lo[d] = (float)range6[2 * d];
    g.hi[d] = (float)range6[2 * d + 1];
    g.cell[d] = (float)((range6[2 * d + 1] - range6[2 * d]) / (double)bev_size3[d]);
  }
  g.phi_hi = (float)rv4[0];
  g.dphi = (float)rv4[1];
  g.th_hi = (float)rv4[2];
  g.dtheta = (float)rv4[3];
}

}  // namespace smos

using namespace smos;

extern "C" int smos_prep_transform_mask(const float* scan, int64_t n, const double* pose_diff, const double* range6,
                                        float* moved, int32_t* mask, smos_stream_t stream) {
  SMOS_REQUIRE(n >= 0 && range6, "prep_transform_mask: bad arguments");
  if (n == 0) return SMOS_OK;
  SMOS_REQUIRE(scan && moved && mask, "prep_transform_mask: null device pointer");
  PrepPose p;
  p.identity = pose_diff ? 0 : 1;
  for (int i = 0; i < 12; ++i) p.m[i] = pose_diff ? pose_diff[i] : 0.0;
  PrepGeom g;
  const int64_t dummy[3] = {1, 1, 1};
  const double rv[4] = {0, 1, 0, 1};
  fill_geom(g, range6, dummy, rv);
  hipLaunchKernelGGL(prep_transform_mask, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, scan, n, p, g, moved, mask);
  return check_launch("prep_transform_mask");
}

extern "C" int smos_prep_emit(const float* moved, const int32_t* mask, const int32_t* prefix, int64_t n, int32_t t,
                              int32_t T, int64_t N, int32_t V, const float* tta_sx, const float* tta_sy,
                              const double* range6, const int64_t* bev_size3, const double* rv4, float* xyzi, float* coord,
                              float* sphere, smos_stream_t stream) {
  SMOS_REQUIRE(n >= 0 && N > 0 && T > 0 && t >= 0 && t < T && V >= 1 && V <= 4, "prep_emit: bad arguments");
  SMOS_REQUIRE(tta_sx && tta_sy && range6 && bev_size3 && rv4 && xyzi && coord && sphere, "prep_emit: null pointer");
  SMOS_REQUIRE(n == 0 || (moved && mask && prefix), "prep_emit: null scan pointers");
  EmitArgs a;
  a.moved = moved; a.mask = mask; a.prefix = prefix; a.xyzi = xyzi; a.coord = coord; a.sphere = sphere;
  a.n = n; a.N = N; a.t = t; a.T = T; a.V = V;
  for (int v = 0; v < 4; ++v) {
    a.sx[v] = v < V ? tta_sx[v] : 1.f;
    a.sy[v] = v < V ? tta_sy[v] : 1.f;
  }
  fill_geom(a.g, range6, bev_size3, rv4);
  hipLaunchKernelGGL(prep_emit, dim3(grid_for(n > N ? n : N)), dim3(kBlock), 0, (hipStream_t)stream, a);
  return check_launch("prep_emit");
}

extern "C" int smos_prep_unpad_labels(const uint8_t* labels, int64_t N, const int32_t* mask, const int32_t* prefix, int64_t n,
                                      uint8_t* raw, smos_stream_t stream) {
  SMOS_REQUIRE(n >= 0 && N >= 0, "prep_unpad_labels: bad sizes");
  if (n == 0) return SMOS_OK;
  SMOS_REQUIRE(labels && mask && prefix && raw, "prep_unpad_labels: null device pointer");
  hipLaunchKernelGGL(prep_unpad_labels, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, labels, N, mask, prefix, n, raw);
  return check_launch("prep_unpad_labels");
}
