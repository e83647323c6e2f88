// Grid -> point bilinear gather for gfx950.
// Replaces networks/backbone.py:453-475 (BilinearSample: F.grid_sample(bilinear, zeros,
// align_corners=True) on a grid built as 2*coord*scale/(size-1) - 1).  The float32 normalise /
// un-normalise round trip of that formulation is reproduced operation by operation (__f*_rn
// intrinsics, so the compiler neither contracts nor reassociates it): the sampling position then
// matches the reference's to the last bit and only the four-tap blend differs by FMA rounding.
#include "smos_common.h"

namespace smos {

struct BilGeom {
  int64_t gs_b, gs_c, gs_h, gs_w;  // grid strides
  int64_t os_b, os_c, os_n;        // out strides
  int32_t H, W, K;
  float sy, sx;
};

struct Taps {
  int64_t o[4];
  float w[4];
};

// pixel position exactly as backbone.py:467-468 + ATen's grid_sampler_unnormalize(align_corners=True)
__device__ __forceinline__ float pixel_pos(float c, float s, int size) {
  const float sm1 = (float)(size - 1);
  float gn = __fsub_rn(__fdiv_rn(__fmul_rn(__fmul_rn(2.0f, c), s), sm1), 1.0f);
  return __fmul_rn(__fdiv_rn(__fadd_rn(gn, 1.0f), 2.0f), sm1);
}

__device__ __forceinline__ Taps make_taps(const float* __restrict__ crow, const BilGeom& g) {
  Taps t;
  const float iy = pixel_pos(crow[0], g.sy, g.H);
  const float ix = pixel_pos(crow[1], g.sx, g.W);
  const float fy = floorf(iy), fx = floorf(ix);
  // weights in ATen's order: nw, ne, sw, se (GridSampler: nw = (ix_se - ix) * (iy_se - iy), ...)
  const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix;
  const float wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
  t.w[0] = wx0 * wy0;
  t.w[1] = wx1 * wy0;
  t.w[2] = wx0 * wy1;
  t.w[3] = wx1 * wy1;
  // NaN / far-away coordinates (the reference's -1000 / -4000 padding rows) fail every range test
  const bool finite = (iy > -2.0f) && (iy < (float)(g.H + 1)) && (ix > -2.0f) && (ix < (float)(g.W + 1));
  const int y0 = finite ? (int)fy : -5, x0 = finite ? (int)fx : -5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = y0 + (k >> 1), x = x0 + (k & 1);
    const bool in = (y >= 0) && (y < g.H) && (x >= 0) && (x < g.W);
    t.o[k] = in ? (int64_t)y * g.gs_h + (int64_t)x * g.gs_w : (int64_t)-1;
    if (!in) t.w[k] = 0.0f;
  }
  return t;
}

// lane = point, loop over channels: scattered tap reads (neighbouring points share cache lines),
// coalesced writes when out is channel-major [B,C,N]
__global__ __launch_bounds__(kBlock) void bil_points(const float* __restrict__ grid, const float* __restrict__ coord,
                                                     float* __restrict__ out, BilGeom g, int64_t B, int64_t C,
                                                     int64_t N, int c_chunk) {
  const int64_t total = B * N;
  const int c0 = blockIdx.y * c_chunk;
  const int c1 = (int)min((int64_t)(c0 + c_chunk), C);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / N, n = i - b * N;
    const Taps t = make_taps(coord + i * g.K, g);
    const float* __restrict__ gb = grid + b * g.gs_b;
    float* __restrict__ ob = out + b * g.os_b + n * g.os_n;
    for (int c = c0; c < c1; ++c) {
      const float* gc = gb + (int64_t)c * g.gs_c;
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t.o[k] >= 0) acc += gc[t.o[k]] * t.w[k];
      ob[(int64_t)c * g.os_c] = acc;
    }
  }
}

// lane = channel, a group of kG lanes per point: channels-last grid and point-major out -> every tap
// and every store is one contiguous row
template <int kG>
__global__ __launch_bounds__(kBlock) void bil_rows(const float* __restrict__ grid, const float* __restrict__ coord,
                                                   float* __restrict__ out, BilGeom g, int64_t B, int64_t C,
                                                   int64_t N) {
  constexpr int kGroups = kBlock / kG;
  const int lane = threadIdx.x % kG, grp = threadIdx.x / kG;
  const int64_t total = B * N;
  for (int64_t i = (int64_t)blockIdx.x * kGroups + grp; i < total; i += (int64_t)gridDim.x * kGroups) {
    const int64_t b = i / N, n = i - b * N;
    const Taps t = make_taps(coord + i * g.K, g);
    const float* __restrict__ gb = grid + b * g.gs_b;
    float* __restrict__ ob = out + b * g.os_b + n * g.os_n;
    for (int c = lane; c < C; c += kG) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t.o[k] >= 0) acc += gb[t.o[k] + c] * t.w[k];
      ob[c] = acc;
    }
  }
}

}  // namespace smos

using namespace smos;

extern "C" int smos_bilinear_gather_fwd(const float* grid, const int64_t* grid_stride, const float* coord, int32_t K,
                                        float* out, const int64_t* out_stride, int64_t B, int64_t C, int64_t H,
                                        int64_t W, int64_t N, const float* scale, smos_stream_t stream) {
  SMOS_REQUIRE(B >= 0 && C >= 0 && N >= 0 && H > 0 && W > 0, "bilinear_gather: bad sizes");
  SMOS_REQUIRE(H < (1 << 24) && W < (1 << 24), "bilinear_gather: grid too large");
  SMOS_REQUIRE(K >= 2, "bilinear_gather: coord needs >= 2 columns, got %d", (int)K);
  if (B == 0 || C == 0 || N == 0) return SMOS_OK;
  SMOS_REQUIRE(grid && coord && out && grid_stride && out_stride && scale, "bilinear_gather: null pointer");
  BilGeom g;
  g.gs_b = grid_stride[0]; g.gs_c = grid_stride[1]; g.gs_h = grid_stride[2]; g.gs_w = grid_stride[3];
  g.os_b = out_stride[0]; g.os_c = out_stride[1]; g.os_n = out_stride[2];
  g.H = (int)H; g.W = (int)W; g.K = K; g.sy = scale[0]; g.sx = scale[1];
  hipStream_t s = (hipStream_t)stream;
  const int64_t pts = B * N;
  if (g.gs_c == 1 && g.os_c == 1 && C >= 8) {
    int G = 8;
    while (G < 64 && G < C) G <<= 1;
    dim3 gr(grid_for(pts * G, kBlock, 256 * 16));
    switch (G) {
      case 8: hipLaunchKernelGGL((bil_rows<8>), gr, dim3(kBlock), 0, s, grid, coord, out, g, B, C, N); break;
      case 16: hipLaunchKernelGGL((bil_rows<16>), gr, dim3(kBlock), 0, s, grid, coord, out, g, B, C, N); break;
      case 32: hipLaunchKernelGGL((bil_rows<32>), gr, dim3(kBlock), 0, s, grid, coord, out, g, B, C, N); break;
      default: hipLaunchKernelGGL((bil_rows<64>), gr, dim3(kBlock), 0, s, grid, coord, out, g, B, C, N); break;
    }
  } else {
    int chunks = 1;
    if (pts < (1 << 20) && C >= 16) {
      chunks = (int)(((int64_t)(1 << 20) + pts - 1) / pts);
      if (chunks > C / 8) chunks = (int)(C / 8);
      if (chunks < 1) chunks = 1;
    }
    const int c_chunk = (int)((C + chunks - 1) / chunks);
    chunks = (int)((C + c_chunk - 1) / c_chunk);
    hipLaunchKernelGGL(bil_points, dim3(grid_for(pts, kBlock, 256 * 16), chunks), dim3(kBlock), 0, s, grid, coord,
                       out, g, B, C, N, c_chunk);
  }
  return check_launch("bilinear_gather_fwd");
}
