// Sparse first stage of the BEV encoder for gfx950.
//
// header_bev[0] is a DownSample2D (networks/backbone.py:136-159) on the 192-channel 512 x 512 input grid:
//     out = relu( conv3x3_s2(x) + maxpool3x3_s2( conv1x1(x) ) + bias )          (BatchNorm folded)
// -- 10.5 GFLOP per sample, the largest block after conv_1, on a grid of which about one cell in five
// is occupied: a max-pool scatter leaves every cell no LiDAR point fell into at exactly 0, and both convolutions are
// linear, so empty cells contribute nothing.  The engine therefore
//   1. marks the occupied cells from the point coordinates (stem_mark, this file),
//   2. compacts them with one prefix sum (stem_compact; rows ordered by the PARITY CLASS of the cell: with stride 2 a
//      cell at odd y feeds kernel rows ky in {0, 2}, a cell at even y only ky = 1, same for x -- 4, 2, 2 or 1 of the 9
//      taps, 2.25 on average),
//   3. multiplies the occupied rows [n_c, 192], read in place from the grid, with the class's tap weights
//      [(taps_c + 1) * 32, 192] on the matrix cores (stem_gemm; the "+ 1" block is the 1x1 pool-branch convolution);
//      the row counts stay on the device, so there is no host round trip and the sequence can be graph-captured,
//   4. and assembles the output per pixel in a fixed tap order (stem_epilogue): deterministic, no atomics.
// 5 x fewer FLOPs than the dense convolutions and the 805 MB grid is read only where it is occupied.
#include "smos_common.h"
#include <hipcub/hipcub.hpp>

namespace smos {

// Parity-class-major ("permuted") index of a cell: rows of the compacted list come out grouped by class, then sample,
// then position, from ONE exclusive prefix sum over the flags.
__device__ __forceinline__ int64_t perm_index(int b, int y, int x, int B, int H, int W) {
  const int cls = (y & 1) * 2 + (x & 1), hh = H >> 1, wh = W >> 1;
  return (((int64_t)cls * B + b) * hh + (y >> 1)) * wh + (x >> 1);
}

// flags[perm_index(b, y, x)] = 1 for every cell a point of any of the T frames falls into (same cell rule as the
// scatter, point_deep_cuda_kernel.cu:39-47: valid when -1 < coord < size, cell = trunc(coord))
__global__ __launch_bounds__(kBlock) void stem_mark(const float* __restrict__ coord, int K, int B, int T, int64_t N, int H, int W,
                                                    int32_t* __restrict__ flags) {
  const int64_t total = (int64_t)B * T * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float py = coord[i * K], px = coord[i * K + 1];
    const bool ok = (py > -1.0f) && (py < (float)H) && (px > -1.0f) && (px < (float)W);
    if (!ok) continue;
    const int b = (int)(i / ((int64_t)T * N));
    flags[perm_index(b, (int)py, (int)px, B, H, W)] = 1;
  }
}

// scan = exclusive prefix sum of flags (permuted order).  row_cell[row] = natural cell id (b*H + y)*W + x of the row;
// row_of[natural cell] = row or -1; meta[0..3] = rows per class, meta[4..7] = first row of each class.
__global__ __launch_bounds__(kBlock) void stem_rows(const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int B, int H,
                                                    int W, int32_t* __restrict__ row_cell, int32_t* __restrict__ row_of,
                                                    int32_t* __restrict__ meta) {
  const int hh = H >> 1, wh = W >> 1;
  const int64_t per = (int64_t)B * hh * wh, total = 4 * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cls = (int)(i / per);
    int64_t r = i - (int64_t)cls * per;
    const int b = (int)(r / ((int64_t)hh * wh));
    r -= (int64_t)b * hh * wh;
    const int y = (int)(r / wh) * 2 + (cls >> 1), x = (int)(r % wh) * 2 + (cls & 1);
    const int32_t cell = (b * H + y) * W + x;
    const int32_t f = flags[i], at = scan[i];
    row_of[cell] = f ? at : -1;
    if (f) row_cell[at] = cell;
    if (r == 0 && b == 0) meta[4 + cls] = at;                              // first entry of the class
    if (i == (int64_t)(cls + 1) * per - 1) meta[8 + cls] = at + f;          // one past the class's last row
  }
}

__global__ void stem_meta(int32_t* meta) {
  if (threadIdx.x < 4) meta[threadIdx.x] = meta[8 + threadIdx.x] - meta[4 + threadIdx.x];
}

// zero-fill of the compact row table: only the rows that exist (their number is on the device)
__global__ __launch_bounds__(kBlock) void stem_zero_rows(float4* __restrict__ rows, const int32_t* __restrict__ meta, int row_f4) {
  const int64_t total = (int64_t)meta[11] * row_f4;    // meta[8 + 3] = one past the last row of the last class
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    rows[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Y_cls[r][mt*32 + c] = sum_k W_cls[mt*32 + c][k] * X[row_cell[start + r]][k]   (k = 0..191) on the matrix cores, in the
// transposed form of pointnet_scatter (point_fused.hip): output channel on the MFMA row, cell on the column.
//   A operand (weights): whole class in LDS as [mt][k-step s][lane], lane (m, h) holding W[mt*32+m][h*96+s];
//   B operand (cells):   lane (p, h) holds channels h*96 .. h*96+95 of cell p -- 24 contiguous float4 loads of its row.
// kM = taps + 1 blocks of 32 output channels.  The row count is read from device memory (meta): no host round trip.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kStemK = 192, kStemSteps = kStemK / 2, kStemBlock = 512;

template <int kM>
__device__ __forceinline__ void stem_gemm_class(const float* __restrict__ bev, const int32_t* __restrict__ row_cell,
                                                const int32_t* __restrict__ meta, int cls, const float* __restrict__ wprep,
                                                float* __restrict__ y, float* lds_w, int block, int n_blocks) {
  for (int i = threadIdx.x; i < kM * kStemSteps * 64; i += kStemBlock) lds_w[i] = wprep[i];   // [kM][96][64]
  __syncthreads();
  const int n = meta[cls], start = meta[4 + cls];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int col = lane & 31, hh = lane >> 5;
  constexpr int kWaves = kStemBlock / 64;
  for (int tile = block * kWaves + wave; tile * 32 < n; tile += n_blocks * kWaves) {
    const int r = tile * 32 + col;
    const bool valid = r < n;
    // row_cell == null: bev already IS the compact row table (smos_pointnet_scatter_rows), row = start + r
    const int32_t cell = valid ? (row_cell ? row_cell[start + r] : start + r) : 0;
    const float4* src = reinterpret_cast<const float4*>(bev + (int64_t)cell * kStemK + hh * kStemSteps);
    // K in four quarters of 24 steps: the next quarter's 6 float4 of the row are in flight while the current quarter
    // feeds the matrix core (keeping all 96 row values live at once spills under the 256-register budget)
    constexpr int kQ = 4, kQSteps = kStemSteps / kQ;
    float4 cur[kQSteps / 4], nxt[kQSteps / 4];
#pragma unroll
    for (int j = 0; j < kQSteps / 4; ++j) cur[j] = src[j];
    f32x16 acc[kM];
#pragma unroll
    for (int mt = 0; mt < kM; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][q] = 0.0f;
#pragma unroll 1
    for (int qt = 0; qt < kQ; ++qt) {
      if (qt + 1 < kQ) {
#pragma unroll
        for (int j = 0; j < kQSteps / 4; ++j) nxt[j] = src[(qt + 1) * (kQSteps / 4) + j];
      }
      const float* wq = lds_w + (qt * kQSteps) * 64 + lane;
#pragma unroll
      for (int s = 0; s < kQSteps; ++s) {
        const float4 v = cur[s >> 2];
        const float b = (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
#pragma unroll
        for (int mt = 0; mt < kM; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[(mt * kStemSteps + s) * 64], b, acc[mt], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < kQSteps / 4; ++j) cur[j] = nxt[j];
    }
    if (valid) {
      float* dst = y + (int64_t)r * (kM * 32) + 4 * hh;
#pragma unroll
      for (int mt = 0; mt < kM; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(dst + mt * 32 + 8 * g) =
              make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
    }
  }
}

// ONE launch for the four parity classes: the blocks are split between the classes in proportion to their work
// (equal row counts by symmetry, 2 : 3 : 3 : 5 output blocks), every block keeps its class's weights in LDS.
struct StemGemmArgs {
  const float* bev;
  const int32_t* row_cell;
  const int32_t* meta;
  const float* wprep[4];
  float* y[4];
  int first_block[5];
};

__global__ __launch_bounds__(kStemBlock) void stem_gemm(StemGemmArgs a) {
  extern __shared__ float lds_w[];
  const int bid = blockIdx.x;
  if (bid < a.first_block[1])
    stem_gemm_class<2>(a.bev, a.row_cell, a.meta, 0, a.wprep[0], a.y[0], lds_w, bid - a.first_block[0], a.first_block[1] - a.first_block[0]);
  else if (bid < a.first_block[2])
    stem_gemm_class<3>(a.bev, a.row_cell, a.meta, 1, a.wprep[1], a.y[1], lds_w, bid - a.first_block[1], a.first_block[2] - a.first_block[1]);
  else if (bid < a.first_block[3])
    stem_gemm_class<3>(a.bev, a.row_cell, a.meta, 2, a.wprep[2], a.y[2], lds_w, bid - a.first_block[2], a.first_block[3] - a.first_block[2]);
  else
    stem_gemm_class<5>(a.bev, a.row_cell, a.meta, 3, a.wprep[3], a.y[3], lds_w, bid - a.first_block[3], a.first_block[4] - a.first_block[3]);
}

struct StemEpiArgs {
  const float* y[4];       // per parity class c = (y & 1) * 2 + (x & 1): rows [n_c, (taps_c + 1) * C]
  const int32_t* meta;     // meta[4 + c] = global row id of the class's first row
  const int32_t* row_of;   // [B, H, W] global row id of an occupied cell, -1 otherwise
  const float* bias;       // [C]
  float* out;              // [B, Ho, Wo, *] channels-last, row pitch op
  int64_t op;
  int B, H, W, Ho, Wo;
};

// lane = channel (C = 32): a wave handles two output pixels per iteration.
// out[b, ho, wo, c] = relu( sum_{ky,kx} Y[cell(2ho-1+ky, 2wo-1+kx)][slot(ky,kx)][c]
//                           + max_{valid window cells}( occupied ? Y[cell][q slot][c] : 0 ) + bias[c] )
template <int kC>
__global__ __launch_bounds__(kBlock) void stem_epilogue(StemEpiArgs a) {
  const int c = threadIdx.x % kC;
  const int64_t n_out = (int64_t)a.B * a.Ho * a.Wo;
  const float bias = a.bias[c];
  const int start[4] = {a.meta[4], a.meta[5], a.meta[6], a.meta[7]};
  for (int64_t o = (int64_t)blockIdx.x * (kBlock / kC) + threadIdx.x / kC; o < n_out; o += (int64_t)gridDim.x * (kBlock / kC)) {
    const int wo = (int)(o % a.Wo);
    const int64_t t = o / a.Wo;
    const int ho = (int)(t % a.Ho), b = (int)(t / a.Ho);
    // all nine row ids first, then all eighteen loads (an empty or out-of-range tap reads row 0 of its class and is
    // discarded by the selects), then the sums in fixed tap order
    int rid[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int y = 2 * ho - 1 + t9 / 3, x = 2 * wo - 1 + t9 % 3;
      const bool inb = y >= 0 && y < a.H && x >= 0 && x < a.W;
      rid[t9] = inb ? a.row_of[((int64_t)b * a.H + y) * a.W + x] : -2;     // -1: empty cell, -2: outside the grid
    }
    float va[9], vq[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int ky = t9 / 3, kx = t9 % 3;
      // a cell reached through an even ky has odd y, i.e. belongs to a class with two kernel rows (slots ky / 2)
      const int ey = (ky & 1) ^ 1, ex = (kx & 1) ^ 1;
      const int cls = ey * 2 + ex, taps = (1 + ey) * (1 + ex);
      const int slot = (ey ? ky >> 1 : 0) * (1 + ex) + (ex ? kx >> 1 : 0);
      const float* row = a.y[cls] + (int64_t)max(rid[t9] - start[cls], 0) * ((taps + 1) * kC);
      va[t9] = row[slot * kC + c];
      vq[t9] = row[taps * kC + c];
    }
    float acc = 0.0f, qmax = -INFINITY;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      acc = rid[t9] >= 0 ? acc + va[t9] : acc;
      // an empty cell inside the grid contributes 0 to the pooled branch; outside the grid nothing (padding of the pool)
      qmax = rid[t9] >= 0 ? fmaxf(qmax, vq[t9]) : (rid[t9] == -1 ? fmaxf(qmax, 0.0f) : qmax);
    }
    a.out[o * a.op + c] = fmaxf((acc + qmax) + bias, 0.0f);
  }
}

}  // namespace smos

using namespace smos;

extern "C" int smos_stem_mark(const float* coord, int32_t K, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W, int32_t* flags,
                              smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && T > 0 && N >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && K >= 2 && B * H * W < (1LL << 31),
               "stem_mark: bad sizes (H and W must be even)");
  if (N == 0) return SMOS_OK;
  SMOS_REQUIRE(coord && flags, "stem_mark: null device pointer");
  hipLaunchKernelGGL(stem_mark, dim3(grid_for(B * T * N)), dim3(kBlock), 0, (hipStream_t)stream, coord, (int)K, (int)B, (int)T, N,
                     (int)H, (int)W, flags);
  return check_launch("stem_mark");
}

extern "C" int64_t smos_stem_scan_bytes(int64_t cells) {
  if (cells <= 0 || cells >= (1LL << 31)) return -1;
  size_t bytes = 0;
  if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int)cells) != hipSuccess) return -1;
  return (int64_t)bytes;
}

extern "C" int smos_stem_compact(const int32_t* flags, int64_t B, int64_t H, int64_t W, int32_t* scan, void* scan_ws,
                                 int64_t scan_ws_bytes, int32_t* row_cell, int32_t* row_of, int32_t* meta, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && B * H * W < (1LL << 31), "stem_compact: bad sizes");
  SMOS_REQUIRE(flags && scan && scan_ws && row_cell && row_of && meta, "stem_compact: null device pointer");
  const int64_t cells = B * H * W;
  size_t need = 0;
  SMOS_REQUIRE(hipcub::DeviceScan::ExclusiveSum(nullptr, need, flags, scan, (int)cells) == hipSuccess && (int64_t)need <= scan_ws_bytes,
               "stem_compact: scan workspace too small");
  hipStream_t s = (hipStream_t)stream;
  size_t bytes = (size_t)scan_ws_bytes;
  if (hipcub::DeviceScan::ExclusiveSum(scan_ws, bytes, flags, scan, (int)cells, s) != hipSuccess) {
    set_error("stem_compact: prefix sum failed");
    return SMOS_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(stem_rows, dim3(grid_for(cells)), dim3(kBlock), 0, s, flags, (const int32_t*)scan, (int)B, (int)H, (int)W, row_cell,
                     row_of, meta);
  hipLaunchKernelGGL(stem_meta, dim3(1), dim3(64), 0, s, meta);
  return check_launch("stem_compact");
}

extern "C" int smos_stem_gemm(const float* bev, const int32_t* row_cell, const int32_t* meta, const float* const* wprep4,
                              float* const* y4, int64_t Cin, int64_t Cout, smos_stream_t stream) {
  SMOS_REQUIRE(Cin == kStemK && Cout == 32, "stem_gemm: built for 192 -> 32 channels");
  SMOS_REQUIRE(bev && meta && wprep4 && y4, "stem_gemm: null pointer");
  for (int c = 0; c < 4; ++c) SMOS_REQUIRE(wprep4[c] && y4[c], "stem_gemm: null class pointer");
  const size_t lds = (size_t)5 * kStemSteps * 64 * sizeof(float);   // the largest class: 4 taps + the pool branch
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&stem_gemm), lds, 0, &ks, "stem_gemm")) return rc;
  const int cus = ks.cus;
  StemGemmArgs a;
  a.bev = bev; a.row_cell = row_cell; a.meta = meta;
  for (int c = 0; c < 4; ++c) {
    a.wprep[c] = wprep4[c];
    a.y[c] = y4[c];
  }
  // one block per CU (120 KB of LDS); shares 2 : 3 : 3 : 5 of at least one block each
  const int blocks = cus < 4 ? 4 : cus;
  const int share[4] = {2, 3, 3, 5};
  int at = 0;
  for (int c = 0; c < 4; ++c) {
    a.first_block[c] = at;
    int nb = blocks * share[c] / 13;
    at += nb < 1 ? 1 : nb;
  }
  a.first_block[4] = at;
  hipLaunchKernelGGL(stem_gemm, dim3(at), dim3(kStemBlock), lds, (hipStream_t)stream, a);
  return check_launch("stem_gemm");
}

extern "C" int smos_stem_zero_rows(float* rows, const int32_t* meta, int64_t row_floats, smos_stream_t stream) {
  SMOS_REQUIRE(rows && meta && row_floats > 0 && row_floats % 4 == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0,
               "stem_zero_rows: bad arguments");
  hipLaunchKernelGGL(stem_zero_rows, dim3(256 * 8), dim3(kBlock), 0, (hipStream_t)stream, reinterpret_cast<float4*>(rows), meta,
                     (int)(row_floats / 4));
  return check_launch("stem_zero_rows");
}

extern "C" int smos_stem_epilogue(const float* const* y4, const int32_t* meta, const int32_t* row_of, const float* bias,
                                  float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t C, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 1 && W > 1 && C == 32 && out_pitch >= C, "stem_epilogue: bad sizes (C must be 32)");
  SMOS_REQUIRE(y4 && meta && row_of && bias && out, "stem_epilogue: null pointer");
  StemEpiArgs a;
  for (int k = 0; k < 4; ++k) a.y[k] = y4[k];
  a.meta = meta; a.row_of = row_of; a.bias = bias; a.out = out; a.op = out_pitch;
  a.B = (int)B; a.H = (int)H; a.W = (int)W;
  a.Ho = (int)((H + 2 - 3) / 2 + 1);
  a.Wo = (int)((W + 2 - 3) / 2 + 1);
  const int64_t n_out = B * a.Ho * a.Wo;
  hipLaunchKernelGGL((stem_epilogue<32>), dim3(grid_for(n_out * 32, kBlock, 256 * 32)), dim3(kBlock), 0, (hipStream_t)stream, a);
  return check_launch("stem_epilogue");
}
