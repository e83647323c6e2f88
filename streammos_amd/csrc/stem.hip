// Sparse first stage of the BEV encoder for gfx950.
//
// header_bev[0] is a DownSample2D (networks/backbone.py:136-159) on the 192-channel 512 x 512 input grid:
//     out = relu( conv3x3_s2(x) + maxpool3x3_s2( conv1x1(x) ) + bias )          (BatchNorm folded)
// -- 10.5 GFLOP per sample, the largest block after conv_1, on a grid of which about one cell in five
// is occupied: a max-pool scatter leaves every cell no LiDAR point fell into at exactly 0, and both convolutions are
// linear, so empty cells contribute nothing.  The engine therefore
//   1. marks the occupied cells from the point coordinates (stem_mark, this file),
//   2. compacts them with one single-pass prefix sum (stem_scan; rows ordered by the PARITY CLASS of the cell: with stride 2 a
//      cell at odd y feeds kernel rows ky in {0, 2}, a cell at even y only ky = 1, same for x -- 4, 2, 2 or 1 of the 9
//      taps, 2.25 on average),
//   3. multiplies the occupied rows [n_c, 192], read in place from the grid, with the class's tap weights
//      [(taps_c + 1) * 32, 192] on the matrix cores (stem_gemm; the "+ 1" block is the 1x1 pool-branch convolution);
//      the row counts stay on the device, so there is no host round trip and the sequence can be graph-captured,
//   4. and assembles the output per pixel in a fixed tap order (stem_epilogue): deterministic, no atomics.
// 5 x fewer FLOPs than the dense convolutions and the 805 MB grid is read only where it is occupied.
#include "smos_common.h"

namespace smos {

// Parity-class-major ("permuted") index of a cell: rows of the compacted list come out grouped by class, then sample,
// then position, from ONE exclusive prefix sum over the flags.
__device__ __forceinline__ int64_t perm_index(int b, int y, int x, int B, int H, int W) {
  const int cls = (y & 1) * 2 + (x & 1), hh = H >> 1, wh = W >> 1;
  return (((int64_t)cls * B + b) * hh + (y >> 1)) * wh + (x >> 1);
}

// Single-pass scan state (decoupled look-back): tile descriptors = one 64-bit word {status, value} each, written and read
// with ONE agent-scope 8-byte atomic (the value travels inside the flag word, so no fence is needed), plus the ticket
// counter that hands out tile ids in execution order (a tile only ever waits for tiles that have already started).
constexpr int kScanItems = 8, kScanTile = kBlock * kScanItems, kScanMaxTiles = 1 << 16;
constexpr unsigned long long kScanAggregate = 1ULL << 32, kScanInclusive = 2ULL << 32;

// flags[perm_index(b, y, x)] = 1 for every cell a point of any of the T frames falls into (same cell rule as the
// scatter, point_deep_cuda_kernel.cu:39-47: valid when -1 < coord < size, cell = trunc(coord)).  flags must be all zero
// on entry; stem_scan, which consumes them, leaves them all zero again.  Block 0 also re-arms the scan state of the
// stem_scan launch that follows on the stream (no memset launch, nothing frozen into a captured graph).
__global__ __launch_bounds__(kBlock) void stem_mark(const float* __restrict__ coord, int K, int B, int T, int64_t N, int H, int W,
                                                    int32_t* __restrict__ flags, unsigned long long* __restrict__ scan_state,
                                                    int n_tiles) {
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i <= n_tiles; i += kBlock) scan_state[i] = 0;      // [0] = ticket counter, [1 + t] = tile t
  const int64_t total = (int64_t)B * T * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float py = coord[i * K], px = coord[i * K + 1];
    const bool ok = (py > -1.0f) && (py < (float)H) && (px > -1.0f) && (px < (float)W);
    if (!ok) continue;
    const int b = (int)(i / ((int64_t)T * N));
    flags[perm_index(b, (int)py, (int)px, B, H, W)] = 1;
  }
}

// One pass over the flags (permuted order = row order): exclusive prefix sum by decoupled look-back, then per cell
// row_of[natural cell] = row or -1, row_cell[row] = natural cell id (b*H + y)*W + x, the flags cleared for the next frame,
// the class boundaries in meta (meta[4 + c] = first row of class c, meta[8 + c] = one past its last row, so meta[11] = the
// number of rows), and the zero fill of exactly the rows this tile creates in the compact table (rows, row_f4 float4 per
// row; may be null).  Replaces a 3-kernel library scan + 3 bookkeeping kernels + the zero-fill launch.
__global__ __launch_bounds__(kBlock) void stem_scan(int32_t* __restrict__ flags, int B, int H, int W,
                                                    unsigned long long* __restrict__ scan_state, int32_t* __restrict__ row_cell,
                                                    int32_t* __restrict__ row_of, int32_t* __restrict__ meta,
                                                    float4* __restrict__ rows, int row_f4) {
  __shared__ int s_tile, s_wave[kBlock / 64], s_prefix;
  const int hh = H >> 1, wh = W >> 1;
  const int64_t per = (int64_t)B * hh * wh, total = 4 * per;
  if (threadIdx.x == 0) s_tile = (int)atomicAdd(scan_state, 1ULL);
  __syncthreads();
  const int tile = s_tile;
  const int64_t i0 = (int64_t)tile * kScanTile + (int64_t)threadIdx.x * kScanItems;
  // this thread's kScanItems consecutive flags (total is a multiple of 4; tiles may end ragged)
  int f[kScanItems], count = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k += 4) {
    int4 v = make_int4(0, 0, 0, 0);
    if (i0 + k < total) {
      v = *reinterpret_cast<const int4*>(flags + i0 + k);
      *reinterpret_cast<int4*>(flags + i0 + k) = make_int4(0, 0, 0, 0);
    }
    f[k] = v.x; f[k + 1] = v.y; f[k + 2] = v.z; f[k + 3] = v.w;
    count += v.x + v.y + v.z + v.w;
  }
  // block-wide exclusive scan of the per-thread counts: wave shuffles, then the four wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = count;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int wave_off = 0, aggregate = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    if (w < wave) wave_off += s_wave[w];
    aggregate += s_wave[w];
  }
  // decoupled look-back (one lane): publish the aggregate, walk back until a tile that knows its inclusive prefix
  if (threadIdx.x == 0) {
    unsigned long long* desc = scan_state + 1;
    int prefix = 0;
    if (tile > 0) {
      __hip_atomic_store(desc + tile, kScanAggregate | (unsigned)aggregate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int t = tile - 1;; --t) {
        unsigned long long d;
        while (((d = __hip_atomic_load(desc + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) == 0)
          __builtin_amdgcn_s_sleep(1);
        prefix += (int)(unsigned)d;
        if ((d >> 32) == 2) break;
      }
    }
    __hip_atomic_store(desc + tile, kScanInclusive | (unsigned)(prefix + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_prefix = prefix;
  }
  __syncthreads();
  const int tile_first = s_prefix;
  int at = tile_first + wave_off + incl - count;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = i0 + k;
    if (i < total) {
      const int cls = (int)(i / per);
      int64_t r = i - (int64_t)cls * per;
      const int b = (int)(r / ((int64_t)hh * wh));
      r -= (int64_t)b * hh * wh;
      const int y = (int)(r / wh) * 2 + (cls >> 1), x = (int)(r % wh) * 2 + (cls & 1);
      const int32_t cell = (b * H + y) * W + x;
      row_of[cell] = f[k] ? at : -1;
      if (f[k]) row_cell[at] = cell;
      if (i == (int64_t)cls * per) meta[4 + cls] = at;                           // first cell of the class
      if (i == (int64_t)(cls + 1) * per - 1) meta[8 + cls] = at + f[k];          // its last cell
      at += f[k];
    }
  }
  // zero fill of the rows this tile created: [tile_first, tile_first + aggregate)
  if (rows) {
    float4* dst = rows + (int64_t)tile_first * row_f4;
    const int64_t n = (int64_t)aggregate * row_f4;
    for (int64_t j = threadIdx.x; j < n; j += kBlock) dst[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// Y_cls[r][mt*32 + c] = sum_k W_cls[mt*32 + c][k] * X[row_cell[start + r]][k]   (k = 0..191) on the matrix cores, in the
// transposed form of pointnet_scatter (point_fused.hip): output channel on the MFMA row, cell on the column.
//   A operand (weights): whole class in LDS as [mt][k-step s][lane], lane (m, h) holding W[mt*32+m][h*96+s];
//   B operand (cells):   lane (p, h) holds channels h*96 .. h*96+95 of cell p -- 24 contiguous float4 loads of its row.
// kM = taps + 1 blocks of 32 output channels.  The row count is read from device memory (meta): no host round trip.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Diagnostic builds only (tools/ablate_stem.sh): -DSMOS_STEM_ABLATE=<bits> removes 1 the row loads, 2 the MFMAs, 4 the Y stores
// from stem_gemm to time what is left; results are wrong.  The shipped library is built without it.
#ifdef SMOS_STEM_ABLATE
#define STEM_AB(bit) ((SMOS_STEM_ABLATE) & (bit))
#else
#define STEM_AB(bit) 0
#endif
constexpr int kStemK = 192, kStemSteps = kStemK / 2, kStemBlock = 512;

// One 32-cell tile x NM blocks of 32 output channels starting at block mt0: acc[NM], the cells' 192 values streamed in four
// K-quarters (the next quarter's 6 float4 of the row are in flight while the current one feeds the matrix core; keeping all 96
// row values live at once spills under the 256-register budget).  ldy = the class's row length (kM * 32).
template <int NM>
__device__ __forceinline__ void stem_tile(const float* __restrict__ bev, const int32_t* __restrict__ row_cell, int start, int n,
                                          int tile, int mt0, int ldy, const float* lds_w, float* __restrict__ y, int lane) {
  const int col = lane & 31, hh = lane >> 5;
  const int r = tile * 32 + col;
  const bool valid = r < n;
  // row_cell == null: bev already IS the compact row table (smos_pointnet_scatter_rows), row = start + r
  const int32_t cell = valid ? (row_cell ? row_cell[start + r] : start + r) : 0;
  const float4* src = reinterpret_cast<const float4*>(bev + (int64_t)cell * kStemK + hh * kStemSteps);
  constexpr int kQ = 4, kQSteps = kStemSteps / kQ;
  float4 cur[kQSteps / 4], nxt[kQSteps / 4];
#pragma unroll
  for (int j = 0; j < kQSteps / 4; ++j) cur[j] = STEM_AB(1) ? make_float4((float)lane, 1.f, 2.f, (float)j) : src[j];
  f32x16 acc[NM];
#pragma unroll
  for (int mt = 0; mt < NM; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[mt][q] = 0.0f;
#pragma unroll 1
  for (int qt = 0; qt < kQ; ++qt) {
    if (qt + 1 < kQ) {
#pragma unroll
      for (int j = 0; j < kQSteps / 4; ++j)
        nxt[j] = STEM_AB(1) ? make_float4((float)lane, 1.f, (float)qt, (float)j) : src[(qt + 1) * (kQSteps / 4) + j];
    }
    const float* wq = lds_w + (mt0 * kStemSteps + qt * kQSteps) * 64 + lane;
#pragma unroll
    for (int s = 0; s < kQSteps; ++s) {
      const float4 v = cur[s >> 2];
      const float b = (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
#pragma unroll
      for (int mt = 0; mt < NM; ++mt) {
        if (STEM_AB(2)) acc[mt][s & 15] += wq[(mt * kStemSteps + s) * 64] * b;
        else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[(mt * kStemSteps + s) * 64], b, acc[mt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < kQSteps / 4; ++j) cur[j] = nxt[j];
  }
  if (valid && !(STEM_AB(4) && acc[0][0] != 12345.678f)) {
    float* dst = y + (int64_t)r * ldy + mt0 * 32 + 4 * hh;
#pragma unroll
    for (int mt = 0; mt < NM; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst + mt * 32 + 8 * g) =
            make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
  }
}

// A wave's share [lo, hi) of a class's work units (unit u = (tile u / kM, output block u % kM)): whole tiles with all kM
// accumulators; a ragged head or tail is ONE pass over the tile's rows with as many accumulators as it has output blocks (the
// two waves that share a tile each read its rows once).
template <int kM>
__device__ __forceinline__ void stem_part(const float* __restrict__ bev, const int32_t* __restrict__ row_cell, int start, int n,
                                          int tile, int mt0, int cnt, const float* lds_w, float* __restrict__ y, int lane) {
  switch (cnt) {                                           // wave-uniform
    case 1: stem_tile<1>(bev, row_cell, start, n, tile, mt0, kM * 32, lds_w, y, lane); break;
    case 2: if constexpr (kM > 2) stem_tile<2>(bev, row_cell, start, n, tile, mt0, kM * 32, lds_w, y, lane); break;
    case 3: if constexpr (kM > 3) stem_tile<3>(bev, row_cell, start, n, tile, mt0, kM * 32, lds_w, y, lane); break;
    case 4: if constexpr (kM > 4) stem_tile<4>(bev, row_cell, start, n, tile, mt0, kM * 32, lds_w, y, lane); break;
    default: break;
  }
}

template <int kM>
__device__ __forceinline__ void stem_units(const float* __restrict__ bev, const int32_t* __restrict__ row_cell, int start, int n,
                                           int lo, int hi, const float* lds_w, float* __restrict__ y, int lane) {
  int u = lo;
  if (u < hi && u % kM != 0) {                             // head: the rest of a tile another wave started
    const int end = (u / kM + 1) * kM < hi ? (u / kM + 1) * kM : hi;
    stem_part<kM>(bev, row_cell, start, n, u / kM, u % kM, end - u, lds_w, y, lane);
    u = end;
  }
  for (; u + kM <= hi; u += kM) stem_tile<kM>(bev, row_cell, start, n, u / kM, 0, kM * 32, lds_w, y, lane);
  if (u < hi) stem_part<kM>(bev, row_cell, start, n, u / kM, 0, hi - u, lds_w, y, lane);     // tail: the start of the next tile
}

// ONE launch for the four parity classes.  Work unit = (32-cell tile, 32 output channels) = 96 MFMAs; the units of the four
// classes form one list (class 0 first) that is cut into equal contiguous ranges per block and, inside a block, per wave --
// from the DEVICE-side row counts (meta), so the split follows the frame's real occupancy and every wave gets the same number
// of units +- 1.  (Until r04 the blocks were split 2 : 3 : 3 : 5 between the classes and a wave took whole tiles: with 2.2
// tiles of 5 output blocks per wave in the largest class, the slowest SIMD carried 30 units against a mean of 21.5.)  A block
// keeps the weights of the class it works on in LDS; the few blocks whose range crosses a class boundary reload them once.
struct StemGemmArgs {
  const float* bev;
  const int32_t* row_cell;
  const int32_t* meta;
  const float* wprep[4];
  float* y[4];
};

__global__ __launch_bounds__(kStemBlock) void stem_gemm(StemGemmArgs a) {
  extern __shared__ float lds_w[];
  constexpr int kMs[4] = {2, 3, 3, 5};
  int start[4], n[4], units[4];
  int64_t total = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    start[c] = a.meta[4 + c];
    n[c] = a.meta[8 + c] - start[c];
    units[c] = ((n[c] + 31) / 32) * kMs[c];
    total += units[c];
  }
  const int64_t b_lo = total * blockIdx.x / gridDim.x, b_hi = total * (blockIdx.x + 1) / gridDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int kWaves = kStemBlock / 64;
  int64_t base = 0;
  bool loaded = false;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t lo = b_lo > base ? b_lo : base, hi = b_hi < base + units[c] ? b_hi : base + units[c];
    if (lo < hi) {                                         // block-uniform
      if (loaded) __syncthreads();                         // every wave is done with the previous class's weights
      for (int i = threadIdx.x; i < kMs[c] * kStemSteps * 64; i += kStemBlock) lds_w[i] = a.wprep[c][i];   // [kM][96][64]
      __syncthreads();
      loaded = true;
      const int l = (int)(lo - base), len = (int)(hi - lo);
      const int w_lo = l + (int)((int64_t)len * wave / kWaves), w_hi = l + (int)((int64_t)len * (wave + 1) / kWaves);
      if (c == 0) stem_units<2>(a.bev, a.row_cell, start[c], n[c], w_lo, w_hi, lds_w, a.y[c], lane);
      else if (c == 3) stem_units<5>(a.bev, a.row_cell, start[c], n[c], w_lo, w_hi, lds_w, a.y[c], lane);
      else stem_units<3>(a.bev, a.row_cell, start[c], n[c], w_lo, w_hi, lds_w, a.y[c], lane);
    }
    base += units[c];
  }
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

// lane = 4 channels (C = 32: eight lanes per output pixel, eight pixels per wave instruction; round 4 -- it was one lane per
// channel, two pixels per instruction: 4 x the instructions for the same bytes).
// out[b, ho, wo, c] = relu( sum_{ky,kx} Y[cell(2ho-1+ky, 2wo-1+kx)][slot(ky,kx)][c]
//                           + max_{valid window cells}( occupied ? Y[cell][q slot][c] : 0 ) + bias[c] )
// The sums run in the same fixed tap order per channel as before: the same bits.
template <int kC>
__global__ __launch_bounds__(kBlock) void stem_epilogue(StemEpiArgs a) {
  constexpr int kL = kC / 4;
  const int c = 4 * (threadIdx.x % kL);
  const int n_out = a.B * a.Ho * a.Wo;            // < 2^31 (host check): 32-bit index arithmetic throughout
  const float4 bias = *reinterpret_cast<const float4*>(a.bias + c);
  const int start[4] = {a.meta[4], a.meta[5], a.meta[6], a.meta[7]};
  // the nine row ids come through a raw buffer descriptor: a tap outside the grid uses an offset past the end (reads 0)
  // instead of sitting under a lane-dependent branch -- nine conditional loads were nine exec-masked branches, each followed
  // by a full wait
  const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.row_of), 0, a.B * a.H * a.W * 4, 0x00020000);
  for (int o = (int)blockIdx.x * (kBlock / kL) + (int)threadIdx.x / kL; o < n_out; o += (int)gridDim.x * (kBlock / kL)) {
    const int wo = o % a.Wo;
    const int t = o / a.Wo;
    const int ho = t % a.Ho, b = t / a.Ho;
    // all nine row ids first, then all eighteen loads (an empty or out-of-range tap reads row 0 of its class and is
    // discarded by the selects), then the sums in fixed tap order
    int rid[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int y = 2 * ho - 1 + t9 / 3, x = 2 * wo - 1 + t9 % 3;
      const bool inb = (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
      const int r = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrd, inb ? (unsigned)((b * a.H + y) * a.W + x) * 4u : 0x80000000u, 0, 0);
      rid[t9] = inb ? r : -2;     // -1: empty cell, -2: outside the grid
    }
    float4 va[9], vq[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int ky = t9 / 3, kx = t9 % 3;
      // a cell reached through an even ky has odd y, i.e. belongs to a class with two kernel rows (slots ky / 2)
      const int ey = (ky & 1) ^ 1, ex = (kx & 1) ^ 1;
      const int cls = ey * 2 + ex, taps = (1 + ey) * (1 + ex);
      const int slot = (ey ? ky >> 1 : 0) * (1 + ex) + (ex ? kx >> 1 : 0);
      const float* row = a.y[cls] + (int64_t)max(rid[t9] - start[cls], 0) * ((taps + 1) * kC);
      va[t9] = *reinterpret_cast<const float4*>(row + slot * kC + c);
      vq[t9] = *reinterpret_cast<const float4*>(row + taps * kC + c);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), qmax = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const bool occ = rid[t9] >= 0, empty = rid[t9] == -1;
      acc.x = occ ? acc.x + va[t9].x : acc.x;
      acc.y = occ ? acc.y + va[t9].y : acc.y;
      acc.z = occ ? acc.z + va[t9].z : acc.z;
      acc.w = occ ? acc.w + va[t9].w : acc.w;
      // an empty cell inside the grid contributes 0 to the pooled branch; outside the grid nothing (padding of the pool)
      qmax.x = occ ? fmaxf(qmax.x, vq[t9].x) : (empty ? fmaxf(qmax.x, 0.0f) : qmax.x);
      qmax.y = occ ? fmaxf(qmax.y, vq[t9].y) : (empty ? fmaxf(qmax.y, 0.0f) : qmax.y);
      qmax.z = occ ? fmaxf(qmax.z, vq[t9].z) : (empty ? fmaxf(qmax.z, 0.0f) : qmax.z);
      qmax.w = occ ? fmaxf(qmax.w, vq[t9].w) : (empty ? fmaxf(qmax.w, 0.0f) : qmax.w);
    }
    float4 r;
    r.x = fmaxf((acc.x + qmax.x) + bias.x, 0.0f);
    r.y = fmaxf((acc.y + qmax.y) + bias.y, 0.0f);
    r.z = fmaxf((acc.z + qmax.z) + bias.z, 0.0f);
    r.w = fmaxf((acc.w + qmax.w) + bias.w, 0.0f);
    *reinterpret_cast<float4*>(a.out + (int64_t)o * a.op + c) = r;
  }
}

}  // namespace smos

using namespace smos;

extern "C" int64_t smos_stem_scan_state_words(int64_t cells) {
  if (cells <= 0 || cells >= (1LL << 31)) return -1;
  const int64_t tiles = (cells + kScanTile - 1) / kScanTile;
  return tiles <= kScanMaxTiles ? 1 + tiles : -1;
}

extern "C" int smos_stem_mark(const float* coord, int32_t K, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W, int32_t* flags,
                              uint64_t* scan_state, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && T > 0 && N >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && K >= 2 && B * H * W < (1LL << 31),
               "stem_mark: bad sizes (H and W must be even)");
  SMOS_REQUIRE(flags && scan_state && (N == 0 || coord) && smos_stem_scan_state_words(B * H * W) > 0, "stem_mark: null device pointer / grid too large");
  const int n_tiles = (int)(smos_stem_scan_state_words(B * H * W) - 1);
  hipLaunchKernelGGL(stem_mark, dim3(grid_for(B * T * N > 0 ? B * T * N : 1)), dim3(kBlock), 0, (hipStream_t)stream, coord, (int)K, (int)B,
                     (int)T, N, (int)H, (int)W, flags, reinterpret_cast<unsigned long long*>(scan_state), n_tiles);
  return check_launch("stem_mark");
}

extern "C" int smos_stem_scan(int32_t* flags, int64_t B, int64_t H, int64_t W, uint64_t* scan_state, int32_t* row_cell,
                              int32_t* row_of, int32_t* meta, float* rows, int64_t row_floats, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && smos_stem_scan_state_words(B * H * W) > 0, "stem_scan: bad sizes");
  SMOS_REQUIRE(flags && scan_state && row_cell && row_of && meta && (reinterpret_cast<uintptr_t>(flags) & 15) == 0,
               "stem_scan: null / unaligned device pointer");
  SMOS_REQUIRE(!rows || (row_floats > 0 && row_floats % 4 == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0),
               "stem_scan: bad row table");
  const int n_tiles = (int)(smos_stem_scan_state_words(B * H * W) - 1);
  // every tile waits only for tiles with a smaller ticket, i.e. for blocks that are already running
  hipLaunchKernelGGL(stem_scan, dim3(n_tiles), dim3(kBlock), 0, (hipStream_t)stream, flags, (int)B, (int)H, (int)W,
                     reinterpret_cast<unsigned long long*>(scan_state), row_cell, row_of, meta, reinterpret_cast<float4*>(rows),
                     (int)(row_floats / 4));
  return check_launch("stem_scan");
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
  // one block per CU (120 KB of LDS); the kernel splits the work from the device-side row counts
  const int at = cus < 1 ? 1 : cus;
  hipLaunchKernelGGL(stem_gemm, dim3(at), dim3(kStemBlock), lds, (hipStream_t)stream, a);
  return check_launch("stem_gemm");
}

extern "C" int smos_stem_epilogue(const float* const* y4, const int32_t* meta, const int32_t* row_of, const float* bias,
                                  float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t C, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 1 && W > 1 && C == 32 && out_pitch >= C && out_pitch % 4 == 0, "stem_epilogue: bad sizes (C must be 32, pitch a multiple of 4)");
  SMOS_REQUIRE(y4 && meta && row_of && bias && out, "stem_epilogue: null pointer");
  SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(y4[0]) |
                 reinterpret_cast<uintptr_t>(y4[1]) | reinterpret_cast<uintptr_t>(y4[2]) | reinterpret_cast<uintptr_t>(y4[3])) & 15) == 0,
               "stem_epilogue: pointers must be 16-byte aligned");
  SMOS_REQUIRE(B * H * W < (1LL << 29), "stem_epilogue: grid too large for 32-bit buffer offsets");
  StemEpiArgs a;
  for (int k = 0; k < 4; ++k) a.y[k] = y4[k];
  a.meta = meta; a.row_of = row_of; a.bias = bias; a.out = out; a.op = out_pitch;
  a.B = (int)B; a.H = (int)H; a.W = (int)W;
  a.Ho = (int)((H + 2 - 3) / 2 + 1);
  a.Wo = (int)((W + 2 - 3) / 2 + 1);
  const int64_t n_out = B * a.Ho * a.Wo;
  hipLaunchKernelGGL((stem_epilogue<32>), dim3(grid_for(n_out * 8, kBlock, 256 * 32)), dim3(kBlock), 0, (hipStream_t)stream, a);
  return check_launch("stem_epilogue");
}
