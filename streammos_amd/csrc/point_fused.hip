// Fused point-side kernels of the inference engine (gfx950).
//
//   pointnet_scatter : per-point MLP 7 -> 64 -> 64 (BatchNorm folded, ReLU) on the matrix cores (exact-f32 MFMA, both
//                      layers chained in registers) and max-scattered straight into the channels-last BEV grid
//                      -- the (BS*T, 64, N) feature tensor of the reference (491 MB written by point_pre and
//                      read back by VoxelMaxPool, models/StreamMOS.py:101-102) never exists.
//   gather_scatter   : grid -> point bilinear gather fused with the point -> grid max scatter of the other
//                      view (B2P -> P2R and R2P -> P2B, networks/multi_view_encoder.py:395-404,410-419);
//                      optionally also emits the gathered point features (x1_point / the final
//                      bev_grid2point output) in point-major rows.
//   nhwc_to_nchw     : tiled transpose of a channels-last scatter target into a (strided) NCHW slice.
//
// Common structure ("LDS-staged accumulation"): phase 1 runs with lane = point, which is what the
// per-point arithmetic and the coalesced reads of channel-major inputs want; the 64-point x 32-channel
// result tile is staged in LDS (row pitch 36 floats: conflict-free for the 128-bit writes and the 32-bit
// row reads); phase 2 runs with lane = channel, so every wave instruction is two contiguous 128-byte rows
// (two points) -- the shape at which MI355X's memory-side atomics run at full rate -- instead of 64
// scattered 4-byte atomics.  Values <= 0 are skipped against the zero-filled grid exactly as in
// voxel_maxpool.hip (all of these features are post-ReLU or convex blends of post-ReLU maps).
#include "smos_common.h"

namespace smos {

constexpr int kTileP = 64;   // points per wave tile
constexpr int kTileC = 32;   // channels per round
constexpr int kPitch = 36;   // floats per LDS row

__device__ __forceinline__ int cell_2d(const float* __restrict__ row, float sy, float sx, int H, int W) {
  // same rule as voxel_maxpool.hip::cell_offset (reference: point_deep_cuda_kernel.cu:39-47)
  const float py = __fmul_rn(row[0], sy), px = __fmul_rn(row[1], sx);
  const bool ok = (py > -1.0f) && (py < (float)H) && (px > -1.0f) && (px < (float)W);
  return ok ? (int)py * W + (int)px : -1;
}

// phase 2: lane = channel; lanes 0-31 walk points 0..31 of the tile, lanes 32-63 points 32..63.
// LiDAR points arrive ring by ring in azimuth order, so consecutive points very often fall into the same
// cell (near ground rings put tens of points into one BEV cell): each half keeps a running maximum while
// the cell does not change and issues ONE row atomic per run instead of one per point.
__device__ __forceinline__ void rows_phase(const float* __restrict__ tile, const int* __restrict__ cells, int lane,
                                           float* __restrict__ grid_base, int64_t cell_pitch, int ch0,
                                           float* __restrict__ pts_base, int64_t po_n, int n_valid) {
  const int c = lane & 31, p0 = (lane >> 5) * (kTileP / 2);
  int cur = -1;
  float best = 0.0f;
#pragma unroll 4
  for (int i = 0; i < kTileP / 2; ++i) {
    const int p = p0 + i;
    const float v = tile[p * kPitch + c];
    if (pts_base && p < n_valid) pts_base[(int64_t)p * po_n + ch0 + c] = v;
    if (!grid_base) continue;
    const int cell = cells[p];
    if (cell != cur) {
      if (cur >= 0 && best > 0.0f)
        atomicMax(reinterpret_cast<int*>(grid_base + (int64_t)cur * cell_pitch + ch0 + c), __float_as_int(best));
      cur = cell;
      best = 0.0f;
    }
    best = fmaxf(best, v);
  }
  if (grid_base && cur >= 0 && best > 0.0f)
    atomicMax(reinterpret_cast<int*>(grid_base + (int64_t)cur * cell_pitch + ch0 + c), __float_as_int(best));
}

// ---------------------------------------------------------------------------------------------
// PointNet(7 -> 64 -> 64) + scatter
// ---------------------------------------------------------------------------------------------
struct PnsArgs {
  const float* xyzi;   // [S, 7, N]
  const float* coord;  // [S, N, K]
  const float* w1;     // [64, 7]  (BN folded)
  const float* b1;     // [64]
  const float* w2;     // [64, 64]
  const float* b2;     // [64]
  float* bev;          // [B, H, W, T*64] zero-filled; or, with row_of, compact rows [n_rows, T*64]
  const int32_t* row_of;  // null, or [B, H, W] row of every occupied cell (csrc/stem.hip): scatter into compact rows
  float* pts_out;      // [B, N, *] rows of the t == 0 sample (row pitch po_n), or null
  const int32_t* n_live;  // device, or null: points [*n_live, N) of the t == 0 scans are the padding tail: their rows are not wanted
  int64_t bev_sb, po_b, po_n;
  int S, T, N, K, H, W, tiles_per_sample;
};

// Matrix-core formulation.  Both layers are computed TRANSPOSED, C = W * X with the output channel on the MFMA row
// and the point on the MFMA column (v_mfma_f32_32x32x2_f32: exact f32, the fp32 vector rate):
//   * the weights are the A operand: one VGPR per (32-channel block, k-step), 8 + 64 registers loaded ONCE per wave and
//     kept for the whole persistent loop -- no per-tile weight traffic at all (the scalar-load version re-streams 18 KB
//     per tile through a 16 KB scalar cache that it thrashes; measured: 0.40 ms, 0.26 ms with half the weights);
//   * the points are the B operand: lane l holds feature 2s + (l >> 5) of point (l & 31), read straight from the
//     channel-major input; feature 7 is the constant 1 that carries the folded layer-1 bias;
//   * a 32x32 result tile has its point on the lane and its channels in the 16 accumulator registers, which IS the B
//     operand layout of the next layer: register r of lane half h is channel 8(r>>2) + 4h + (r&3), so layer 2 walks the
//     hidden channels in that order (its A operand is loaded with the same permutation) and takes relu(C1) with no lane
//     movement and no LDS.
// The 32-point x 64-channel output tile goes through LDS once (point-major rows, pitch 68) and leaves with lane =
// channel: every store / atomic is one contiguous 256-byte row.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kNt = 32;        // points per MFMA column tile
constexpr int kPitchM = 68;    // floats per LDS row (64 channels + 4: conflict-free 128-bit row writes)

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kBlock) void pointnet_scatter(PnsArgs a) {
  __shared__ float lds_tile[kBlock / kWave][kNt * kPitchM];
  __shared__ float lds_b2[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 31, hh = lane >> 5;
  float w1a[2][4], w2a[2][32];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = 2 * s + hh;
      w1a[mt][s] = k < 7 ? a.w1[(mt * 32 + col) * 7 + k] : a.b1[mt * 32 + col];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int r = s & 15;
      const int k = (s >> 4) * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
      w2a[mt][s] = a.w2[(mt * 32 + col) * 64 + k];
    }
  }
  if (threadIdx.x < 64) lds_b2[threadIdx.x] = a.b2[threadIdx.x];
  __syncthreads();

  // tile bookkeeping is wave-uniform 32-bit arithmetic (the host checks that the tile count fits): it stays on the
  // scalar unit, which matters because VALU instructions issue through the same port as the MFMAs
  const int nt_per_sample = (a.N + kNt - 1) / kNt;
  const int n_nt = a.S * nt_per_sample;
  float* tile = lds_tile[wave];
  const int nt_step = (int)gridDim.x * (kBlock / kWave);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  // inputs of one column tile: 4 features per lane + the two grid coordinates of the lane's point
  struct TileIn {
    float x[4], cy, cx;      // as loaded: the selects that turn them into operands are applied where they are consumed --
    bool has;                // applied here, hipcc hoists them right behind the loads and the prefetch is waited for at once
  };
  // Every load of the software pipeline is an UNCONDITIONAL raw buffer load: a lane without a point uses an offset past
  // the end of the buffer and reads 0.  Written as "has ? ptr[i] : 0" the loads sit under lane-dependent branches, hipcc's
  // wait-count pass loses track of them and puts s_waitcnt vmcnt(0) in front of the next vector-memory-dependent
  // instruction -- here right behind the prefetch, so every tile sat through a full memory latency before its matrix
  // phase (ISA: vmcnt(0) 30 instructions after the loads; MFMA busy 36 %).
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xyzi), 0, a.S * 7 * a.N * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t csrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.coord), 0, a.S * a.N * a.K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int32_t*>(a.row_of), 0, a.row_of ? (a.S / a.T) * a.H * a.W * 4 : 0, 0x00020000);
  constexpr unsigned kOob = 0x80000000u;
  auto fetch = [&](int nt, TileIn& in) {
    // tile nt = column tile nt / S of sample nt % S (sample fastest): a wave's tiles nt, nt + step, .. then spread evenly over
    // the point range of every sample, so every wave meets the same share of the scans' padding tails (sample-major order
    // left a wave 15 to 20 live tiles of its 29)
    const bool live = nt < n_nt;
    const int s = live ? nt % a.S : 0;
    const int n0 = live ? (nt / a.S) * kNt : 0;
    const bool has = live & (n0 + col < a.N);
    const unsigned xoff = (unsigned)((s * 7 + hh) * a.N + n0 + col) * 4u;      // feature f = 2 q + hh of the lane's point
    const unsigned fstep = (unsigned)a.N * 8u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = has & (2 * q + hh < 7);
      in.x[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xsrd, ok ? xoff + fstep * q : kOob, 0, 0));
    }
    const unsigned coff = has ? (unsigned)((s * a.N + n0 + col) * a.K) * 4u : kOob;
    in.cy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(csrd, coff, 0, 0));
    in.cx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(csrd, coff + 4u, 0, 0));      // (the last argument is the cache policy)
    in.has = has;
  };

  // target slot of the lane's point: its cell of the dense grid, or (compact target) the row of that cell -- looked up
  // here, per lane and inside the prefetch shadow, not in the serial flush loop.  Equal cells <=> equal rows, so the
  // run detection works on either.
  // split in two: the request (cell from the coordinates + the row lookup issued) and the resolve (first use of the
  // looked-up row), so that a whole matrix phase lies between them
  auto cell_request = [&](const TileIn& in, int tile, int& c, int& r) {
    const float yx[2] = {in.has ? in.cy : -1.0f, in.has ? in.cx : -1.0f};   // -1 is outside the half-open test of cell_2d
    c = cell_2d(yx, 1.0f, 1.0f, a.H, a.W);
    const unsigned roff = (c >= 0) ? (unsigned)(((tile % a.S) / a.T) * a.H * a.W + c) * 4u : kOob;
    r = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrd, roff, 0, 0);      // no row table: zero-sized buffer, reads 0
  };
  auto cell_resolve = [&](int c, int r) { return ((a.row_of != nullptr) & (c >= 0)) ? r : c; };
  const int n_live = a.n_live ? min(max(*a.n_live, 0), a.N) : a.N;
  int nt = (int)blockIdx.x * (kBlock / kWave) + wave_u;
  TileIn cur, nxt;
  fetch(nt, cur);
  fetch(nt + nt_step, nxt);
  int cell;
  {
    int c0, r0;
    cell_request(cur, nt, c0, r0);
    cell = cell_resolve(c0, r0);
    // everything the loop carries has landed before the first trip: otherwise the loop header inherits pending loads from
    // this path, and the wait the compiler puts there is, on the back edge, a wait for the previous tile's atomics
    asm volatile("" : "+v"(cur.x[0]), "+v"(cur.x[1]), "+v"(cur.x[2]), "+v"(cur.x[3]), "+v"(nxt.x[0]), "+v"(nxt.x[1]), "+v"(nxt.x[2]),
                 "+v"(nxt.x[3]), "+v"(nxt.cy), "+v"(nxt.cx), "+v"(cell));
  }
  for (; nt < n_nt; nt += nt_step) {
    // Software pipeline, two tiles deep: the inputs of tile + 2 and the row lookup of tile + 1 (whose coordinates arrived
    // during the previous matrix phase) are requested here and waited for after this tile's matrix phase, BEFORE this
    // tile's stores and atomics are issued.  The vector-memory counter of gfx9 retires in order and the number of atomics
    // is data dependent, so a wait placed after them is a wait for all of them (vmcnt(0)) -- with a one-deep pipeline the
    // looked-up row of the next tile was exactly such a wait at the loop's back edge.
    TileIn nn;
    fetch(nt + 2 * nt_step, nn);
    int c_next, r_next;
    cell_request(nxt, nt + nt_step, c_next, r_next);
    const int s = nt % a.S;
    const int n0 = (nt / a.S) * kNt;
    const int b = s / a.T, t = s - b * a.T;
    float xk[4] = {cur.x[0], cur.x[1], cur.x[2], hh ? 1.0f : cur.x[3]};      // feature 7 = 1 carries the folded layer-1 bias
    // the shuffle is its own statement: inside the short-circuit expression it would run with lane 31 masked off, and
    // a lane that reads a masked-off lane gets 0 -- indistinguishable from cell 0
    const int cell_after = __shfl_down(cell, 1);
    const uint32_t tails = (uint32_t)__ballot(hh == 0 && (col == kNt - 1 || cell != cell_after));
    // a tile none of whose points falls into the grid (the padding tail of a scan: 25-40 % of the rows) produces nothing
    // unless its point rows are wanted (t == 0): skip the matrix work, keep the software pipeline moving
    if (!(a.pts_out && t == 0 && n0 < n_live) && __ballot(cell >= 0) == 0) {
      // (waited for on this path too: the register copies of the rotation sit in the shared loop latch, and a wait placed
      // there would run on the main path as well -- as a wait for its atomics)
      asm volatile("" : "+v"(nn.x[0]), "+v"(nn.x[1]), "+v"(nn.x[2]), "+v"(nn.x[3]), "+v"(nn.cy), "+v"(nn.cx), "+v"(r_next));
      cell = cell_resolve(c_next, r_next);
      cur = nxt;
      nxt = nn;
      continue;
    }

    f32x16 c1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) c1[mt][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) c1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1a[mt][q], xk[q], c1[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) c1[mt][r] = fmaxf(c1[mt][r], 0.0f);

    f32x16 c2[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bias = *reinterpret_cast<const float4*>(&lds_b2[mt * 32 + 8 * g + 4 * hh]);
        c2[mt][4 * g + 0] = bias.x; c2[mt][4 * g + 1] = bias.y; c2[mt][4 * g + 2] = bias.z; c2[mt][4 * g + 3] = bias.w;
      }
#pragma unroll
    for (int q = 0; q < 32; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        c2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[mt][q], c1[q >> 4][q & 15], c2[mt], 0, 0, 0);

    float* row = tile + col * kPitchM + 4 * hh;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(row + mt * 32 + 8 * g) =
            make_float4(fmaxf(c2[mt][4 * g], 0.0f), fmaxf(c2[mt][4 * g + 1], 0.0f), fmaxf(c2[mt][4 * g + 2], 0.0f),
                        fmaxf(c2[mt][4 * g + 3], 0.0f));
    wave_sync();

    // lane = channel: 32 points, each one 256-byte row.  All 32 LDS reads are issued before the first use, the point
    // rows leave first, then same-cell runs are reduced in registers and flushed with one row atomic per run; the run
    // ends are a wave-uniform bit mask, so the control flow is scalar.
    const int n_valid = min(kNt, a.N - n0);
    const unsigned ulane = lane;   // zero-extended lane offset: lets the stores use the scalar-base addressing form
    float v[kNt];
#pragma unroll
    for (int i = 0; i < kNt; ++i) v[i] = tile[i * kPitchM + lane];
    // the prefetch is waited for here (tied to the last LDS read so that the scheduler cannot hoist the wait)
    asm volatile("" : "+v"(nn.x[0]), "+v"(nn.x[1]), "+v"(nn.x[2]), "+v"(nn.x[3]), "+v"(nn.cy), "+v"(nn.cx), "+v"(r_next), "+v"(v[kNt - 1]));
    const int cell_next = cell_resolve(c_next, r_next);
    if (a.pts_out && t == 0) {
      float* prow = a.pts_out + (int64_t)b * a.po_b + (int64_t)n0 * a.po_n;   // wave-uniform
      if (n_valid == kNt) {
#pragma unroll
        for (int i = 0; i < kNt; ++i) (prow + (int64_t)i * a.po_n)[ulane] = v[i];
      } else {
#pragma unroll
        for (int i = 0; i < kNt; ++i)
          if (i < n_valid) (prow + (int64_t)i * a.po_n)[ulane] = v[i];
      }
    }
    // wave-uniform target of this (sample, scan): the sample's slab of the dense grid, or the compact row table
    float* gsample = a.bev + (a.row_of ? (int64_t)0 : (int64_t)b * a.bev_sb) + t * 64;
    const int64_t cell_pitch = (int64_t)a.T * 64;
    float best = 0.0f;
#pragma unroll
    for (int i = 0; i < kNt; ++i) {
      best = fmaxf(best, v[i]);
      if ((tails >> i) & 1) {
        const int ci = __builtin_amdgcn_readlane(cell, i);
        if (ci >= 0 && best > 0.0f)
          atomicMax(reinterpret_cast<int*>(gsample + (int64_t)ci * cell_pitch) + ulane, __float_as_int(best));
        best = 0.0f;
      }
    }
    wave_sync();
    cur = nxt;
    nxt = nn;
    cell = cell_next;
  }
}

// ---------------------------------------------------------------------------------------------
// bilinear gather (NCHW or any-stride grid) -> [optional point rows] -> [optional max scatter, channels-last]
// ---------------------------------------------------------------------------------------------
struct GsArgs {
  const float* grid;    // [B, C, Hg, Wg] element strides gs_*
  const float* gcoord;  // [B, N, Kg]  gather coordinates
  const float* scoord;  // [B, N, Ks]  scatter coordinates (or null)
  float* out;           // [B, Ho, Wo, C] zero-filled channels-last scatter target (or null)
  float* pts_out;       // [B, N, *] point rows (row pitch po_n, channel offset applied by the host) or null
  int64_t gs_b, gs_c, gs_h, gs_w, out_sb, po_b, po_n;
  int B, C, N, Kg, Ks, Hg, Wg, Ho, Wo, tiles_per_sample;
  float gsy, gsx, ssy, ssx;
};

__device__ __forceinline__ float pix(float c, float s, int size) {
  // networks/backbone.py:467-468 + ATen grid_sampler_unnormalize(align_corners=True); see bilinear_gather.hip
  const float sm1 = (float)(size - 1);
  const float gn = __fsub_rn(__fdiv_rn(__fmul_rn(__fmul_rn(2.0f, c), s), sm1), 1.0f);
  return __fmul_rn(__fdiv_rn(__fadd_rn(gn, 1.0f), 2.0f), sm1);
}

__global__ __launch_bounds__(kBlock) void gather_scatter(GsArgs a) {
  __shared__ float lds_tile[kBlock / kWave][kTileP * kPitch];
  __shared__ int lds_cell[kBlock / kWave][kTileP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n_tiles = (int64_t)a.B * a.tiles_per_sample;
  for (int64_t tile0 = (int64_t)blockIdx.x * (kBlock / kWave); tile0 < n_tiles; tile0 += (int64_t)gridDim.x * (kBlock / kWave)) {
    const int64_t tile = tile0 + wave;
    const bool live = tile < n_tiles;
    const int b = live ? (int)(tile / a.tiles_per_sample) : 0;
    const int n0 = live ? (int)(tile - (int64_t)b * a.tiles_per_sample) * kTileP : 0;
    const int n = n0 + lane;
    const bool has = live && n < a.N;

    int64_t off[4] = {-1, -1, -1, -1};
    float wt[4] = {0.f, 0.f, 0.f, 0.f};
    if (has) {
      const float* cr = a.gcoord + ((int64_t)b * a.N + n) * a.Kg;
      const float iy = pix(cr[0], a.gsy, a.Hg), ix = pix(cr[1], a.gsx, a.Wg);
      const float fy = floorf(iy), fx = floorf(ix);
      const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
      const bool fin = (iy > -2.0f) && (iy < (float)(a.Hg + 1)) && (ix > -2.0f) && (ix < (float)(a.Wg + 1));
      const int y0 = fin ? (int)fy : -5, x0 = fin ? (int)fx : -5;
      const float w4[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int y = y0 + (k >> 1), xx = x0 + (k & 1);
        const bool in = (y >= 0) && (y < a.Hg) && (xx >= 0) && (xx < a.Wg);
        off[k] = in ? (int64_t)y * a.gs_h + (int64_t)xx * a.gs_w : (int64_t)-1;
        wt[k] = in ? w4[k] : 0.0f;
      }
    }
    lds_cell[wave][lane] = (has && a.scoord) ? cell_2d(a.scoord + ((int64_t)b * a.N + n) * a.Ks, a.ssy, a.ssx, a.Ho, a.Wo) : -1;
    const float* gb = a.grid + (int64_t)b * a.gs_b;
    const int n_valid = live ? min(kTileP, a.N - n0) : 0;
    float* grid_base = a.out ? a.out + (int64_t)b * a.out_sb : nullptr;
    float* pts_base = (a.pts_out && live) ? a.pts_out + (int64_t)b * a.po_b + (int64_t)n0 * a.po_n : nullptr;
    for (int c0 = 0; c0 < a.C; c0 += kTileC) {
      float* row = &lds_tile[wave][lane * kPitch];
#pragma unroll 2
      for (int j4 = 0; j4 < kTileC; j4 += 4) {
        float y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* gc = gb + (int64_t)(c0 + j4 + u) * a.gs_c;
          float acc = 0.0f;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (off[k] >= 0) acc += gc[off[k]] * wt[k];
          y[u] = acc;
        }
        *reinterpret_cast<float4*>(row + j4) = make_float4(y[0], y[1], y[2], y[3]);
      }
      __syncthreads();
      if (live) rows_phase(lds_tile[wave], lds_cell[wave], lane, grid_base, (int64_t)a.C, c0, pts_base, a.po_n, n_valid);
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// channels-last [B, HW, C] -> NCHW slice (batch stride ds_b, channel stride ds_c, contiguous planes)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void nhwc_to_nchw(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                       int64_t HW, int64_t ds_b, int64_t ds_c) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const int64_t p0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* sp = src + (int64_t)b * HW * C;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int64_t p = p0 + r;
    const int c = c0 + tx;
    t[r][tx] = (p < HW && c < C) ? sp[p * C + c] : 0.0f;
  }
  __syncthreads();
  float* dp = dst + (int64_t)b * ds_b;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const int64_t p = p0 + tx;
    if (p < HW && c < C) dp[(int64_t)c * ds_c + p] = t[tx][r];
  }
}

}  // namespace smos

using namespace smos;

static int pointnet_scatter_launch(const float* xyzi, const float* coord, int32_t K, const float* w1, const float* b1,
                                   const float* w2, const float* b2, float* bev, const int32_t* row_of, float* pts_out,
                                   const int32_t* n_live, int64_t po_b, int64_t po_n, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W, int32_t cin,
                                   int32_t cmid, int32_t cout, smos_stream_t stream) {
  if (cin != 7 || cmid != 64 || cout != 64) {
    set_error("pointnet_scatter: built for the 7 -> 64 -> 64 point MLP (got %d -> %d -> %d)", (int)cin, (int)cmid, (int)cout);
    return SMOS_ERR_UNSUPPORTED;
  }
  SMOS_REQUIRE(B > 0 && T > 0 && N > 0 && H > 0 && W > 0 && K >= 2, "pointnet_scatter: bad sizes");
  SMOS_REQUIRE(xyzi && coord && w1 && b1 && w2 && b2 && bev, "pointnet_scatter: null pointer");
  SMOS_REQUIRE(H * W < (1LL << 31) && (!pts_out || po_n >= 64), "pointnet_scatter: bad geometry");
  SMOS_REQUIRE(B * T * 7 * N * 4 < (1LL << 31) && B * T * N * K * 4 < (1LL << 31) && B * H * W * 4 < (1LL << 31),
               "pointnet_scatter: an input larger than 2 GiB (32-bit buffer offsets)");
  PnsArgs a;
  a.xyzi = xyzi; a.coord = coord; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.bev = bev; a.row_of = row_of; a.pts_out = pts_out;
  a.n_live = n_live;
  a.bev_sb = H * W * T * 64; a.po_b = po_b; a.po_n = po_n;
  a.S = (int)(B * T); a.T = (int)T; a.N = (int)N; a.K = K; a.H = (int)H; a.W = (int)W;
  a.tiles_per_sample = (int)((N + kNt - 1) / kNt);
  // persistent waves: exactly as many blocks as are resident at once
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&pointnet_scatter), 0, kBlock, &ks, "pointnet_scatter")) return rc;
  const int64_t resident = (int64_t)ks.per_cu * ks.cus;
  const int64_t n_nt = (int64_t)a.S * ((N + kNt - 1) / kNt);
  SMOS_REQUIRE(n_nt < (1LL << 30), "pointnet_scatter: too many points for 32-bit tile indices");
  const int64_t want = (n_nt + 3) / 4;
  hipLaunchKernelGGL(pointnet_scatter, dim3((unsigned)(want < resident ? want : resident)), dim3(kBlock), 0,
                     (hipStream_t)stream, a);
  return check_launch("pointnet_scatter");
}

extern "C" int smos_pointnet_scatter(const float* xyzi, const float* coord, int32_t K, const float* w1, const float* b1,
                                     const float* w2, const float* b2, float* bev, float* pts_out, int64_t po_b,
                                     int64_t po_n, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W, int32_t cin,
                                     int32_t cmid, int32_t cout, smos_stream_t stream) {
  return pointnet_scatter_launch(xyzi, coord, K, w1, b1, w2, b2, bev, nullptr, pts_out, nullptr, po_b, po_n, B, T, N, H, W, cin, cmid,
                                 cout, stream);
}

extern "C" int smos_pointnet_scatter_rows(const float* xyzi, const float* coord, int32_t K, const float* w1, const float* b1,
                                          const float* w2, const float* b2, float* rows, const int32_t* row_of, float* pts_out,
                                          int64_t po_b, int64_t po_n, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W,
                                          int32_t cin, int32_t cmid, int32_t cout, smos_stream_t stream) {
  SMOS_REQUIRE(row_of, "pointnet_scatter_rows: null row table");
  return pointnet_scatter_launch(xyzi, coord, K, w1, b1, w2, b2, rows, row_of, pts_out, nullptr, po_b, po_n, B, T, N, H, W, cin, cmid,
                                 cout, stream);
}

// n_live (device int32, may be null): the first *n_live points of the current (t == 0) scan of every sample are real; the point
// rows of its padding tail are not written (nothing reads them once the point head knows the same count).
extern "C" int smos_pointnet_scatter_rows_live(const float* xyzi, const float* coord, int32_t K, const float* w1, const float* b1,
                                               const float* w2, const float* b2, float* rows, const int32_t* row_of, float* pts_out,
                                               int64_t po_b, int64_t po_n, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W,
                                               int32_t cin, int32_t cmid, int32_t cout, const int32_t* n_live, smos_stream_t stream) {
  SMOS_REQUIRE(row_of, "pointnet_scatter_rows: null row table");
  return pointnet_scatter_launch(xyzi, coord, K, w1, b1, w2, b2, rows, row_of, pts_out, n_live, po_b, po_n, B, T, N, H, W, cin, cmid,
                                 cout, stream);
}

extern "C" int smos_gather_scatter(const float* grid, const int64_t* grid_stride, const float* gcoord, int32_t Kg,
                                   const float* gscale, const float* scoord, int32_t Ks, const float* sscale, float* out,
                                   float* pts_out, int64_t po_b, int64_t po_n, int64_t B, int64_t C, int64_t Hg, int64_t Wg,
                                   int64_t N, int64_t Ho, int64_t Wo, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && C > 0 && C % kTileC == 0 && N > 0 && Hg > 0 && Wg > 0 && Kg >= 2, "gather_scatter: bad sizes (C must be a multiple of 32)");
  SMOS_REQUIRE(grid && grid_stride && gcoord && gscale, "gather_scatter: null pointer");
  SMOS_REQUIRE(out || pts_out, "gather_scatter: nothing to produce");
  SMOS_REQUIRE(!out || (scoord && sscale && Ks >= 2 && Ho > 0 && Wo > 0 && Ho * Wo < (1LL << 31)), "gather_scatter: bad scatter target");
  SMOS_REQUIRE(!pts_out || po_n >= C, "gather_scatter: point row pitch smaller than C");
  GsArgs a;
  a.grid = grid; a.gcoord = gcoord; a.scoord = out ? scoord : nullptr; a.out = out; a.pts_out = pts_out;
  a.gs_b = grid_stride[0]; a.gs_c = grid_stride[1]; a.gs_h = grid_stride[2]; a.gs_w = grid_stride[3];
  a.out_sb = Ho * Wo * C; a.po_b = po_b; a.po_n = po_n;
  a.B = (int)B; a.C = (int)C; a.N = (int)N; a.Kg = Kg; a.Ks = Ks; a.Hg = (int)Hg; a.Wg = (int)Wg; a.Ho = (int)Ho; a.Wo = (int)Wo;
  a.tiles_per_sample = (int)((N + kTileP - 1) / kTileP);
  a.gsy = gscale[0]; a.gsx = gscale[1];
  a.ssy = out ? sscale[0] : 0.f; a.ssx = out ? sscale[1] : 0.f;
  const int64_t tiles = B * a.tiles_per_sample;
  const int64_t blocks = (tiles + 3) / 4;
  hipLaunchKernelGGL(gather_scatter, dim3((unsigned)(blocks < 256 * 32 ? blocks : 256 * 32)), dim3(kBlock), 0,
                     (hipStream_t)stream, a);
  return check_launch("gather_scatter");
}

extern "C" int smos_nhwc_to_nchw(const float* src, float* dst, int64_t ds_b, int64_t ds_c, int64_t B, int64_t C, int64_t HW,
                                 smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && C > 0 && HW > 0 && src && dst, "nhwc_to_nchw: bad arguments");
  SMOS_REQUIRE(B <= 65535 && (C + 31) / 32 <= 65535, "nhwc_to_nchw: too many batches / channels");
  dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(nhwc_to_nchw, grid, dim3(kBlock), 0, (hipStream_t)stream, src, dst, (int)C, HW, ds_b, ds_c);
  return check_launch("nhwc_to_nchw");
}
