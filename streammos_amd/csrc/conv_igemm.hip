// General channels-last fp32 convolution (KH x KW, stride 1 or 2, Cin and Cout multiples of 32) as an implicit GEMM on the
// matrix cores of gfx950, with the bias + activation (+ residual) epilogue fused -- every 2-D convolution of the
// network (networks/backbone.py:9-34,136-159; networks/multi_view_encoder.py:460-497) runs on this one kernel.
//
//     C[cout][pixel] = sum_{tap, cin} W[cout][cin][tap] * X[pixel * stride + tap - pad][cin]
//
// Transposed form, as in the other matrix-core kernels of this library: output channel on the MFMA row, 32 consecutive
// output pixels of one image row on the column, v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered fmaf chain).
//
//   work item  = 4 output rows x 32 columns x (32 * MT) output channels; one block (4 waves) per item, one row per wave;
//                blocks are persistent and own a contiguous range of items (cout tile fastest).
//   stage      = 32 input channels of one tap (16 k-steps): 16 * MT MFMAs per wave.  The K loop of a tile is a sequence
//                of KH * KW * Cin / 32 stages, and the stages of consecutive tiles form ONE stream: operand requests run
//                three stages ahead straight across tile boundaries, so a wave's pipeline never drains.
//   B operand  = activations, read directly from global memory (L1 / L2 absorb the tap re-reads) through a buffer
//                descriptor: lane (p, h) loads four float4 = channels 8 j + 4 h + (0..3) of the stage's 32; lanes outside
//                the image use an offset past the end and read zeros.  Four named register sets rotate.
//   A operand  = weights, STREAMED through a four-slot LDS ring (4 * MT KB per stage) instead of being resident: the
//                block's 256 threads request the slice of stage g + 3 in stage g (one float4 each per MT), keep it in
//                registers for two stages and store it to the ring in stage g + 2; one barrier per stage publishes it.
//                A lane reads its fragment as float4 = four consecutive k-steps (host-side operand order, ops.conv_prepare).
//                16 * MT KB of LDS per block: several blocks per CU coexist with the other HIP stream's kernels (the first
//                version of an own conv kept 144 KB of weights resident at C = 64 and was crowded out of the pipeline).
//   waits      = vmcnt retires in order, so a wait for one load is a wait for every older one: every operand is
//                requested >= 2 stages before its first use and nothing young is ever waited for -- in particular the
//                epilogue issues no load (bias lives in LDS, the residual tile is requested one stage early), otherwise it
//                would drain the whole prefetch once per tile (measured with in-kernel stamps: 19-23 % of a wave's time).
//   epilogue   = out = act(acc + bias [+ residual]) from the accumulators, 16-byte stores (lane (p, h), register r <->
//                channel 32 mt + 8 (r >> 2) + 4 h + (r & 3)); inputs, residual and output may be channel slices of
//                wider channels-last buffers (row pitches).
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"

namespace smos {

template <int MT, bool RES, bool SUMS = false>
__global__ __launch_bounds__(256, 2) void conv_igemm(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 ring[];     // 4 slots x 256 * MT float4, then Cout bias floats
  constexpr int kSlot = 256 * MT;
  float* bias_lds = reinterpret_cast<float*>(ring + 4 * kSlot);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  // a block owns a contiguous range of items (cout tile fastest, so the weight slices of its stages are one cyclic
  // sequence).  Blocks b, b + 8, b + 16 .. share an XCD and its L2 (dispatch is round-robin over the 8 XCDs): they get
  // neighbouring ranges, so that the rows one block reads as halo are the rows its neighbour reads as centre (speed only).
  const int per_block = (a.n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nb = (int)gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = (int)blockIdx.x & 7;
  const int lblock = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
  const int first = lblock * per_block;
  const int iters = a.n_items - first < per_block ? a.n_items - first : per_block;
  const int total = iters * a.nstage;
  if (total <= 0) return;

  // Bookkeeping is kept off the per-stage path: the scalar unit is shared by the CU's waves, and a hundred cursor
  // instructions per stage cost more wave time than the stage's 16 MFMAs (in-kernel stamps).  Per stage: three counters and
  // one running offset; per TILE (a rare, wave-uniform branch at the end of a stage): the divisions that locate it.
  auto tile_of = [&](int it) {
    ConvTile t;
    t.valid = it < iters;
    const int q = t.valid ? first + it : first;
    t.ct = q % a.nct;
    int u = q / a.nct;
    t.x0 = (u % a.xt) * 32;
    u /= a.xt;
    t.y = (u % a.hq) * 4 + wave;
    t.b = u / a.hq;
    t.valid = t.valid & (t.y < a.Ho);
    return t;
  };

  // activations / residual / output as raw buffers: a lane outside the image gets an offset past the end (loads return
  // zero = the convolution's padding, stores are dropped).  No load sits under a lane-dependent branch -- hipcc would stop
  // counting and wait vmcnt(0) at the next use.
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? a.r_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.o_bytes, 0x00020000);
  // The bias is requested here, as the OLDEST load of the kernel, and written to LDS only just before the prologue's barrier:
  // staged with a plain loop up front, every wave sat through one full memory latency before it issued its first operand
  // request (two serialised latencies per launch; the small layers run one item per block).  Past Cout the buffer returns 0.
  const __amdgpu_buffer_rsrc_t bsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? a.cout * 4 : 0, 0x00020000);
  float bias_r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias_r[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bsrd, (unsigned)(tid + 256 * k) * 4u, 0, 0));

  // ---- activation requests (three stages ahead): position inside the tile = counters + a running element offset ----
  const int xp = (int)a.xp;
  const int d_dx = xp - (a.nch - 1) * 32;                              // next tap column: back to channel chunk 0
  const int d_dy = (a.W - (a.KW - 1)) * xp - (a.nch - 1) * 32;         // next tap row: back to tap column 0
  int pb_it = 0, pb_left = a.nstage, pb_ch = 0, pb_dx = 0, pb_dy = 0, pb_delta = 0;
  int pb_y0, pb_xs, pb_base;      // first input row (scalar), first input column (per lane), element offset of (row, column, 4 h)
  bool pb_valid;
  auto locate_b = [&]() {
    const ConvTile t = tile_of(pb_it);
    pb_valid = t.valid;
    pb_y0 = t.y * a.S - a.PH;
    pb_xs = (t.x0 + p) * a.S - a.PW;
    pb_base = ((t.b * a.H + pb_y0) * a.W + pb_xs) * xp + 4 * h;
  };
  locate_b();
  auto load_b = [&](float4 (&bset)[4]) {
    // bitwise: a short-circuit && becomes a branch and splits the stage body
    const bool ok = pb_valid & ((unsigned)(pb_y0 + pb_dy) < (unsigned)a.H) & ((unsigned)(pb_xs + pb_dx) < (unsigned)a.W);
    const unsigned voff = ok ? (unsigned)(pb_base + pb_delta) * 4u : 0x80000000u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrd, voff + 32u * j, 0, 0);
      bset[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  unsigned pb_voff = 0x80000000u;
  auto addr_b = [&]() {
    const bool ok = pb_valid & ((unsigned)(pb_y0 + pb_dy) < (unsigned)a.H) & ((unsigned)(pb_xs + pb_dx) < (unsigned)a.W);
    pb_voff = ok ? (unsigned)(pb_base + pb_delta) * 4u : 0x80000000u;
  };
  auto load_b_half = [&](float4 (&bset)[4], int j0) {
#pragma unroll
    for (int j = j0; j < j0 + 2; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrd, pb_voff + 32u * j, 0, 0);
      bset[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto advance_b = [&]() {          // channel chunk fastest, then the tap column, then the tap row; selects only
    const bool ch_wrap = pb_ch + 1 == a.nch;
    const bool dx_wrap = ch_wrap & (pb_dx + 1 == a.KW);
    pb_delta += ch_wrap ? (dx_wrap ? d_dy : d_dx) : 32;
    pb_ch = ch_wrap ? 0 : pb_ch + 1;
    pb_dx = dx_wrap ? 0 : pb_dx + (ch_wrap ? 1 : 0);
    pb_dy += dx_wrap ? 1 : 0;
    --pb_left;
  };
  auto next_tile_b = [&]() {        // rare path
    pb_left = a.nstage;
    pb_ch = pb_dx = pb_dy = pb_delta = 0;
    ++pb_it;
    locate_b();
  };

  // ---- weight requests (three stages ahead): the slices of consecutive stages are consecutive, cyclically ----
  const int n_slices = a.nct * a.nstage;
  int pa_slice = (first % a.nct) * a.nstage, pa_g = 0;
  // weight slices in flight: two sets of named registers (arrays here end up in scratch memory for MT > 1)
  auto load_a = [&](float4& r0, float4& r1, float4& r2, float4& r3) {
    const float4* wsrc = a.w + (int64_t)(pa_g < total ? pa_slice : 0) * kSlot + tid;      // past the end: any valid slice
    r0 = wsrc[0];
    if constexpr (MT > 1) r1 = wsrc[256];
    if constexpr (MT > 2) {
      r2 = wsrc[512];
      r3 = wsrc[768];
    }
    ++pa_g;
    pa_slice = pa_slice + 1 == n_slices ? 0 : pa_slice + 1;
  };
  auto park = [&](int slot, const float4& r0, const float4& r1, const float4& r2, const float4& r3) {
    float4* dst = ring + slot * kSlot + tid;
    dst[0] = r0;
    if constexpr (MT > 1) dst[256] = r1;
    if constexpr (MT > 2) {
      dst[512] = r2;
      dst[768] = r3;
    }
  };
  auto read_a = [&](float4 (&af)[4][MT], int slot, int i4) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[i4][mt] = ring[slot * kSlot + (i4 * MT + mt) * 64 + lane];
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

  [[maybe_unused]] auto mfma_group = [&](const float4 (&af)[4][MT], const float4& bv, int i4) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i4][mt].x, bv.x, acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i4][mt].y, bv.y, acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i4][mt].z, bv.z, acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i4][mt].w, bv.w, acc[mt], 0, 0, 0);
  };

  auto mfma_half = [&](const float4 (&af)[4][MT], const float4& bv, int i4, bool lo) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo ? af[i4][mt].x : af[i4][mt].z, lo ? bv.x : bv.z, acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo ? af[i4][mt].y : af[i4][mt].w, lo ? bv.y : bv.w, acc[mt], 0, 0, 0);
  };

  // ---- the stage being computed: a countdown to the end of its tile; the tile itself is located when it is needed ----
  int c_it = 0, c_left = a.nstage;
  // The residual tile is requested one stage before the stage that ends the tile (a wave-uniform branch; vmcnt retires in
  // order, so requested any later the epilogue's wait for it would drain the whole operand prefetch).
  u32x4 rr[RES ? 4 * MT : 1];
  auto request_residual = [&]() {
    if constexpr (RES) {
      const ConvTile t = tile_of(c_it);
      const int x = t.x0 + p;
      const bool want = t.valid & (x < a.Wo);
      const int pix = (t.b * a.Ho + t.y) * a.Wo + x;
      const unsigned roff = want ? (unsigned)(pix * (int)a.rp + t.ct * 32 * MT + 4 * h) * 4u : 0x80000000u;
#pragma unroll
      for (int k = 0; k < 4 * MT; ++k) rr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrd, roff + 32u * k, 0, 0);
    }
  };

  auto epilogue = [&]() {
    const ConvTile t = tile_of(c_it);
    const int x = t.x0 + p;
    const bool store = t.valid & (x < a.Wo);
    const int pix = (t.b * a.Ho + t.y) * a.Wo + x;
    const int cbase = t.ct * 32 * MT + 4 * h;
    const unsigned ooff = store ? (unsigned)(pix * (int)a.op + cbase) * 4u : 0x80000000u;
    // SUMS: this wave's row of the item = chunk ((y / 4) * xt + x0 / 32) * 4 + wave of sample b; a row past the image, or an
    // item past the block's range, still writes its (zero) sums so that every chunk of the table is defined
    float* srow = nullptr;
    if constexpr (SUMS) {
      const bool in_range = c_it < iters;
      const int chunk = (((t.y - wave) >> 2) * a.xt + (t.x0 >> 5)) * 4 + wave;
      srow = in_range ? a.sums + ((int64_t)t.b * (a.hq * a.xt * 4) + chunk) * a.cout + cbase : nullptr;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(bias_lds + cbase + mt * 32 + 8 * g);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = acc[mt][4 * g + c] + bb[c];
          if constexpr (RES) v += __uint_as_float(rr[4 * mt + g][c]);
          // none / ReLU / LeakyReLU without a branch: max(v, 0) + slope * min(v, 0), slope = 1 / 0 / 0.01 (one of the two
          // terms is always zero, so this is exact)
          o[c] = __builtin_fmaf(a.slope, fminf(v, 0.f), fmaxf(v, 0.f));
          acc[mt][4 * g + c] = 0.0f;
        }
        if constexpr (SUMS) {
          float4 sv;
          sv.x = half_wave_sum(store ? o[0] : 0.f);
          sv.y = half_wave_sum(store ? o[1] : 0.f);
          sv.z = half_wave_sum(store ? o[2] : 0.f);
          sv.w = half_wave_sum(store ? o[3] : 0.f);
          if (p == 31 && srow) *reinterpret_cast<float4*>(srow + mt * 32 + 8 * g) = sv;
        }
        u32x4 ov;
        ov.x = __float_as_uint(o[0]); ov.y = __float_as_uint(o[1]); ov.z = __float_as_uint(o[2]); ov.w = __float_as_uint(o[3]);
        if (SMOS_CONV_KEEPS(8) || ov.x == 0x7fc12345u)      // (diagnostic builds can drop the stores: conv_diag.h)
          __builtin_amdgcn_raw_buffer_store_b128(ov, osrd, ooff + 4u * (mt * 32 + 8 * g), 0, 0);
      }
    }
  };

  SMOS_STAMPS_DECLARE();
  float4 b0[4], b1[4], b2[4], b3[4], af[4][MT];
  float4 ae0, ae1, ae2, ae3, ao0, ao1, ao2, ao3;       // weight slices of even / odd stages on their way to the ring
  // ---- prologue: slice 0 in the ring, slices 1 and 2 in registers, activations of stages 0, 1 and 2 requested ----
  // (requests in the order of the steady state -- weights, then activations, stage by stage -- so that the in-flight
  // picture hipcc's wait-count pass sees at the loop head is the same from the prologue and from the back edge)
  load_a(ae0, ae1, ae2, ae3);
  load_b(b0);
  advance_b();
  if (pb_left == 0) next_tile_b();
  park(0, ae0, ae1, ae2, ae3);
  load_a(ao0, ao1, ao2, ao3);
  load_b(b1);
  advance_b();
  if (pb_left == 0) next_tile_b();
  load_a(ae0, ae1, ae2, ae3);
  load_b(b2);
  advance_b();
  if (pb_left == 0) next_tile_b();
  if (RES && c_left == 1) request_residual();
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (256 * k < a.cout) bias_lds[tid + 256 * k] = bias_r[k];      // Cout <= 2048; rounded up to whole 256s in the LDS size
  ring_barrier();
  read_a(af, 0, 0);
  read_a(af, 0, 1);

  // ---- one stage g.  bc: activations of this stage (landed); bp: register set stage g + 3 is requested into;
  //      sc / sn: ring slots of this stage and of the next; (n0..n3): registers holding slice g + 1 (requested two stages ago).
  //   G0 | park slice g + 1, request slice g + 3 into the same registers, fragments i4 = 2, 3 of this stage
  //   G1 | request the activations of stage g + 3
  //   G2 | step the request position; barrier (publishes slice g + 1); fragments i4 = 0, 1 of the next stage
  //   G3 | rare branches: end of a tile (epilogue), residual request for a tile about to end, next tile of the requests
  // Slot sn = (g + 1) % 4 was last read in stage g - 3.
#if SMOS_CONV_SCHED == 1
// Finer cut: eight half groups of 2 * MT MFMAs, one small piece of the stage's non-matrix work after each.  A piece only
// overlaps with the matrix pipe while the MFMA issued just before it is executing (64 cycles), so the pieces are kept to a
// handful of instructions each.
#define SMOS_STAGE(bc, bp, sc, sn, n0, n1, n2, n3) \
  do {                                             \
    mfma_half(af, bc[0], 0, true);                 \
    SMOS_FENCE();                                  \
    park(sn, n0, n1, n2, n3);                      \
    SMOS_FENCE();                                  \
    mfma_half(af, bc[0], 0, false);                \
    SMOS_FENCE();                                  \
    load_a(n0, n1, n2, n3);                        \
    SMOS_FENCE();                                  \
    mfma_half(af, bc[1], 1, true);                 \
    SMOS_FENCE();                                  \
    read_a(af, sc, 2);                             \
    read_a(af, sc, 3);                             \
    SMOS_FENCE();                                  \
    mfma_half(af, bc[1], 1, false);                \
    SMOS_FENCE();                                  \
    addr_b();                                      \
    load_b_half(bp, 0);                            \
    SMOS_FENCE();                                  \
    mfma_half(af, bc[2], 2, true);                 \
    SMOS_FENCE();                                  \
    load_b_half(bp, 2);                            \
    SMOS_FENCE();                                  \
    mfma_half(af, bc[2], 2, false);                \
    SMOS_FENCE();                                  \
    advance_b();                                   \
    SMOS_FENCE();                                  \
    mfma_half(af, bc[3], 3, true);                 \
    SMOS_FENCE();                                  \
    ring_barrier();                                \
    read_a(af, sn, 0);                             \
    read_a(af, sn, 1);                             \
    SMOS_FENCE();                                  \
    mfma_half(af, bc[3], 3, false);                \
    SMOS_FENCE();                                  \
    if (--c_left == 0) {                           \
      epilogue();                                  \
      c_left = a.nstage;                           \
      ++c_it;                                      \
    }                                              \
    if (RES && c_left == 1) request_residual();    \
    if (pb_left == 0) next_tile_b();               \
  } while (0)
#endif      // SMOS_CONV_SCHED == 0 (diagnostic builds): conv_diag.h

  SMOS_STAMPS_BEGIN();
  // Four stages per trip (the register sets and ring slots rotate with period 4); the loop runs whole trips only and the
  // last 0..3 stages follow as straight-line code.  hipcc's wait-count pass honours every path of the control-flow graph:
  // with guarded stages inside the loop ("if (g + 1 < total) stage 2; if (g + 2 < total) stage 3; ...") there are paths on
  // which a stage is skipped and a later one runs, and with a break in the middle the structurizer's exit block falls
  // through to the loop header -- on such paths the register set a stage consumes was requested only one stage earlier, so
  // the pass emitted vmcnt(3) instead of vmcnt(13) at the head of two of the four stages and the three-stage prefetch
  // collapsed to less than one (found by reading the ISA; the stage heads must read vmcnt(13) / (15) / (19) for MT 1 / 2 / 4).
  int g = 0;
#pragma unroll 1
  for (; g + 4 <= total; g += 4) {
    SMOS_STAGE(b0, b3, 0, 1, ao0, ao1, ao2, ao3);
    SMOS_STAGE(b1, b0, 1, 2, ae0, ae1, ae2, ae3);
    SMOS_STAGE(b2, b1, 2, 3, ao0, ao1, ao2, ao3);
    SMOS_STAGE(b3, b2, 3, 0, ae0, ae1, ae2, ae3);
  }
  if (g < total) {
    SMOS_STAGE(b0, b3, 0, 1, ao0, ao1, ao2, ao3);
    if (g + 1 < total) {
      SMOS_STAGE(b1, b0, 1, 2, ae0, ae1, ae2, ae3);
      if (g + 2 < total) SMOS_STAGE(b2, b1, 2, 3, ao0, ao1, ao2, ao3);
    }
  }
  SMOS_STAMPS_END();
}

}  // namespace smos

using namespace smos;

static const int kDefaultBlocksPerCu[3] = {2, 2, 2};      // MT = 1, 2, 4

template <int MT, bool RES, bool SUMS = false>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
  const size_t lds = (size_t)4 * 256 * MT * sizeof(float4) + (size_t)((a.cout + 255) / 256 * 256) * sizeof(float);
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&conv_igemm<MT, RES, SUMS>), 4 * 256 * MT * sizeof(float4) + 8192, 256, &ks, "conv_cl"))
    return rc;
  // Resident blocks per CU.  Alone on the GPU two are best (tools/ubench_conv.py); inside the two-stream step the other
  // stream's kernels need room on the CU to run beside a convolution at all.  SMOS_CONV_BLOCKS_PER_CU = "n" or "n1,n2,n4"
  // (per MT) overrides the default for tuning.
  static const int want_per_cu = [] {
    int v[3] = {kDefaultBlocksPerCu[0], kDefaultBlocksPerCu[1], kDefaultBlocksPerCu[2]};
    if (const char* e = getenv("SMOS_CONV_BLOCKS_PER_CU")) {
      int a = 0, b = 0, c = 0;
      const int n = sscanf(e, "%d,%d,%d", &a, &b, &c);
      if (n == 1 && a >= 1 && a <= 8) v[0] = v[1] = v[2] = a;
      if (n == 3 && a >= 1 && a <= 8 && b >= 1 && b <= 8 && c >= 1 && c <= 8) v[0] = a, v[1] = b, v[2] = c;
    }
    return v[MT == 1 ? 0 : MT == 2 ? 1 : 2];
  }();
  const int per_cu = ks.per_cu < want_per_cu ? ks.per_cu : want_per_cu;
  const int64_t cap = conv_grid_cap((int64_t)ks.cus * per_cu);
  const unsigned grid = (unsigned)(a.n_items < cap ? a.n_items : cap);
  hipLaunchKernelGGL((conv_igemm<MT, RES, SUMS>), dim3(grid), dim3(256), lds, s, a);
  return check_launch("conv_cl");
}

// w: [Cout][Cin][KH][KW] reordered by ops.conv_prepare for the given MT.  x / res / out: channels-last rows with the given
// pitches (floats), 16-byte aligned.  Replaces conv2d -> BatchNorm (folded) -> ReLU / LeakyReLU (-> + residual -> ReLU) of
// networks/backbone.py:136-159 and multi_view_encoder.py:460-497 in one launch.
extern "C" int64_t smos_conv_cl_sum_chunks(int64_t Ho, int64_t Wo) { return ((Ho + 3) / 4) * ((Wo + 31) / 32) * 4; }

extern "C" int smos_conv_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, const float* res,
                            int64_t res_pitch, float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t Cin,
                            int64_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_h, int32_t pad_w, int32_t mt,
                            int32_t act, float* chan_sums, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && (mt == 1 || mt == 2 || mt == 4) &&
                   Cout % (32 * mt) == 0, "conv_cl: Cin must be a multiple of 32 and Cout of 32 * mt (mt in {1, 2, 4})");
  SMOS_REQUIRE(KH >= 1 && KW >= 1 && KH <= 7 && KW <= 7 && (stride == 1 || stride == 2) && pad_h >= 0 && pad_w >= 0 &&
                   act >= 0 && act <= 2, "conv_cl: kernel up to 7 x 7, stride 1 or 2");
  const int64_t Ho = (H + 2 * pad_h - KH) / stride + 1, Wo = (W + 2 * pad_w - KW) / stride + 1;
  SMOS_REQUIRE(Ho > 0 && Wo > 0 && Cout <= 2048, "conv_cl: empty output / more than 2048 output channels");
  SMOS_REQUIRE(x && wprep && out && x_pitch >= Cin && out_pitch >= Cout && x_pitch % 4 == 0 && out_pitch % 4 == 0 &&
                   (!res || (res_pitch >= Cout && res_pitch % 4 == 0)), "conv_cl: null pointer / bad pitch");
  SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(wprep)) & 15) == 0,
               "conv_cl: pointers must be 16-byte aligned");
  const int64_t hq = (Ho + 3) / 4, xt = (Wo + 31) / 32, nct = Cout / (32 * mt);
  SMOS_REQUIRE(B * H * W * x_pitch * 4 < (1LL << 31) && B * Ho * Wo * out_pitch * 4 < (1LL << 31) &&
                   (!res || B * Ho * Wo * res_pitch * 4 < (1LL << 31)), "conv_cl: a tensor larger than 2 GiB (32-bit buffer offsets)");
  SMOS_REQUIRE(B * hq * xt * nct < (1LL << 30) && (int64_t)KH * KW * (Cin / 32) * nct < (1LL << 20), "conv_cl: too many tiles");
  ConvArgs a;
  a.x = x; a.w = reinterpret_cast<const float4*>(wprep); a.bias = bias; a.res = res; a.out = out; a.sums = chan_sums;
  a.xp = x_pitch; a.rp = res_pitch; a.op = out_pitch;
  a.B = (int)B; a.H = (int)H; a.W = (int)W; a.Ho = (int)Ho; a.Wo = (int)Wo;
  a.KH = KH; a.KW = KW; a.S = stride; a.PH = pad_h; a.PW = pad_w;
  a.nch = (int)(Cin / 32); a.nstage = KH * KW * a.nch; a.nct = (int)nct;
  a.hq = (int)hq; a.xt = (int)xt; a.n_items = (int)(B * hq * xt * nct);
  a.slope = act == 0 ? 1.0f : act == 1 ? 0.0f : 0.01f;
  a.x_bytes = (int)(B * H * W * x_pitch * 4);
  SMOS_STAMPS_HOST(a);
  a.r_bytes = res ? (int)(B * Ho * Wo * res_pitch * 4) : 0;
  a.o_bytes = (int)(B * Ho * Wo * out_pitch * 4);
  a.cout = (int)Cout;
  if (chan_sums) {
    SMOS_REQUIRE(!res && (reinterpret_cast<uintptr_t>(chan_sums) & 15) == 0, "conv_cl: channel sums need res == NULL and 16-byte alignment");
    if (mt == 1) return launch_conv<1, false, true>(a, (hipStream_t)stream);
    if (mt == 2) return launch_conv<2, false, true>(a, (hipStream_t)stream);
    return launch_conv<4, false, true>(a, (hipStream_t)stream);
  }
  if (res) {
    SMOS_REQUIRE(mt <= 2, "conv_cl: a residual input needs mt <= 2 (register budget)");
    return mt == 1 ? launch_conv<1, true>(a, (hipStream_t)stream) : launch_conv<2, true>(a, (hipStream_t)stream);
  }
  if (mt == 1) return launch_conv<1, false>(a, (hipStream_t)stream);
  if (mt == 2) return launch_conv<2, false>(a, (hipStream_t)stream);
  return launch_conv<4, false>(a, (hipStream_t)stream);
}
