// Stride-1 channels-last fp32 convolution with KW in {3, 5, 7} and 32 or 64 output channels per block: the same implicit GEMM as
// csrc/conv_igemm.hip (transposed form, streamed weight ring, fused epilogue) with a different ACTIVATION path.
//
// conv_igemm requests the B operand of every stage straight from global memory: lane (p, h) = pixel p, 16 bytes per load,
// so one load instruction touches 32 cache lines for 32 bytes each, and the KW taps of a kernel row re-request the same
// row shifted by one pixel.  The ablation (profiles/r02c_conv_ablation.txt) prices those requests at 8-17 % of a layer at
// 32 output channels per block, and the counters (profiles/r02c_conv_memory_path_pmc.txt) show the texture addresser 44 %
// busy for 10 bytes per clock per CU.  Here a wave stages each input row ONCE per (kernel row, 32-channel chunk):
//   group  = (ky, chunk): KW stages (kx = 0 .. KW-1) that share one staged row of 32 + KW - 1 pixels x 32 channels;
//   fill   = 5 fully coalesced 16-byte loads per lane (8 lanes per 128-byte pixel chunk: 8 lines per instruction instead
//            of 32), requested in the first stage of the PREVIOUS group, written to the wave's own LDS row buffer
//            (pixel pitch 36 floats: conflict-free 16-byte reads) in its last stage -- no barrier, the buffer is private;
//   B read = four ds_read_b128 per stage at pixel offset kx, issued one stage ahead into the other register set.
// Weights, epilogue, residual, channel sums: as in conv_igemm (one stage = 16 MFMAs; slices through the 4-slot ring).
// Stage order inside a tile: (ky, chunk, kx) -- ops.conv_prepare(order="rows") packs the weights accordingly.
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"

namespace smos {

template <int KW, int MT, bool RES, bool SUMS>
__global__ __launch_bounds__(256, 2) void conv_rows(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 ring[];     // 4 slots x 256 * MT float4 | bias | 4 waves x 2 row buffers
  constexpr int kSlot = 256 * MT;
  constexpr int kWt = 32 + KW - 1;                  // pixels of a staged row
  constexpr int kNL = (kWt * 8 + 63) / 64;          // float4 per lane of a staged row
  constexpr int kPitch = 36;                        // floats per pixel in LDS
  constexpr int kRow = kWt * kPitch;                // floats per row buffer (multiple of 4)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  float* bias_lds = reinterpret_cast<float*>(ring + 4 * kSlot);
  float* rowbuf = bias_lds + (a.cout + 255) / 256 * 256 + wave * 2 * kRow;

  const int per_block = (a.n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nb = (int)gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = (int)blockIdx.x & 7;
  const int lblock = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
  const int first = lblock * per_block;
  const int iters = a.n_items - first < per_block ? a.n_items - first : per_block;
  const int ngroup = a.KH * a.nch;                  // groups per tile
  const int total_groups = iters * ngroup;
  if (total_groups <= 0) return;
  const int total = total_groups * KW;              // stages

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

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? a.r_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? a.cout * 4 : 0, 0x00020000);
  float bias_r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bias_r[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bsrd, (unsigned)(tid + 256 * k) * 4u, 0, 0));

  // ---- row staging: the request position runs one group ahead of the group being computed ----
  const int xp = (int)a.xp;
  const int fq = lane & 7, fpix0 = lane >> 3;          // piece i of a row: pixel fpix0 + 8 i, float4 fq of its 32 channels
  int f_it = 0, f_ky = 0, f_ch = 0;
  int f_rowbase = 0, f_xs0 = 0;
  bool f_ok = false;
  auto locate_f = [&]() {          // scalar: row (tile f_it, kernel row f_ky) of the input
    const ConvTile t = tile_of(f_it);
    const int yin = t.y - a.PH + f_ky;
    f_ok = t.valid & ((unsigned)yin < (unsigned)a.H);
    f_xs0 = t.x0 - a.PW;
    f_rowbase = ((t.b * a.H + yin) * a.W + f_xs0) * xp;
  };
  locate_f();
  float4 fr[kNL];
  auto fill_request = [&]() {
#pragma unroll
    for (int i = 0; i < kNL; ++i) {
      const int pix = fpix0 + 8 * i;
      const bool ok = f_ok & (pix < kWt) & ((unsigned)(f_xs0 + pix) < (unsigned)a.W);
      const unsigned voff = ok ? (unsigned)(f_rowbase + pix * xp + f_ch * 32 + fq * 4) * 4u : 0x80000000u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrd, voff, 0, 0);
      fr[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto fill_advance = [&]() {      // next group: channel chunk fastest, then the kernel row, then the tile
    if (++f_ch == a.nch) {
      f_ch = 0;
      if (++f_ky == a.KH) {
        f_ky = 0;
        ++f_it;
      }
      locate_f();
    }
  };
  auto fill_write = [&](int buf) {
    float* dst = rowbuf + buf * kRow + fq * 4;
#pragma unroll
    for (int i = 0; i < kNL; ++i) {
      const int pix = fpix0 + 8 * i;
      if (8 * i + 7 < kWt || pix < kWt) *reinterpret_cast<float4*>(dst + pix * kPitch) = fr[i];
    }
  };
  const float* b_lane = rowbuf + p * kPitch + 4 * h;
  auto read_b = [&](float4 (&bset)[4], int buf, int kx) {
    const float* src = b_lane + buf * kRow + kx * kPitch;
#pragma unroll
    for (int j = 0; j < 4; ++j) bset[j] = *reinterpret_cast<const float4*>(src + 8 * j);
  };

  // ---- weight requests (three stages ahead), as in conv_igemm ----
  const int n_slices = a.nct * a.nstage;
  int pa_slice = (first % a.nct) * a.nstage, pa_g = 0;
  auto load_a = [&](float4& r0, float4& r1) {      // named registers: an array here ends up in scratch for MT > 1
    const float4* wsrc = a.w + (int64_t)(pa_g < total ? pa_slice : 0) * kSlot + tid;
    r0 = wsrc[0];
    if constexpr (MT > 1) r1 = wsrc[256];
    ++pa_g;
    pa_slice = pa_slice + 1 == n_slices ? 0 : pa_slice + 1;
  };
  float4* ring_tid = ring + tid;
  const float4* ring_lane = ring + lane;
  auto park = [&](int slot, const float4& r0, const float4& r1) {
    ring_tid[slot * kSlot] = r0;
    if constexpr (MT > 1) ring_tid[slot * kSlot + 256] = r1;
  };
  auto read_a = [&](float4 (&af)[4][MT], int slot, int i4) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[i4][mt] = ring_lane[slot * kSlot + (i4 * MT + mt) * 64];
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
  auto mfma_half = [&](const float4 (&af)[4][MT], const float4& bv, int i4, bool lo) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo ? af[i4][mt].x : af[i4][mt].z, lo ? bv.x : bv.z, acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(lo ? af[i4][mt].y : af[i4][mt].w, lo ? bv.y : bv.w, acc[mt], 0, 0, 0);
  };

  int c_it = 0, c_left = a.nstage;
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
    float* srow = nullptr;
    if constexpr (SUMS) {
      const int chunk = (((t.y - wave) >> 2) * a.xt + (t.x0 >> 5)) * 4 + wave;
      srow = a.sums + ((int64_t)t.b * (a.hq * a.xt * 4) + chunk) * a.cout + cbase;
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
          o[c] = __builtin_fmaf(a.slope, fminf(v, 0.f), fmaxf(v, 0.f));
          acc[mt][4 * g + c] = 0.0f;
        }
        if constexpr (SUMS) {
          float4 sv;
          sv.x = half_wave_sum(store ? o[0] : 0.f);
          sv.y = half_wave_sum(store ? o[1] : 0.f);
          sv.z = half_wave_sum(store ? o[2] : 0.f);
          sv.w = half_wave_sum(store ? o[3] : 0.f);
          if (p == 31) *reinterpret_cast<float4*>(srow + mt * 32 + 8 * g) = sv;
        }
        u32x4 ov;
        ov.x = __float_as_uint(o[0]); ov.y = __float_as_uint(o[1]); ov.z = __float_as_uint(o[2]); ov.w = __float_as_uint(o[3]);
        __builtin_amdgcn_raw_buffer_store_b128(ov, osrd, ooff + 4u * (mt * 32 + 8 * g), 0, 0);
      }
    }
  };

  float4 bA[4], bB[4], af[4][MT];
  float4 ae0, ae1, ao0, ao1;      // weight slices of odd / even stages on their way to the ring (stage g parks slice g + 1)
  // ---- prologue: row of group 0 staged, weight slice 0 in the ring, slices 1 and 2 in registers ----
  load_a(ae0, ae1);
  fill_request();                       // group 0
  fill_advance();
  park(0, ae0, ae1);
  fill_write(0);
  load_a(ao0, ao1);
  load_a(ae0, ae1);
  if (RES && c_left == 1) request_residual();
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (256 * k < a.cout) bias_lds[tid + 256 * k] = bias_r[k];
  ring_barrier();
  read_a(af, 0, 0);
  read_a(af, 0, 1);
  read_b(bA, 0, 0);

  // One stage (kx of group G).  bc: activations of this stage; bn: set the next stage's are read into.  n0: register holding
  // weight slice g + 1.  buf = G & 1.  The pieces sit after the half groups of two MFMAs as in conv_igemm:
  //   park slice g + 1 | request slice g + 3 | fragments 2, 3 | [first stage: request the row of group G + 1]
  //   | [last stage: write that row to the other buffer] | next stage's activations | barrier, fragments 0, 1 of the next
  //   stage | rare: end of a tile
  int g = 0;                            // stage counter (ring slot = g & 3)
#define SMOS_RSTAGE(KX, bc, bn, n0, n1, buf)                                        \
  do {                                                                          \
    const int sc_ = g & 3, sn_ = (g + 1) & 3;                                   \
    mfma_half(af, bc[0], 0, true);                                              \
    SMOS_FENCE();                                                               \
    park(sn_, n0, n1);                                                          \
    SMOS_FENCE();                                                               \
    mfma_half(af, bc[0], 0, false);                                             \
    SMOS_FENCE();                                                               \
    load_a(n0, n1);                                                             \
    SMOS_FENCE();                                                               \
    mfma_half(af, bc[1], 1, true);                                              \
    SMOS_FENCE();                                                               \
    read_a(af, sc_, 2);                                                         \
    read_a(af, sc_, 3);                                                         \
    SMOS_FENCE();                                                               \
    mfma_half(af, bc[1], 1, false);                                             \
    SMOS_FENCE();                                                               \
    if constexpr ((KX) == 0) {                                                  \
      fill_request();                                                           \
      fill_advance();                                                           \
    }                                                                           \
    SMOS_FENCE();                                                               \
    mfma_half(af, bc[2], 2, true);                                              \
    SMOS_FENCE();                                                               \
    if constexpr ((KX) == KW - 1) fill_write((buf) ^ 1);                        \
    SMOS_FENCE();                                                               \
    mfma_half(af, bc[2], 2, false);                                             \
    SMOS_FENCE();                                                               \
    if constexpr ((KX) == KW - 1) read_b(bn, (buf) ^ 1, 0);                     \
    else read_b(bn, (buf), (KX) + 1);                                           \
    SMOS_FENCE();                                                               \
    mfma_half(af, bc[3], 3, true);                                              \
    SMOS_FENCE();                                                               \
    ring_barrier();                                                             \
    read_a(af, sn_, 0);                                                         \
    read_a(af, sn_, 1);                                                         \
    SMOS_FENCE();                                                               \
    mfma_half(af, bc[3], 3, false);                                             \
    SMOS_FENCE();                                                               \
    ++g;                                                                        \
    if (--c_left == 0) {                                                        \
      epilogue();                                                               \
      c_left = a.nstage;                                                        \
      ++c_it;                                                                   \
    }                                                                           \
    if (RES && c_left == 1) request_residual();                                 \
  } while (0)

  // a group of KW stages starting at an even (E) or odd (O) stage: register sets alternate with the stage parity
#define SMOS_RGROUP_E(buf)                                       \
  do {                                                           \
    SMOS_RSTAGE(0, bA, bB, ao0, ao1, buf);                            \
    SMOS_RSTAGE(1, bB, bA, ae0, ae1, buf);                            \
    SMOS_RSTAGE(2, bA, bB, ao0, ao1, buf);                            \
    if constexpr (KW > 3) {                                      \
      SMOS_RSTAGE(3, bB, bA, ae0, ae1, buf);                          \
      SMOS_RSTAGE(4, bA, bB, ao0, ao1, buf);                          \
    }                                                            \
    if constexpr (KW > 5) {                                      \
      SMOS_RSTAGE(5, bB, bA, ae0, ae1, buf);                          \
      SMOS_RSTAGE(6, bA, bB, ao0, ao1, buf);                          \
    }                                                            \
  } while (0)
#define SMOS_RGROUP_O(buf)                                       \
  do {                                                           \
    SMOS_RSTAGE(0, bB, bA, ae0, ae1, buf);                            \
    SMOS_RSTAGE(1, bA, bB, ao0, ao1, buf);                            \
    SMOS_RSTAGE(2, bB, bA, ae0, ae1, buf);                            \
    if constexpr (KW > 3) {                                      \
      SMOS_RSTAGE(3, bA, bB, ao0, ao1, buf);                          \
      SMOS_RSTAGE(4, bB, bA, ae0, ae1, buf);                          \
    }                                                            \
    if constexpr (KW > 5) {                                      \
      SMOS_RSTAGE(5, bA, bB, ao0, ao1, buf);                          \
      SMOS_RSTAGE(6, bB, bA, ae0, ae1, buf);                          \
    }                                                            \
  } while (0)

  // whole trips of two groups (KW is odd: the stage parity flips with every group), then at most one group; see
  // conv_igemm for why the loop body must not contain skipped stages
  int G = 0;
#pragma unroll 1
  for (; G + 2 <= total_groups; G += 2) {
    SMOS_RGROUP_E(0);
    SMOS_RGROUP_O(1);
  }
  if (G < total_groups) SMOS_RGROUP_E(0);
#undef SMOS_RSTAGE
#undef SMOS_RGROUP_E
#undef SMOS_RGROUP_O
}

}  // namespace smos

using namespace smos;

template <int KW, int MT, bool RES, bool SUMS>
static int launch_rows(const ConvArgs& a, hipStream_t s) {
  const size_t rows = (size_t)4 * 2 * (32 + KW - 1) * 36 * sizeof(float);
  const size_t lds = (size_t)4 * 256 * MT * sizeof(float4) + (size_t)((a.cout + 255) / 256 * 256) * sizeof(float) + rows;
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&conv_rows<KW, MT, RES, SUMS>), 4 * 256 * MT * sizeof(float4) + 8192 + rows, 256, &ks, "conv_rows_cl"))
    return rc;
  const int per_cu = ks.per_cu < 2 ? ks.per_cu : 2;
  const int64_t cap = conv_grid_cap((int64_t)ks.cus * per_cu);
  const unsigned grid = (unsigned)(a.n_items < cap ? a.n_items : cap);
  hipLaunchKernelGGL((conv_rows<KW, MT, RES, SUMS>), dim3(grid), dim3(256), lds, s, a);
  return check_launch("conv_rows_cl");
}

// Same contract as smos_conv_cl with stride 1, "same" padding, KW in {3, 5, 7} and mt in {1, 2}, except for the weight
// order: wprep[((((ct * KH + ky) * (Cin / 32) + cc) * KW + kx) * 4 + i4) * mt + m][lane][c] (ops.conv_prepare(..., order="rows")).
extern "C" int smos_conv_rows_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, const float* res,
                                 int64_t res_pitch, float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t Cin,
                                 int64_t Cout, int32_t KH, int32_t KW, int32_t mt, int32_t act, float* chan_sums,
                                 smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && (mt == 1 || mt == 2) && Cout % (32 * mt) == 0 &&
                   Cout <= 2048, "conv_rows_cl: Cin a multiple of 32, Cout of 32 * mt (mt in {1, 2}; Cout <= 2048)");
  SMOS_REQUIRE(KH >= 1 && KH <= 7 && (KH & 1) && (KW == 3 || KW == 5 || KW == 7) && act >= 0 && act <= 2,
               "conv_rows_cl: odd KH <= 7, KW in {3, 5, 7}");
  SMOS_REQUIRE(x && wprep && out && x_pitch >= Cin && out_pitch >= Cout && x_pitch % 4 == 0 && out_pitch % 4 == 0 &&
                   (!res || (res_pitch >= Cout && res_pitch % 4 == 0)) && !(res && chan_sums), "conv_rows_cl: null pointer / bad pitch");
  SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(wprep) | reinterpret_cast<uintptr_t>(chan_sums)) & 15) == 0,
               "conv_rows_cl: pointers must be 16-byte aligned");
  const int64_t hq = (H + 3) / 4, xt = (W + 31) / 32, nct = Cout / (32 * mt);
  SMOS_REQUIRE(B * H * W * x_pitch * 4 < (1LL << 31) && B * H * W * out_pitch * 4 < (1LL << 31) &&
                   (!res || B * H * W * res_pitch * 4 < (1LL << 31)), "conv_rows_cl: a tensor larger than 2 GiB (32-bit buffer offsets)");
  SMOS_REQUIRE(B * hq * xt * nct < (1LL << 30) && (int64_t)KH * KW * (Cin / 32) * nct < (1LL << 20), "conv_rows_cl: too many tiles");
  ConvArgs a;
  a.x = x; a.w = reinterpret_cast<const float4*>(wprep); a.bias = bias; a.res = res; a.out = out; a.sums = chan_sums;
  a.xp = x_pitch; a.rp = res_pitch; a.op = out_pitch;
  a.B = (int)B; a.H = (int)H; a.W = (int)W; a.Ho = (int)H; a.Wo = (int)W;
  a.KH = KH; a.KW = KW; a.S = 1; a.PH = KH / 2; a.PW = KW / 2;
  a.nch = (int)(Cin / 32); a.nstage = KH * KW * a.nch; a.nct = (int)nct;
  a.hq = (int)hq; a.xt = (int)xt; a.n_items = (int)(B * hq * xt * nct);
  a.slope = act == 0 ? 1.0f : act == 1 ? 0.0f : 0.01f;
  a.x_bytes = (int)(B * H * W * x_pitch * 4);
  SMOS_STAMPS_HOST_NONE(a);
  a.r_bytes = res ? (int)(B * H * W * res_pitch * 4) : 0;
  a.o_bytes = (int)(B * H * W * out_pitch * 4);
  a.cout = (int)Cout;
  hipStream_t s = (hipStream_t)stream;
#define SMOS_ROWS_DISPATCH(KW_)                                                                  \
  if (KW == KW_) {                                                                               \
    if (mt == 1) {                                                                               \
      if (chan_sums) return launch_rows<KW_, 1, false, true>(a, s);                              \
      if (res) return launch_rows<KW_, 1, true, false>(a, s);                                    \
      return launch_rows<KW_, 1, false, false>(a, s);                                            \
    }                                                                                            \
    if (chan_sums) return launch_rows<KW_, 2, false, true>(a, s);                                \
    if (res) return launch_rows<KW_, 2, true, false>(a, s);                                      \
    return launch_rows<KW_, 2, false, false>(a, s);                                              \
  }
  SMOS_ROWS_DISPATCH(3)
  SMOS_ROWS_DISPATCH(5)
  SMOS_ROWS_DISPATCH(7)
#undef SMOS_ROWS_DISPATCH
  return SMOS_ERR_UNSUPPORTED;
}
