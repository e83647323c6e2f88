// 3x3 convolution of a bilinearly upsampled map without upsampling it first (gfx950).
//
// The decoder (multi_view_encoder.py:441-453) resizes the 128-channel maps of the two coarser stages to 256 x 256
// (F.interpolate, bilinear, align_corners=True), concatenates them with the 64-channel fine map and runs conv_1
// (3x3, 320 -> 128): 48.3 GFLOP per sample, 37 % of all FLOPs of the network -- 80 % of them spent on
// inputs that are interpolations of 16x / 4x smaller maps.  Channel mixing and spatial operators commute:
//     conv3x3(up(x)) = sum_{ky,kx} shift_{ky,kx}( up( W_{ky,kx} x ) ),        W_{ky,kx}: the [Cout, Cin] matrix of one tap,
// so the nine tap products are taken at the SOURCE resolution (one GEMM [pixels, Cin] x [Cin, 9 Cout], 16x / 4x fewer
// pixels) and only the cheap spatial part runs at 256 x 256, separably:
//   x pass (upconv_xpass): T[b, ky, ys, X, c]  = sum_kx  up_x( Z[b, ys, :, (3 ky + kx) C + c] )[X + kx - 1]
//   y pass (upconv_ypass): out[b, Y, X, c]     = act( conv_a + bias + sum_src sum_ky  up_y( T_src[b, ky, :, X, c] )[Y + ky - 1] )
// with the zero padding of the convolution applied at the upsampled resolution (taps that leave the 256 x 256 image are
// dropped) and conv_a the ordinary convolution of the channels that are NOT upsampled.  Exact in real arithmetic;
// in float32 it differs from the direct form by summation order only.  conv_1 falls from 48.3 to 15.7 GFLOP per sample
// and the 320-channel concatenated input (336 MB) is never built.
// Interpolation weights: ATen's align_corners=True formula, as in upsample_concat_cl (cl_kernels.hip).
#include "smos_common.h"

namespace smos {

struct Lerp {
  int i0, step;     // source index and 0/1 step to the second tap
  float w0, w1;
};

__device__ __forceinline__ Lerp lerp_of(int dst, int n_src, int n_dst) {
  const float r = n_dst > 1 ? (float)(n_src - 1) / (float)(n_dst - 1) : 0.0f;
  const float s = r * dst;
  Lerp l;
  l.i0 = (int)s;
  l.step = (l.i0 < n_src - 1) ? 1 : 0;
  l.w1 = s - l.i0;
  l.w0 = 1.0f - l.w1;
  return l;
}

// Both passes issue ALL their loads unconditionally and up front: a tap that leaves the image reads a clamped (valid)
// address and enters with weight 0.  Written with "if (outside) continue" every tap's pair of loads sat under its own
// lane-dependent branch, and hipcc's wait-count pass answered each with s_waitcnt vmcnt(0): six to seven serialised memory
// round trips per output.  Index arithmetic is 32-bit (the host checks the element counts), addresses 64-bit.

// z [B, Hs, Ws, 9*C] (tap-major blocks of C channels), t [B, 3, Hs, Wo, C]; one thread = 4 channels of one (b, ky, ys, X)
__global__ __launch_bounds__(kBlock) void upconv_xpass(const float* __restrict__ z, float* __restrict__ t, int B, int Hs, int Ws,
                                                       int C4, int Wo) {
  const int total = B * 3 * Hs * Wo * C4;
  const int C = C4 * 4;
  for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < total; i += (int)(gridDim.x * blockDim.x)) {
    const int q = (i % C4) * 4;
    int r = i / C4;
    const int X = r % Wo;
    r /= Wo;
    const int ys = r % Hs;
    r /= Hs;
    const int ky = r % 3, b = r / 3;
    const float* zrow = z + ((int64_t)(b * Hs + ys) * Ws) * (9 * C) + (3 * ky) * C + q;
    float4 v0[3], v1[3];
    float w0[3], w1[3];
    bool inb[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xs = X + kx - 1;
      const bool in = (xs >= 0) & (xs < Wo);          // zero padding of the convolution, at the upsampled resolution
      const Lerp l = lerp_of(min(max(xs, 0), Wo - 1), Ws, Wo);
      const float* p = zrow + (int64_t)l.i0 * (9 * C) + kx * C;
      v0[kx] = *reinterpret_cast<const float4*>(p);
      v1[kx] = *reinterpret_cast<const float4*>(p + (int64_t)l.step * (9 * C));
      w0[kx] = l.w0;
      w1[kx] = l.w1;
      inb[kx] = in;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      // the loads are unconditional (clamped address); a padded tap is dropped by a select on the VALUE at its consumer,
      // not by a zero weight: 0 * Inf / NaN of a border element must not leak into the output
      const float4 a0 = inb[kx] ? v0[kx] : zero4, a1 = inb[kx] ? v1[kx] : zero4;
      acc.x += w0[kx] * a0.x + w1[kx] * a1.x; acc.y += w0[kx] * a0.y + w1[kx] * a1.y;
      acc.z += w0[kx] * a0.z + w1[kx] * a1.z; acc.w += w0[kx] * a0.w + w1[kx] * a1.w;
    }
    *reinterpret_cast<float4*>(t + (int64_t)i * 4) = acc;
  }
}

struct YSrc {
  const float* t;   // [B, 3, Hs, Wo, C]; an absent source is passed as a copy of the other one with on = 0
  int Hs;
  float on;         // 1 / 0
};

// out = act(conv_a + bias + sum over the sources and ky of the y-interpolated x-pass rows); act: 0 none, 1 ReLU, 2 LeakyReLU(0.01)
__global__ __launch_bounds__(kBlock) void upconv_ypass(const float* __restrict__ conv_a, int64_t ap, const float* __restrict__ bias,
                                                       YSrc s1, YSrc s2, float* __restrict__ out, int64_t op, int B, int Ho, int Wo,
                                                       int C4, int act) {
  const int total = B * Ho * Wo * C4;
  const int C = C4 * 4;
  for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < total; i += (int)(gridDim.x * blockDim.x)) {
    const int q = (i % C4) * 4;
    int r = i / C4;
    const int X = r % Wo;
    r /= Wo;
    const int Y = r % Ho, b = r / Ho;
    const int64_t pix = (int64_t)(b * Ho + Y) * Wo + X;
    float4 acc = *reinterpret_cast<const float4*>(conv_a + pix * ap + q);
    const float4 bv = *reinterpret_cast<const float4*>(bias + q);
    float4 v0[2][3], v1[2][3];
    float w0[2][3], w1[2][3];
    bool inb[2][3];
#pragma unroll
    for (int si = 0; si < 2; ++si) {
      const YSrc s = si == 0 ? s1 : s2;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int ysrc = Y + ky - 1;
        const bool in = (ysrc >= 0) & (ysrc < Ho);
        const Lerp l = lerp_of(min(max(ysrc, 0), Ho - 1), s.Hs, Ho);
        const float* p = s.t + ((int64_t)((b * 3 + ky) * s.Hs + l.i0) * Wo + X) * C + q;
        v0[si][ky] = *reinterpret_cast<const float4*>(p);
        v1[si][ky] = *reinterpret_cast<const float4*>(p + (int64_t)l.step * Wo * C);
        w0[si][ky] = l.w0;
        w1[si][ky] = l.w1;
        inb[si][ky] = in & (s.on != 0.0f);
      }
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float4 a0 = inb[si][ky] ? v0[si][ky] : zero4, a1 = inb[si][ky] ? v1[si][ky] : zero4;   // select on the value (NaN-safe)
        acc.x += w0[si][ky] * a0.x + w1[si][ky] * a1.x; acc.y += w0[si][ky] * a0.y + w1[si][ky] * a1.y;
        acc.z += w0[si][ky] * a0.z + w1[si][ky] * a1.z; acc.w += w0[si][ky] * a0.w + w1[si][ky] * a1.w;
      }
    float4 o = make_float4(acc.x + bv.x, acc.y + bv.y, acc.z + bv.z, acc.w + bv.w);
    if (act == 1) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    } else if (act == 2) {
      o.x = o.x > 0.f ? o.x : 0.01f * o.x; o.y = o.y > 0.f ? o.y : 0.01f * o.y;
      o.z = o.z > 0.f ? o.z : 0.01f * o.z; o.w = o.w > 0.f ? o.w : 0.01f * o.w;
    }
    *reinterpret_cast<float4*>(out + pix * op + q) = o;
  }
}

// ---- both passes in one kernel: the x-pass rows never leave the registers ----
// One thread = 4 channels of one output COLUMN segment (b, X, rows [Y0, Y0 + strip)): it walks down the strip and keeps,
// per source, a window of three x-pass rows T[ky][base .. base + 2][X] (3 ky x 3 rows x float4).  Output row Y reads the
// source rows i0(Y + ky - 1) and i0 + step for ky = 0..2, all inside [i0(Y - 1), i0(Y - 1) + 2] when the y ratio is at most
// 1/2 (the host checks it), so the window only slides down, by one row at a time, and each x-pass row is computed once
// per strip.  The window slots are picked with wave-uniform selects (Y is the same for the whole block).  Same
// operations in the same order as upconv_xpass followed by upconv_ypass: z is read once (plus 2 rows per strip), t
// (0.3 GB written and read back at the network's sizes) does not exist.
struct XYSrc {
  const float* z;   // [B, Hs, Ws, 9 * C]; an absent source has on = 0 and is never read
  int Hs, Ws;
  int on;
};

struct XTaps {      // the three kx taps of one thread's column X in one source: offsets into a z row and weights
  int o0[3], o1[3];
  float w0[3], w1[3];
  bool in[3];
};

__device__ __forceinline__ XTaps x_taps(int X, int Ws, int Wo, int C) {
  XTaps t;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int xs = X + kx - 1;
    t.in[kx] = (xs >= 0) & (xs < Wo);
    const Lerp l = lerp_of(min(max(xs, 0), Wo - 1), Ws, Wo);
    t.o0[kx] = l.i0 * (9 * C) + kx * C;
    t.o1[kx] = t.o0[kx] + l.step * (9 * C);
    t.w0[kx] = l.w0;
    t.w1[kx] = l.w1;
  }
  return t;
}

// T[ky][ys][X] for ky = 0..2 (upconv_xpass's sum, same order); zrow = z + ((b * Hs + ys) * Ws) * 9 C + q
__device__ __forceinline__ void x_row(const float* __restrict__ zrow, const XTaps& xt, int C, float4 (&t)[3]) {
  float4 v0[3][3], v1[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      v0[ky][kx] = *reinterpret_cast<const float4*>(zrow + 3 * ky * C + xt.o0[kx]);
      v1[ky][kx] = *reinterpret_cast<const float4*>(zrow + 3 * ky * C + xt.o1[kx]);
    }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    float4 acc = zero4;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float4 a0 = xt.in[kx] ? v0[ky][kx] : zero4, a1 = xt.in[kx] ? v1[ky][kx] : zero4;   // select on the value (NaN-safe)
      acc.x += xt.w0[kx] * a0.x + xt.w1[kx] * a1.x; acc.y += xt.w0[kx] * a0.y + xt.w1[kx] * a1.y;
      acc.z += xt.w0[kx] * a0.z + xt.w1[kx] * a1.z; acc.w += xt.w0[kx] * a0.w + xt.w1[kx] * a1.w;
    }
    t[ky] = acc;
  }
}

__device__ __forceinline__ float4 pick3(int i, const float4& a, const float4& b, const float4& c) {
  return i == 0 ? a : (i == 1 ? b : c);
}

__global__ __launch_bounds__(kBlock, 2) void upconv_xy(const float* __restrict__ conv_a, int64_t ap, const float* __restrict__ bias, XYSrc s1,
                                                    XYSrc s2, float* __restrict__ out, int64_t op, int B, int Ho, int Wo, int C4,
                                                    int strip, int act) {
  const int C = C4 * 4;
  const int per_row = Wo * C4;                       // threads of one (b, strip): a block never straddles two of them
  const int blocks_per_row = (per_row + kBlock - 1) / kBlock;
  const int n_strips = (Ho + strip - 1) / strip;
  const int n_units = B * n_strips * blocks_per_row;
  for (int unit = (int)blockIdx.x; unit < n_units; unit += (int)gridDim.x) {
    const int i = (unit % blocks_per_row) * kBlock + (int)threadIdx.x;
    const int bs = unit / blocks_per_row;
    const int Y0 = (bs % n_strips) * strip, b = bs / n_strips;
    if (i >= per_row) continue;
    const int q = (i % C4) * 4, X = i / C4;
    const int Y1 = min(Y0 + strip, Ho);
    const float4 bv = *reinterpret_cast<const float4*>(bias + q);
    float4 win[2][3][3];                             // [source][slot][ky]
    int base[2] = {0, 0};
    XTaps xt[2];
#pragma unroll
    for (int si = 0; si < 2; ++si) {
      const XYSrc s = si == 0 ? s1 : s2;
      if (!s.on) continue;
      xt[si] = x_taps(X, s.Ws, Wo, C);
      base[si] = lerp_of(max(Y0 - 1, 0), s.Hs, Ho).i0;
#pragma unroll 1
      for (int slot = 0; slot < 3; ++slot) {         // rolled: one row's 18 loads in flight at a time, not 54
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          win[si][0][ky] = win[si][1][ky];
          win[si][1][ky] = win[si][2][ky];
        }
        const int ys = min(base[si] + slot, s.Hs - 1);
        x_row(s.z + ((int64_t)(b * s.Hs + ys) * s.Ws) * (9 * C) + q, xt[si], C, win[si][2]);
      }
    }
    for (int Y = Y0; Y < Y1; ++Y) {
      const int64_t pix = (int64_t)(b * Ho + Y) * Wo + X;
      float4 acc = *reinterpret_cast<const float4*>(conv_a + pix * ap + q);
#pragma unroll
      for (int si = 0; si < 2; ++si) {
        const XYSrc s = si == 0 ? s1 : s2;
        if (!s.on) continue;
        const int lo = lerp_of(max(Y - 1, 0), s.Hs, Ho).i0;
        if (lo > base[si]) {                         // wave-uniform: slide the window down one source row
          base[si] = lo;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            win[si][0][ky] = win[si][1][ky];
            win[si][1][ky] = win[si][2][ky];
          }
          const int ys = min(lo + 2, s.Hs - 1);
          x_row(s.z + ((int64_t)(b * s.Hs + ys) * s.Ws) * (9 * C) + q, xt[si], C, win[si][2]);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int ysrc = Y + ky - 1;
          const bool in = (ysrc >= 0) & (ysrc < Ho);
          const Lerp l = lerp_of(min(max(ysrc, 0), Ho - 1), s.Hs, Ho);
          const int i0 = l.i0 - base[si], i1 = i0 + l.step;
          const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 r0 = pick3(i0, win[si][0][ky], win[si][1][ky], win[si][2][ky]);
          const float4 r1 = pick3(i1, win[si][0][ky], win[si][1][ky], win[si][2][ky]);
          const float4 a0 = in ? r0 : zero4, a1 = in ? r1 : zero4;
          acc.x += l.w0 * a0.x + l.w1 * a1.x; acc.y += l.w0 * a0.y + l.w1 * a1.y;
          acc.z += l.w0 * a0.z + l.w1 * a1.z; acc.w += l.w0 * a0.w + l.w1 * a1.w;
        }
      }
      float4 o = make_float4(acc.x + bv.x, acc.y + bv.y, acc.z + bv.z, acc.w + bv.w);
      if (act == 1) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      } else if (act == 2) {
        o.x = o.x > 0.f ? o.x : 0.01f * o.x; o.y = o.y > 0.f ? o.y : 0.01f * o.y;
        o.z = o.z > 0.f ? o.z : 0.01f * o.z; o.w = o.w > 0.f ? o.w : 0.01f * o.w;
      }
      *reinterpret_cast<float4*>(out + pix * op + q) = o;
    }
  }
}

}  // namespace smos

using namespace smos;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int smos_upconv_xpass(const float* z, float* t, int64_t B, int64_t Hs, int64_t Ws, int64_t C, int64_t Wo,
                                 smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && C > 0 && C % 4 == 0 && Wo > 0, "upconv_xpass: bad sizes (C %% 4 must be 0)");
  SMOS_REQUIRE(z && t && aligned16(z) && aligned16(t), "upconv_xpass: null / unaligned pointer");
  SMOS_REQUIRE(B * 3 * Hs * Wo * (C / 4) < kMaxTotal32 && B * Hs < (1LL << 31), "upconv_xpass: too many elements for 32-bit indices");
  hipLaunchKernelGGL(upconv_xpass, dim3(grid_for(B * 3 * Hs * Wo * (C / 4), kBlock, 256 * 32)), dim3(kBlock), 0, (hipStream_t)stream, z, t,
                     (int)B, (int)Hs, (int)Ws, (int)(C / 4), (int)Wo);
  return check_launch("upconv_xpass");
}

extern "C" int smos_upconv_ypass(const float* conv_a, int64_t a_pitch, const float* bias, const float* t1, int64_t H1, const float* t2,
                                 int64_t H2, float* out, int64_t out_pitch, int64_t B, int64_t Ho, int64_t Wo, int64_t C, int32_t act,
                                 smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 && act >= 0 && act <= 2 && a_pitch % 4 == 0 && out_pitch % 4 == 0,
               "upconv_ypass: bad arguments");
  SMOS_REQUIRE(conv_a && bias && out && aligned16(conv_a) && aligned16(out) && aligned16(bias) && (!t1 || (aligned16(t1) && H1 > 0)) &&
                   (!t2 || (aligned16(t2) && H2 > 0)), "upconv_ypass: null / unaligned pointer");
  SMOS_REQUIRE(B * Ho * Wo * (C / 4) < kMaxTotal32 && (t1 || t2) && B * 3 * (H1 > H2 ? H1 : H2) < (1LL << 31),
               "upconv_ypass: too many elements for 32-bit indices / no source");
  // an absent source: the other one again, switched off -- every load of the kernel stays unconditional
  YSrc s1{t1 ? t1 : t2, (int)(t1 ? H1 : H2), t1 ? 1.0f : 0.0f}, s2{t2 ? t2 : t1, (int)(t2 ? H2 : H1), t2 ? 1.0f : 0.0f};
  hipLaunchKernelGGL(upconv_ypass, dim3(grid_for(B * Ho * Wo * (C / 4), kBlock, 256 * 32)), dim3(kBlock), 0, (hipStream_t)stream, conv_a,
                     a_pitch, bias, s1, s2, out, out_pitch, (int)B, (int)Ho, (int)Wo, (int)(C / 4), (int)act);
  return check_launch("upconv_ypass");
}

extern "C" int smos_upconv_xy_ok(int64_t Hs, int64_t Ho) { return Hs > 0 && Ho > 0 && 2 * (Hs - 1) < Ho - 1 + (Ho == 1); }

extern "C" int smos_upconv_xy(const float* conv_a, int64_t a_pitch, const float* bias, const float* z1, int64_t H1, int64_t W1,
                              const float* z2, int64_t H2, int64_t W2, float* out, int64_t out_pitch, int64_t B, int64_t Ho, int64_t Wo,
                              int64_t C, int32_t act, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 && act >= 0 && act <= 2 && a_pitch % 4 == 0 && out_pitch % 4 == 0,
               "upconv_xy: bad arguments");
  SMOS_REQUIRE(conv_a && bias && out && aligned16(conv_a) && aligned16(out) && aligned16(bias) && (z1 || z2) &&
                   (!z1 || (aligned16(z1) && H1 > 0 && W1 > 0)) && (!z2 || (aligned16(z2) && H2 > 0 && W2 > 0)),
               "upconv_xy: null / unaligned pointer or no source");
  SMOS_REQUIRE((!z1 || smos_upconv_xy_ok(H1, Ho)) && (!z2 || smos_upconv_xy_ok(H2, Ho)),
               "upconv_xy: a source is taller than half the output (use the x pass + y pass pair)");
  SMOS_REQUIRE(B * Ho * Wo * (C / 4) < kMaxTotal32 && (!z1 || B * H1 * W1 * 9 * C < kMaxTotal32) && (!z2 || B * H2 * W2 * 9 * C < kMaxTotal32),
               "upconv_xy: too many elements for 32-bit indices");
  XYSrc s1{z1, (int)H1, (int)W1, z1 ? 1 : 0}, s2{z2, (int)H2, (int)W2, z2 ? 1 : 0};
  // strip height: long strips amortise the two extra x-pass rows a strip computes before its first output row; short ones
  // give more blocks.  32 rows where that still leaves >= 8 waves per CU, else 16, else 8.  (Measured at the network's
  // geometry, interpolation only: 8 rows 0.218 ms, 16 0.188, 32 0.162, 64 0.166, 128 0.164; the x pass + y pass pair 0.259.)
  const int64_t per_row_blocks = (Wo * (C / 4) + kBlock - 1) / kBlock;
  int strip = 32;
  while (strip > 8 && B * ((Ho + strip - 1) / strip) * per_row_blocks < 512) strip >>= 1;
  const int64_t units = B * ((Ho + strip - 1) / strip) * per_row_blocks;
  hipLaunchKernelGGL(upconv_xy, dim3((unsigned)(units < 256 * 32 ? units : 256 * 32)), dim3(kBlock), 0, (hipStream_t)stream, conv_a, a_pitch,
                     bias, s1, s2, out, out_pitch, (int)B, (int)Ho, (int)Wo, (int)(C / 4), strip, (int)act);
  return check_launch("upconv_xy");
}
