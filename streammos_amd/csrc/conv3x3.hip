// 3x3 / stride 1 / pad 1 convolution with Cin = Cout = C in {32, 64}, channels-last, with the bias + activation
// (+ residual) epilogue fused -- the second half of the network's BasicBlocks (networks/backbone.py:87-102) and the
// first conv of each (conv -> BN -> ReLU), for gfx950.
//
// Implicit GEMM in the transposed form used by the other matrix-core kernels of this library (point_fused.hip):
//     C[cout][pixel] = sum_{tap, cin} W[cout][cin][tap] * X[pixel + tap][cin]
// output channel on the MFMA row, 32 consecutive pixels of one image row on the column (v_mfma_f32_32x32x2_f32, exact f32).
//   A operand: ALL weights of the layer resident in LDS in operand order [tap][mt][k-step][lane] (36 KB for C = 32,
//              144 KB for C = 64), loaded once per block; blocks are persistent.
//   B operand: lane (p, h) holds channels h*C/2 .. of pixel (y + ky - 1, x0 + p + kx - 1): C/8 float4 loads per tap, zeros
//              outside the image; the next tap's loads are in flight while the current tap feeds the matrix core.
// Epilogue in registers: out = act(acc + bias [+ residual]), 16-byte stores (lane (p, h), register r <-> channel
// 32 mt + 8 (r >> 2) + 4 h + (r & 3)).
#include "smos_common.h"

namespace smos {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Conv3Args {
  const float* x;      // [B, H, W, *] row pitch xp
  const float* wprep;  // 9 * (C/32) * (C/2) * 64 floats in operand order
  const float* bias;   // [C] or null
  const float* res;    // [B, H, W, *] row pitch rp, or null
  float* out;          // [B, H, W, *] row pitch op
  int64_t xp, rp, op;
  int B, H, W, act;    // act: 0 none, 1 ReLU, 2 LeakyReLU(0.01)
};

template <int C, int kThreads>
__global__ __launch_bounds__(kThreads) void conv3x3_cl(Conv3Args a) {
  constexpr int kK = C / 2, kMt = C / 32, kV = kK / 4;     // k-steps per tap, 32-channel output blocks, float4 per lane and tap
  extern __shared__ float lds_w[];
  for (int i = threadIdx.x; i < 9 * kMt * kK * 64; i += kThreads) lds_w[i] = a.wprep[i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int col = lane & 31, hh = lane >> 5;
  constexpr int kWaves = kThreads / 64;
  const int xblocks = a.W / 32;
  const int n_tiles = a.B * a.H * xblocks;

  auto load_tap = [&](int b, int y, int x0, int tap, float4 (&dst)[kV]) {
    const int yy = y + tap / 3 - 1, xx = x0 + col + tap % 3 - 1;
    const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    const float4* p = reinterpret_cast<const float4*>(a.x + (((int64_t)b * a.H + (ok ? yy : 0)) * a.W + (ok ? xx : 0)) * a.xp + hh * kK);
#pragma unroll
    for (int j = 0; j < kV; ++j) dst[j] = ok ? p[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  for (int tile = blockIdx.x * kWaves + wave; tile < n_tiles; tile += gridDim.x * kWaves) {
    const int xb = tile % xblocks;
    const int t2 = tile / xblocks;
    const int y = t2 % a.H, b = t2 / a.H;
    const int x0 = xb * 32;
    f32x16 acc[kMt];
#pragma unroll
    for (int mt = 0; mt < kMt; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    float4 cur[kV], nxt[kV];
    load_tap(b, y, x0, 0, cur);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) load_tap(b, y, x0, tap + 1, nxt);
      const float* wq = lds_w + (tap * kMt * kK) * 64 + lane;
#pragma unroll
      for (int s = 0; s < kK; ++s) {
        const float4 v = cur[s >> 2];
        const float xv = (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
#pragma unroll
        for (int mt = 0; mt < kMt; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[(mt * kK + s) * 64], xv, acc[mt], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < kV; ++j) cur[j] = nxt[j];
    }
    const int64_t pix = ((int64_t)b * a.H + y) * a.W + x0 + col;
    float* dst = a.out + pix * a.op + 4 * hh;
    const float* rsd = a.res ? a.res + pix * a.rp + 4 * hh : nullptr;
#pragma unroll
    for (int mt = 0; mt < kMt; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = mt * 32 + 8 * g;
        float4 o = make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
        if (a.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(a.bias + ch + 4 * hh);
          o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
        }
        if (rsd) {
          const float4 rv = *reinterpret_cast<const float4*>(rsd + ch);
          o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        if (a.act == 1) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        } else if (a.act == 2) {
          o.x = o.x > 0.f ? o.x : 0.01f * o.x; o.y = o.y > 0.f ? o.y : 0.01f * o.y;
          o.z = o.z > 0.f ? o.z : 0.01f * o.z; o.w = o.w > 0.f ? o.w : 0.01f * o.w;
        }
        *reinterpret_cast<float4*>(dst + ch) = o;
      }
  }
}

}  // namespace smos

using namespace smos;

template <int C, int kThreads>
static int launch_conv3x3(const Conv3Args& a, int per_cu, hipStream_t s, const char* what) {
  const size_t lds = (size_t)9 * (C / 32) * (C / 2) * 64 * sizeof(float);
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&conv3x3_cl<C, kThreads>), lds, 0, &ks, what)) return rc;
  const int cus = ks.cus;
  const int64_t tiles = (int64_t)a.B * a.H * (a.W / 32);
  const int64_t want = (tiles + kThreads / 64 - 1) / (kThreads / 64);
  const int64_t cap = (int64_t)cus * per_cu;
  hipLaunchKernelGGL((conv3x3_cl<C, kThreads>), dim3((unsigned)(want < cap ? want : cap)), dim3(kThreads), lds, s, a);
  return check_launch(what);
}

extern "C" int64_t smos_conv3x3_weight_floats(int64_t C) { return (C == 32 || C == 64) ? 9 * (C / 32) * (C / 2) * 64 : -1; }

extern "C" int smos_conv3x3_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, const float* res,
                               int64_t res_pitch, float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t C,
                               int32_t act, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && W % 32 == 0 && (C == 32 || C == 64) && act >= 0 && act <= 2,
               "conv3x3_cl: built for C in {32, 64} and W a multiple of 32");
  SMOS_REQUIRE(x && wprep && out && x_pitch >= C && out_pitch >= C && x_pitch % 4 == 0 && out_pitch % 4 == 0 &&
                   (!res || (res_pitch >= C && res_pitch % 4 == 0)), "conv3x3_cl: null pointer / bad pitch");
  SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(bias)) & 15) == 0, "conv3x3_cl: pointers must be 16-byte aligned");
  SMOS_REQUIRE(B * H * (W / 32) < (1LL << 31), "conv3x3_cl: too many tiles");
  Conv3Args a;
  a.x = x; a.wprep = wprep; a.bias = bias; a.res = res; a.out = out;
  a.xp = x_pitch; a.rp = res_pitch; a.op = out_pitch;
  a.B = (int)B; a.H = (int)H; a.W = (int)W; a.act = act;
  if (C == 32) return launch_conv3x3<32, 256>(a, 4, (hipStream_t)stream, "conv3x3_cl<32>");
  return launch_conv3x3<64, 512>(a, 1, (hipStream_t)stream, "conv3x3_cl<64>");
}
