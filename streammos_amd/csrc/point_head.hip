// Fused point head for gfx950: CatFusion + PredBranch (networks/backbone.py:387-413, 188-196; models/StreamMOS.py:107-113)
//     rows [P, 192] -> 1x1 192->96 + ReLU -> 1x1 96->64 + ReLU -> 1x1 64->M3 (+bias)        (BatchNorm folded)
// as ONE kernel on the matrix cores.  The three layers are chained in the transposed form of pointnet_scatter
// (point_fused.hip): C = W * X with the output channel on the MFMA row and the point on the column, so that a layer's
// 32x32 result tiles (point on the lane, channels in the accumulator registers) ARE the B operands of the next layer --
// the 96- and 64-channel intermediates (245 + 164 MB at the validation shape) never leave the registers.
//   A operands (weights): all three layers resident in LDS (72 + 24 + 8 KB) in MFMA operand order, prepared on the host;
//   layer 1 B operand: lane (p, h) holds channels h*96 .. h*96+95 of point p, streamed from its row in four K-quarters;
//   layer 2 / 3 walk their input channels in accumulator order: register r of tile mt, lane half h = channel
//   32 mt + 8 (r >> 2) + 4 h + (r & 3).
// Output: logits [B, M3, N] (the reference's (B, 3, N, 1) layout), 32 contiguous floats per channel and tile.
#include "smos_common.h"

namespace smos {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// Diagnostic builds only (tools/ablate_head.sh): -DSMOS_HEAD_ABLATE=<bits> removes 1 the row loads, 2 the layer-1 MFMAs, 4 layers
// 2 / 3, 8 the logit stores to time what is left; results are wrong.  The shipped library is built without it.
#ifdef SMOS_HEAD_ABLATE
#define HEAD_AB(bit) ((SMOS_HEAD_ABLATE) & (bit))
#else
#define HEAD_AB(bit) 0
#endif
constexpr int kHeadBlock = 512;
constexpr int kK1 = 192, kM1 = 96, kM2 = 64;
constexpr int kS1 = kK1 / 2, kS2 = kM1 / 2, kS3 = kM2 / 2;           // k-steps (two k per MFMA)
constexpr int kA1 = (kM1 / 32) * kS1 * 64, kA2 = (kM2 / 32) * kS2 * 64, kA3 = kS3 * 64;   // floats
constexpr int kHeadLds = kA1 + kA2 + kA3 + kM1 + kM2 + 32;            // + biases

struct HeadArgs {
  const float* rows;     // [P, *] row pitch rp
  const float* wprep;    // kHeadLds floats: A1 | A2 | A3 | b1 | b2 | b3 (padded to 32)
  float* out;            // [B, M3, N]
  const int32_t* n_live; // device: points [*n_live, N) of every sample are the padding tail of the scan (null: none)
  int64_t rp;
  int B, N, M3;
};

__global__ __launch_bounds__(kHeadBlock) void point_head(HeadArgs a) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < kHeadLds; i += kHeadBlock) lds[i] = a.wprep[i];
  __syncthreads();
  const float* A1 = lds;
  const float* A2 = lds + kA1;
  const float* A3 = A2 + kA2;
  const float* B1 = A3 + kA3;
  const float* B2 = B1 + kM1;
  const float* B3 = B2 + kM2;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int col = lane & 31, hh = lane >> 5;
  constexpr int kWaves = kHeadBlock / 64;
  const int n_live = a.n_live ? min(max(*a.n_live, 0), a.N) : a.N;
  // The padding tail (datasets/data_StreamMOS.py:568-571 pads every scan to frame_point_num with points at -1000 that
  // val_StreamMOS.py:113 cuts off again): its logits are never read; they are written as zeros, by all threads alike.  The
  // waves then share the LIVE tiles only -- dealing out all tiles round-robin left a wave 5 to 7 live ones of its ~10.
  const int live_tiles = (n_live + 31) / 32, tail0 = live_tiles * 32;
  if (tail0 < a.N) {
    const int64_t per = (int64_t)a.M3 * (a.N - tail0), total = (int64_t)a.B * per;
    for (int64_t i = (int64_t)blockIdx.x * kHeadBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kHeadBlock) {
      const int b = (int)(i / per);
      const int64_t r = i - (int64_t)b * per;
      const int ch = (int)(r / (a.N - tail0));
      a.out[((int64_t)b * a.M3 + ch) * a.N + tail0 + (r - (int64_t)ch * (a.N - tail0))] = 0.0f;
    }
  }
  for (int lt = blockIdx.x * kWaves + wave; lt < a.B * live_tiles; lt += gridDim.x * kWaves) {
    const int b = lt / live_tiles;
    const int n = (lt - b * live_tiles) * 32 + col;
    const bool valid = n < a.N;
    const float4* src = reinterpret_cast<const float4*>(a.rows + ((int64_t)b * a.N + (valid ? n : 0)) * a.rp + hh * kS1);

    // ---- layer 1: 192 -> 96, K streamed in four quarters
    constexpr int kQ = 4, kQSteps = kS1 / kQ;
    float4 cur[kQSteps / 4], nxt[kQSteps / 4];
#pragma unroll
    for (int j = 0; j < kQSteps / 4; ++j) cur[j] = HEAD_AB(1) ? make_float4((float)lane, 1.f, 2.f, (float)j) : src[j];
    f32x16 c1[kM1 / 32];
#pragma unroll
    for (int mt = 0; mt < kM1 / 32; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bias = *reinterpret_cast<const float4*>(B1 + mt * 32 + 8 * g + 4 * hh);
        c1[mt][4 * g] = bias.x; c1[mt][4 * g + 1] = bias.y; c1[mt][4 * g + 2] = bias.z; c1[mt][4 * g + 3] = bias.w;
      }
#pragma unroll 1
    for (int qt = 0; qt < kQ; ++qt) {
      if (qt + 1 < kQ) {
#pragma unroll
        for (int j = 0; j < kQSteps / 4; ++j)
          nxt[j] = HEAD_AB(1) ? make_float4((float)lane, (float)qt, 2.f, (float)j) : src[(qt + 1) * (kQSteps / 4) + j];
      }
      const float* wq = A1 + (qt * kQSteps) * 64 + lane;
#pragma unroll
      for (int s = 0; s < kQSteps; ++s) {
        const float4 v = cur[s >> 2];
        const float x = (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
#pragma unroll
        for (int mt = 0; mt < kM1 / 32; ++mt) {
          if (HEAD_AB(2)) c1[mt][s & 15] += wq[(mt * kS1 + s) * 64] * x;
          else c1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[(mt * kS1 + s) * 64], x, c1[mt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < kQSteps / 4; ++j) cur[j] = nxt[j];
    }

    // ---- layer 2: 96 -> 64 on relu(c1), input channels in accumulator order
    f32x16 c2[kM2 / 32];
    int a2_off = lane, a3_off = lane;
#pragma unroll
    for (int mt = 0; mt < kM2 / 32; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bias = *reinterpret_cast<const float4*>(B2 + mt * 32 + 8 * g + 4 * hh);
        c2[mt][4 * g] = bias.x; c2[mt][4 * g + 1] = bias.y; c2[mt][4 * g + 2] = bias.z; c2[mt][4 * g + 3] = bias.w;
      }
#pragma unroll
    for (int s = 0; s < (HEAD_AB(4) ? 2 : kS2); ++s) {
      // every 8 steps the LDS offset is re-materialised behind the accumulators: keeps the scheduler from hoisting all
      // 96 weight reads of the layer to its top (which spills)
      if (s % 8 == 0) asm volatile("" : "+v"(a2_off), "+v"(c2[0]), "+v"(c2[1]));
      const float x = fmaxf(c1[s >> 4][s & 15], 0.0f);
#pragma unroll
      for (int mt = 0; mt < kM2 / 32; ++mt)
        c2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[(mt * kS2 + s) * 64 + a2_off], x, c2[mt], 0, 0, 0);
    }

    // ---- layer 3: 64 -> M3 (<= 32 rows, zero padded) on relu(c2)
    f32x16 c3;
#pragma unroll
    for (int r = 0; r < 16; ++r) c3[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < (HEAD_AB(4) ? 2 : kS3); ++s) {
      if (s % 8 == 0) asm volatile("" : "+v"(a3_off), "+v"(c3));
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(A3[s * 64 + a3_off], fmaxf(c2[s >> 4][s & 15], 0.0f), c3, 0, 0, 0);
    }

    // row i of the result = 8 (r >> 2) + 4 h + (r & 3): lane half h, register r
    if (valid && !(HEAD_AB(8) && c3[0] != 12345.678f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = 8 * (r >> 2) + 4 * hh + (r & 3);
        if (ch < a.M3) a.out[((int64_t)b * a.M3 + ch) * a.N + n] = n < n_live ? c3[r] + B3[ch] : 0.0f;   // tail inside a live tile
      }
    }
  }
}

}  // namespace smos

using namespace smos;

extern "C" int64_t smos_point_head_weight_floats(void) { return kHeadLds; }

extern "C" int smos_point_head_live(const float* rows, int64_t row_pitch, const float* wprep, float* out, int64_t B, int64_t N,
                                    int64_t K1, int64_t M1, int64_t M2, int64_t M3, const int32_t* n_live, smos_stream_t stream);

extern "C" int smos_point_head(const float* rows, int64_t row_pitch, const float* wprep, float* out, int64_t B, int64_t N,
                               int64_t K1, int64_t M1, int64_t M2, int64_t M3, smos_stream_t stream) {
  return smos_point_head_live(rows, row_pitch, wprep, out, B, N, K1, M1, M2, M3, nullptr, stream);
}

// n_live (device int32, may be null): the first *n_live points of every sample are real, the rest is the scan's padding tail,
// whose logits are written as zeros without being computed.
extern "C" int smos_point_head_live(const float* rows, int64_t row_pitch, const float* wprep, float* out, int64_t B, int64_t N,
                                    int64_t K1, int64_t M1, int64_t M2, int64_t M3, const int32_t* n_live, smos_stream_t stream) {
  SMOS_REQUIRE(K1 == kK1 && M1 == kM1 && M2 == kM2 && M3 >= 1 && M3 <= 32, "point_head: built for 192 -> 96 -> 64 -> (<=32)");
  SMOS_REQUIRE(B > 0 && N > 0 && row_pitch >= K1 && row_pitch % 4 == 0 && B * ((N + 31) / 32) < (1LL << 31), "point_head: bad sizes");
  SMOS_REQUIRE(rows && wprep && out && (reinterpret_cast<uintptr_t>(rows) & 15) == 0, "point_head: null / unaligned pointer");
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&point_head), kHeadLds * sizeof(float), 0, &ks, "point_head")) return rc;
  const int cus = ks.cus;
  HeadArgs a;
  a.rows = rows; a.wprep = wprep; a.out = out; a.n_live = n_live; a.rp = row_pitch; a.B = (int)B; a.N = (int)N; a.M3 = (int)M3;
  const int64_t tiles = B * ((N + 31) / 32);
  const int64_t want = (tiles + 7) / 8;
  hipLaunchKernelGGL(point_head, dim3((unsigned)(want < cus ? want : cus)), dim3(kHeadBlock), kHeadLds * sizeof(float),
                     (hipStream_t)stream, a);
  return check_launch("point_head");
}
