// Declarations of csrc/conv_wino.hip: argument block,
// item coordinates, LDS image constants of the staged input region, the 16-lane row sum.
#pragma once
#include "conv_common.h"

// Diagnostic builds only (tools/ablate_wino.sh): -DSMOS_WINO_ABLATE=<bits> removes one ingredient at a time (1 region
// requests, 2 region stores, 4 weight DMA, 8 barrier, 16 patch reads + transform, 32 A-operand reads, 64 output stores,
// 128 the explicit vmcnt wait) to time what is left; results are wrong.  The shipped library is built without it.
#ifdef SMOS_WINO_ABLATE
#define WINO_AB(bit) ((SMOS_WINO_ABLATE) & (bit))
#else
#define WINO_AB(bit) 0
#endif

namespace smos {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoArgs {
  const float* x;      // [B, H, W, *] row pitch xp (floats)
  const float4* w;     // [cout tile][chunk][k-step][mb][xi][lane = q * 16 + m][nu]  (ops.conv_wino_prepare)
  const float* bias;   // [Cout] or null
  const float* res;    // [B, H, W, *] row pitch rp, or null
  float* out;          // [B, H, W, *] row pitch op
  float* sums;         // SUMS: [B][yb * xb * 4][Cout] per-(item, wave) channel sums of the output
  int64_t xp, rp, op;
  int B, H, W;
  int nchunk;          // Cin / 16
  int nct;             // Cout / (16 * MB)
  int yb, xb;          // ceil(H / 8), ceil(W / 32)
  int n_items;         // B * yb * xb * nct
  float slope;         // activation: max(v, 0) + slope * min(v, 0)
  int x_bytes, r_bytes, o_bytes, cout;
  int group;           // item order: 0 = contiguous item range per block, 1 = nct consecutive blocks share a region range (conv_wino.hip)
};

constexpr int kWPP = 17;                          // words per staged pixel: 16 channels + 1 (odd pitch)
constexpr int kWRegW = 34, kWRegH = 10;           // staged region: (8 + 2) rows x (32 + 2) columns
constexpr int kWRegPix = kWRegW * kWRegH;         // 340 pixels = 1360 float4 = 5.3 per thread
constexpr int kWInWords = kWRegPix * kWPP;        // 5780 words (23 120 B) per buffer

// sum over the 16 lanes of a DPP row, delivered in its last lane (tx = 15); fixed order -> run-to-run identical
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
  return v;
}

struct WinoItem {
  int b, y0, x0, ct;
};

}  // namespace smos
