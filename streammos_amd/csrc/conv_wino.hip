// Stride-1 3x3 convolution (channels-last fp32, "same" padding) in the Winograd F(2x2, 3x3) form on the matrix cores of
// gfx950, with the bias + activation (+ residual, + channel sums) epilogue fused -- the 36 stride-1 3x3 layers of the
// network (networks/backbone.py:136-159 BasicBlock, multi_view_encoder.py:478-497 the 3x3 of an Unbalance block, :446-447
// conv_1 / conv_2) do 4 multiply-adds per output and channel pair instead of 9:
//
//     Y = A^T [ sum_cin (G g G^T) . (B^T d B) ] A        d: 4x4 input patch, Y: 2x2 outputs, "." elementwise
//
// The weights U = G g G^T are transformed in float64 on the host and rounded once (ops.conv_wino_prepare); B^T d B and
// A^T M A consist of additions only (the +-1 / 0 matrices of F(2,3)), so the arithmetic stays plain fp32:
// tools/winograd_numerics.py measures 2.2e-7 .. 5.4e-7 per layer against float64 (direct fp32: 3.1e-7 .. 3.8e-7) and
// 1.2e-6 of the logit range end to end against the reference's golden outputs (bar 2e-5).
//
// Mapping.  The 16 transform positions (xi, nu) are 16 independent GEMMs  M[xi nu][cout][tile] = U[xi nu][cout][cin] *
// V[xi nu][cin][tile]  with K = Cin, run on v_mfma_f32_16x16x4_f32 (exact f32; 16 couts x 16 tiles x 4 cins per
// instruction, 4 accumulator registers): all 16 accumulators of a (16-cout, 16-tile) pair stay in registers for the
// whole K loop -- 64 * MB registers for the 16 * MB couts a wave computes -- and the output transform happens in the
// lane (a lane holds 4 consecutive couts of ONE tile for every (xi, nu)), followed by the epilogue and 16-byte stores.
//
//   work item  = 8 output rows x 32 columns (4 x 16 tiles) x 16 * MB output channels; one block (4 waves) per item, wave w
//                = tile row w (16 tiles side by side: the MFMA column), blocks persistent over a contiguous item range
//                (cout tile fastest, XCD-aware order as in conv_igemm).
//   chunk      = 16 input channels = 4 k-steps; lane (q = lane >> 4, tx = lane & 15) supplies cin 4 q + i of the chunk at
//                k-step i (the weights are packed to match), so its four k-steps read the four channels of ONE float4.
//   B operand  = transformed activations.  The block stages the chunk's (8 + 2) x (32 + 2) input region ONCE from global
//                memory (fully coalesced 16-byte buffer loads, zero padding = out-of-range offsets), transposed on the way
//                into LDS: pixel pitch 17 words, channel 4 q + i at word 4 i + q -- a k-step's 16 patch reads (ds_read_b32,
//                immediate offsets) are then bank-conflict free (bank = 2 tx + q + const).  Per k-step a lane reads its 4x4
//                patch, runs the 32 additions of B^T d B and owns the 16 B operands.  Two region buffers: chunk g + 1 is
//                requested at the start of chunk g and written during its last k-step.
//   A operand  = U, streamed through a three-slot LDS ring by LDS-DMA (no registers on the way), one k-step (16 x 16 MB x 4
//                floats) per slot, requested two k-steps ahead; a lane reads 4 consecutive (nu) operands per ds_read_b128.
//                One barrier per k-step.
//   bytes      = per cin: 64 B x 16 MB of weights + ~85 B x 16 of activations for 16 x 16 MB x 64 MACs: 13 B/clk/CU at
//                MB = 2 with the matrix pipe saturated -- inside what an XCD's L2 serves a CU (~29 B/clk).
#include <stdlib.h>

#include "conv_wino_common.h"

namespace smos {

template <int MB, bool RES, bool SUMS>
__global__ __launch_bounds__(256, 2) void conv_wino(WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#include "conv_wino_body.inc"
}

}  // namespace smos

using namespace smos;

template <int MB, bool RES, bool SUMS>
static int launch_wino(const WinoArgs& a, hipStream_t s) {
  const size_t lds = (size_t)2 * kWInWords * sizeof(float) + (size_t)3 * 256 * MB * sizeof(float4);
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&conv_wino<MB, RES, SUMS>), lds, 256, &ks, "conv_wino_cl")) return rc;
  static const int want_per_cu = [] {
    int v = 2;
    if (const char* e = getenv("SMOS_WINO_BLOCKS_PER_CU")) {
      const int n = atoi(e);
      if (n >= 1 && n <= 8) v = n;
    }
    return v;
  }();
  const int per_cu = ks.per_cu < want_per_cu ? ks.per_cu : want_per_cu;
  const int64_t cap = conv_grid_cap((int64_t)ks.cus * per_cu);
  unsigned grid = (unsigned)(a.n_items < cap ? a.n_items : cap);
  static const bool want_group = [] {
    const char* e = getenv("SMOS_WINO_CT_GROUP");
    return !(e && e[0] == '0');
  }();
  WinoArgs b = a;
  b.group = 0;
  if (want_group && a.nct > 1 && grid >= (unsigned)a.nct) {
    grid -= grid % (unsigned)a.nct;          // n_items is a multiple of nct, so an uncapped grid already is
    b.group = 1;
  }
  hipLaunchKernelGGL((conv_wino<MB, RES, SUMS>), dim3(grid), dim3(256), lds, s, b);
  return check_launch("conv_wino_cl");
}

extern "C" int64_t smos_conv_wino_sum_chunks(int64_t H, int64_t W) { return ((H + 7) / 8) * ((W + 31) / 32) * 4; }

// Stride-1 3x3 "same" convolution in the Winograd F(2x2, 3x3) form.  wprep = ops.conv_wino_prepare(w, mb) (the float64
// G g G^T of the folded weights in operand order); everything else as smos_conv_cl.  mb in {1, 2}: 16 * mb output channels
// per block.  Replaces the same reference layers as smos_conv_cl where the kernel is 3x3 and the stride 1.
extern "C" int smos_conv_wino_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, const float* res,
                                 int64_t res_pitch, float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t Cin,
                                 int64_t Cout, int32_t mb, int32_t act, float* chan_sums, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && (mb == 1 || mb == 2) && Cout % (16 * mb) == 0 &&
                   act >= 0 && act <= 2, "conv_wino_cl: Cin must be a multiple of 16 and Cout of 16 * mb (mb in {1, 2})");
  SMOS_REQUIRE(Cout <= 2048, "conv_wino_cl: more than 2048 output channels");
  SMOS_REQUIRE(x && wprep && out && x_pitch >= Cin && out_pitch >= Cout && x_pitch % 4 == 0 && out_pitch % 4 == 0 &&
                   (!res || (res_pitch >= Cout && res_pitch % 4 == 0)), "conv_wino_cl: null pointer / bad pitch");
  SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(wprep) | reinterpret_cast<uintptr_t>(chan_sums)) & 15) == 0,
               "conv_wino_cl: pointers must be 16-byte aligned");
  SMOS_REQUIRE(B * H * W * x_pitch * 4 < (1LL << 31) && B * H * W * out_pitch * 4 < (1LL << 31) &&
                   (!res || B * H * W * res_pitch * 4 < (1LL << 31)), "conv_wino_cl: a tensor larger than 2 GiB (32-bit buffer offsets)");
  const int64_t yb = (H + 7) / 8, xb = (W + 31) / 32, nct = Cout / (16 * mb);
  SMOS_REQUIRE(B * yb * xb * nct < (1LL << 30) && nct * (Cin / 16) < (1LL << 24), "conv_wino_cl: too many tiles");
  SMOS_REQUIRE(!(chan_sums && res), "conv_wino_cl: channel sums need res == NULL");
  WinoArgs a;
  a.x = x; a.w = reinterpret_cast<const float4*>(wprep); a.bias = bias; a.res = res; a.out = out; a.sums = chan_sums;
  a.xp = x_pitch; a.rp = res_pitch; a.op = out_pitch;
  a.B = (int)B; a.H = (int)H; a.W = (int)W;
  a.nchunk = (int)(Cin / 16); a.nct = (int)nct; a.yb = (int)yb; a.xb = (int)xb; a.n_items = (int)(B * yb * xb * nct);
  a.slope = act == 0 ? 1.0f : act == 1 ? 0.0f : 0.01f;
  a.x_bytes = (int)(B * H * W * x_pitch * 4);
  a.r_bytes = res ? (int)(B * H * W * res_pitch * 4) : 0;
  a.o_bytes = (int)(B * H * W * out_pitch * 4);
  a.cout = (int)Cout;
  hipStream_t s = (hipStream_t)stream;
  if (chan_sums) return mb == 1 ? launch_wino<1, false, true>(a, s) : launch_wino<2, false, true>(a, s);
  if (res) return mb == 1 ? launch_wino<1, true, false>(a, s) : launch_wino<2, true, false>(a, s);
  return mb == 1 ? launch_wino<1, false, false>(a, s) : launch_wino<2, false, false>(a, s);
}
