// One foreign call per residual block: the launches of a BasicBlock (networks/backbone.py:136-159) enqueued back to back by the
// entry points that already exist, so the host pays one call's marshalling instead of three (a launch through the Python
// binding costs 6.8 us, of which hipLaunchKernel is 3.5: tools/ubench_hostcall.py).  No kernel of its own: the results are those
// of the separate calls, bit for bit.
#include "smos_common.h"

extern "C" int64_t smos_basic_block_ws_floats(int64_t B, int64_t H, int64_t W, int64_t C) {
  return B * C * (smos_conv_wino_sum_chunks(H, W) + 1);
}

extern "C" int smos_basic_block_cl(const float* x, int64_t x_pitch, const float* u1, const float* b1, const float* u2,
                                   const float* b2, const float* gw1, const float* gb1, const float* gw2, const float* gb2,
                                   int64_t Cr, float* y, int64_t y_pitch, float* out, int64_t out_pitch, float* ws, int64_t B,
                                   int64_t H, int64_t W, int64_t C, int32_t mb, smos_stream_t stream) {
  SMOS_REQUIRE(x && u1 && u2 && y && out && y != out && x != y, "basic_block_cl: null pointer, or y aliases x / out");
  int rc = smos_conv_wino_cl(x, x_pitch, u1, b1, nullptr, 0, y, y_pitch, B, H, W, C, C, mb, /*relu*/ 1, nullptr, stream);
  if (rc) return rc;
  if (!gw1)   // plain block: out = relu(conv(y) + b2 + x)
    return smos_conv_wino_cl(y, y_pitch, u2, b2, x, x_pitch, out, out_pitch, B, H, W, C, C, mb, 1, nullptr, stream);
  // ChannelAtt block: the second conv leaves its per-chunk channel sums in ws, the gate kernel turns out into
  // relu((out + b2) * gate + x) in place
  SMOS_REQUIRE(gb1 && gw2 && gb2 && b2 && ws && Cr > 0 && x != out, "basic_block_cl: gate parameters / scratch missing, or out aliases x");
  const int64_t chunks = smos_conv_wino_sum_chunks(H, W);
  rc = smos_conv_wino_cl(y, y_pitch, u2, nullptr, nullptr, 0, out, out_pitch, B, H, W, C, C, mb, /*none*/ 0, ws, stream);
  if (rc) return rc;
  return smos_channel_gate_apply_cl(out, out_pitch, b2, gw1, gb1, gw2, gb2, x, x_pitch, out, out_pitch, ws, chunks,
                                    ws + B * chunks * C, B, C, Cr, H * W, stream);
}

// Unbalance_BasicBlock.forward (networks/multi_view_encoder.py:478-497): the k x 3 and 3 x k branches into the two halves of
// `both` (ReLU), then the 3x3 over their concatenation + x, ReLU.  Three launches (smos_conv_wino1d_cl x 2, smos_conv_wino_cl).
extern "C" int smos_unbalance_block_cl(const float* x, int64_t x_pitch, const float* ua, const float* ba, int64_t kha, int64_t kwa,
                                       const float* ub, const float* bb, int64_t khb, int64_t kwb, const float* uc, const float* bc,
                                       float* both, int64_t both_pitch, float* out, int64_t out_pitch, int64_t B, int64_t H,
                                       int64_t W, int64_t C, int32_t mb, smos_stream_t stream) {
  SMOS_REQUIRE(x && ua && ub && uc && both && out && both != out && x != both && x != out && both_pitch >= 2 * C,
               "unbalance_block_cl: null pointer, aliasing maps, or `both` narrower than 2 C channels");
  int rc = smos_conv_wino1d_cl(x, x_pitch, ua, ba, both, both_pitch, B, H, W, C, C, kha, kwa, mb, 1, stream);
  if (rc) return rc;
  rc = smos_conv_wino1d_cl(x, x_pitch, ub, bb, both + C, both_pitch, B, H, W, C, C, khb, kwb, mb, 1, stream);
  if (rc) return rc;
  return smos_conv_wino_cl(both, both_pitch, uc, bc, x, x_pitch, out, out_pitch, B, H, W, 2 * C, C, mb, 1, nullptr, stream);
}
