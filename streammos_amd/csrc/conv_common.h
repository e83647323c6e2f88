// Shared by csrc/conv_igemm.hip and csrc/conv_rows.hip: argument block, ring barrier, scheduling fence, half-wave sum.
#pragma once
#include "smos_common.h"
#include "conv_diag.h"

namespace smos {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* x;      // [B, H, W, *] row pitch xp (floats)
  const float4* w;     // operand order [cout tile][stage][k-step / 4][mt][lane][k-step % 4]
  const float* bias;   // [Cout] or null
  const float* res;    // [B, Ho, Wo, *] row pitch rp, or null
  float* out;          // [B, Ho, Wo, *] row pitch op
  float* sums;         // SUMS: [B][hq * xt * 4][Cout] per-(item, row) channel sums of the output, or unused
  int64_t xp, rp, op;
  int B, H, W, Ho, Wo;
  int KH, KW, S, PH, PW;
  int nch;             // Cin / 32
  int nstage;          // KH * KW * nch
  int nct;             // Cout / (32 * MT)
  int hq, xt;          // ceil(Ho / 4), ceil(Wo / 32)
  int n_items;         // B * hq * xt * nct
  float slope;         // activation: max(v, 0) + slope * min(v, 0) -- 1 none, 0 ReLU, 0.01 LeakyReLU
  SMOS_STAMPS_ARG
  int x_bytes;         // B * H * W * xp * 4 (< 2^31: lanes outside the image use offset 2^31)
  int r_bytes, o_bytes, cout;   // B * Ho * Wo * rp * 4, B * Ho * Wo * op * 4, Cout
};

// Ring barrier.  __syncthreads() would also do, but its workgroup fence makes hipcc wait vmcnt(0) -- draining the operand
// prefetch once per stage.  Only LDS traffic has to be ordered here: every wave drains its own LDS queue (ring stores
// landed, fragment reads returned), then the barrier.  The "memory" clobbers keep the compiler from moving ring accesses
// across it.
__device__ __forceinline__ void ring_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (SMOS_CONV_KEEPS(1)) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Scheduling fence: a wave issues in order, and an MFMA that depends on the previous one (same accumulator) cannot issue
// before it has finished (64 cycles) -- so everything that is NOT an MFMA only overlaps with the matrix pipe if it sits
// BETWEEN MFMAs in program order.  The stage body below is cut into MFMA groups (G) and small bookkeeping segments (M);
// the fences keep hipcc from collecting the segments in front of or behind the MFMA block.
#define SMOS_FENCE()                                                                          \
  do {                                                                                        \
    asm volatile("" ::: "memory"); /* IR level: loads and stores stay on their side */        \
    __builtin_amdgcn_sched_barrier(0); /* machine scheduler: nothing crosses, MFMAs included */ \
  } while (0)

// Sum over the 32 lanes of each half wave, delivered in its last lane (31 / 63): four row_shr steps inside the 16-lane DPP
// rows (an inclusive scan by doubling; lanes without a source read 0), then row_bcast:15 carries lane 15 / 47 into the next
// row.  One fixed order, so the sums are run-to-run identical.
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));   // row_shr:1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));   // row_shr:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));   // row_shr:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));   // row_shr:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
  return v;
}

struct ConvTile {      // where a wave's 32-pixel row segment lies (everything scalar; recomputed once per tile)
  int b, y, x0, ct;
  bool valid;
};

}  // namespace smos
