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
//                blocks are persistent and walk the items in a fixed order, cout tile fastest.
//   stage      = 32 input channels of one tap (16 k-steps): 16 * MT MFMAs per wave.  The K loop of a tile is a sequence
//                of KH * KW * Cin / 32 stages, and the stages of consecutive tiles form ONE stream: the operand prefetch
//                runs two stages ahead straight across tile boundaries, so a wave's pipeline never drains.
//   B operand  = activations, read directly from global memory (L1 / L2 absorb the tap re-reads): lane (p, h) loads four
//                float4 = channels 8 j + 4 h + (0..3) of the stage's 32, zeros outside the image; three named register
//                sets rotate (computing / landed / in flight).
//   A operand  = weights, STREAMED through a three-slot LDS ring (4 * MT KB per stage) instead of being resident: the
//                block's 256 threads fetch the slice of stage g + 2 at the top of stage g (one float4 each per MT), park it
//                in registers during the stage's MFMAs and store it to the ring afterwards; one barrier per stage.  A lane
//                reads its fragment as float4 = four consecutive k-steps (host-side operand order, ops.conv_prepare).
//                12 * MT KB of LDS per block: several blocks per CU coexist with the other HIP stream's kernels (the first
//                version, csrc/conv3x3.hip, kept 144 KB resident at C = 64 and was crowded out of the pipeline).
//   epilogue   = out = act(acc + bias [+ residual]) from the accumulators, 16-byte stores (lane (p, h), register r <->
//                channel 32 mt + 8 (r >> 2) + 4 h + (r & 3)); inputs, residual and output may be channel slices of
//                wider channels-last buffers (row pitches).
#include <stdlib.h>

#include "smos_common.h"

namespace smos {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* x;      // [B, H, W, *] row pitch xp (floats)
  const float4* w;     // operand order [cout tile][stage][k-step / 4][mt][lane][k-step % 4]
  const float* bias;   // [Cout] or null
  const float* res;    // [B, Ho, Wo, *] row pitch rp, or null
  float* out;          // [B, Ho, Wo, *] row pitch op
  int64_t xp, rp, op;
  int B, H, W, Ho, Wo;
  int KH, KW, S, PH, PW;
  int nch;             // Cin / 32
  int nstage;          // KH * KW * nch
  int nct;             // Cout / (32 * MT)
  int hq, xt;          // ceil(Ho / 4), ceil(Wo / 32)
  int n_items;         // B * hq * xt * nct
  float slope;         // activation: max(v, 0) + slope * min(v, 0) -- 1 none, 0 ReLU, 0.01 LeakyReLU
  int x_bytes;         // B * H * W * xp * 4 (< 2^31: lanes outside the image use offset 2^31)
  int r_bytes, o_bytes, cout_bytes;   // B * Ho * Wo * rp * 4, B * Ho * Wo * op * 4, Cout * 4
};


// Ring barrier.  __syncthreads() would also do, but its workgroup fence makes hipcc wait vmcnt(0) -- draining the operand
// prefetch of the next two stages once per stage.  Only LDS traffic has to be ordered here: every wave drains its own LDS
// queue (ring stores landed, fragment reads returned), then the barrier.  The "memory" clobbers keep the compiler from
// moving ring accesses across it.
__device__ __forceinline__ void ring_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
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

struct ConvCursor {    // a position in the block's stream of stages: item (= 4 tiles) and tap / channel chunk inside it
  int it, s, dy, dx, ch;
  int ct, xt, yq, b;
};

template <int MT>
__global__ __launch_bounds__(256, 2) void conv_igemm(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 ring[];     // 3 slots x 256 * MT float4
  constexpr int kSlot = 256 * MT;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  // a block owns a contiguous range of items, so that the cursors advance by carries instead of divisions
  const int per_block = (a.n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int first = (int)blockIdx.x * per_block;
  const int iters = a.n_items - first < per_block ? a.n_items - first : per_block;
  const int total = iters * a.nstage;
  if (total <= 0) return;

  auto cursor_at_first = [&]() {
    ConvCursor c;
    c.it = c.s = c.dy = c.dx = c.ch = 0;
    c.ct = first % a.nct;
    int u = first / a.nct;
    c.xt = u % a.xt;
    u /= a.xt;
    c.yq = u % a.hq;
    c.b = u / a.hq;
    return c;
  };
  // one stage further: channel chunk fastest, then the tap column, the tap row, then the next item (cout tile fastest).
  // Selects only -- the stage body stays one basic block.
  auto advance = [&](ConvCursor& c) {
    const bool ch_wrap = c.ch + 1 == a.nch;
    c.ch = ch_wrap ? 0 : c.ch + 1;
    const bool dx_wrap = ch_wrap && c.dx + 1 == a.KW;
    c.dx = dx_wrap ? 0 : (ch_wrap ? c.dx + 1 : c.dx);
    c.dy = dx_wrap ? c.dy + 1 : c.dy;
    const bool tile_wrap = c.s + 1 == a.nstage;
    c.s = tile_wrap ? 0 : c.s + 1;
    c.dy = tile_wrap ? 0 : c.dy;
    c.it += tile_wrap ? 1 : 0;
    const bool ct_wrap = tile_wrap && c.ct + 1 == a.nct;
    c.ct = ct_wrap ? 0 : (tile_wrap ? c.ct + 1 : c.ct);
    const bool xt_wrap = ct_wrap && c.xt + 1 == a.xt;
    c.xt = xt_wrap ? 0 : (ct_wrap ? c.xt + 1 : c.xt);
    const bool yq_wrap = xt_wrap && c.yq + 1 == a.hq;
    c.yq = yq_wrap ? 0 : (xt_wrap ? c.yq + 1 : c.yq);
    c.b += yq_wrap ? 1 : 0;
  };

  // activations / bias / residual / output as raw buffers: a missing operand is a zero-length buffer (reads as zero); a
  // lane outside the image gets an offset past the end (loads return zero = the convolution's padding, stores are
  // dropped).  No load sits under a lane-dependent branch -- hipcc would stop counting and wait vmcnt(0) at the next use,
  // draining the prefetch every stage.
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? a.cout_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? a.r_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.o_bytes, 0x00020000);

  ConvCursor pfb = cursor_at_first();      // activations: stage being requested (two ahead of the one being computed)
  ConvCursor pfa = cursor_at_first();      // weights: slice being requested (three ahead)
  ConvCursor cur = cursor_at_first();      // stage being computed

  auto load_b = [&](float4 (&bset)[4]) {
    const int y = pfb.yq * 4 + wave;
    const int yy = y * a.S - a.PH + pfb.dy;
    const int xx = (pfb.xt * 32 + p) * a.S - a.PW + pfb.dx;
    // bitwise: a short-circuit && becomes a branch and splits the stage body
    const bool ok = (pfb.it < iters) & (y < a.Ho) & ((unsigned)yy < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W);
    const unsigned off = (unsigned)(((pfb.b * a.H + yy) * a.W + xx) * (int)a.xp + pfb.ch * 32 + 4 * h) * 4u;
    const unsigned voff = ok ? off : 0x80000000u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrd, voff + 32u * j, 0, 0);
      bset[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  // the weight slice in flight: named registers (an array here ends up in scratch memory for MT > 1)
  float4 ar0, ar1, ar2, ar3;
  auto load_a = [&]() {
    const int slice = pfa.it < iters ? pfa.ct * a.nstage + pfa.s : 0;      // past the last tile: any valid slice
    const float4* wsrc = a.w + (int64_t)slice * kSlot + tid;
    ar0 = wsrc[0];
    if constexpr (MT > 1) ar1 = wsrc[256];
    if constexpr (MT > 2) {
      ar2 = wsrc[512];
      ar3 = wsrc[768];
    }
  };
  auto park = [&](int slot) {
    float4* dst = ring + slot * kSlot + tid;
    dst[0] = ar0;
    if constexpr (MT > 1) dst[256] = ar1;
    if constexpr (MT > 2) {
      dst[512] = ar2;
      dst[768] = ar3;
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

  auto mfma_group = [&](const float4 (&af)[4][MT], const float4& bv, int i4) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i4][mt].x, bv.x, acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i4][mt].y, bv.y, acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i4][mt].z, bv.z, acc[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i4][mt].w, bv.w, acc[mt], 0, 0, 0);
  };

  auto epilogue = [&]() {
    const int y = cur.yq * 4 + wave, x = cur.xt * 32 + p;
    const bool store = (y < a.Ho) & (x < a.Wo);
    const int pix = (cur.b * a.Ho + y) * a.Wo + x;
    const int cbase = cur.ct * 32 * MT + 4 * h;
    const unsigned boff = (unsigned)cbase * 4u;
    const unsigned roff = store ? (unsigned)(pix * (int)a.rp + cbase) * 4u : 0x80000000u;
    const unsigned ooff = store ? (unsigned)(pix * (int)a.op + cbase) * 4u : 0x80000000u;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      u32x4 bv[4], rv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bv[g] = __builtin_amdgcn_raw_buffer_load_b128(bsrd, boff + 4u * (mt * 32 + 8 * g), 0, 0);
        rv[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrd, roff + 4u * (mt * 32 + 8 * g), 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = acc[mt][4 * g + c] + __uint_as_float(bv[g][c]);
          v += __uint_as_float(rv[g][c]);
          // none / ReLU / LeakyReLU without a branch: max(v, 0) + slope * min(v, 0), slope = 1 / 0 / 0.01 (one of the two
          // terms is always zero, so this is exact)
          o[c] = __builtin_fmaf(a.slope, fminf(v, 0.f), fmaxf(v, 0.f));
          acc[mt][4 * g + c] = 0.0f;
        }
        u32x4 ov;
        ov.x = __float_as_uint(o[0]); ov.y = __float_as_uint(o[1]); ov.z = __float_as_uint(o[2]); ov.w = __float_as_uint(o[3]);
        __builtin_amdgcn_raw_buffer_store_b128(ov, osrd, ooff + 4u * (mt * 32 + 8 * g), 0, 0);
      }
    }
  };

  float4 b0[4], b1[4], b2[4], af[4][MT];
  // ---- prologue: slices 0 and 1 in the ring, slice 2 in registers, activations of stages 0 and 1 requested ----
  load_a();
  advance(pfa);
  park(0);
  load_a();
  advance(pfa);
  park(1);
  load_a();
  advance(pfa);
  load_b(b0);
  advance(pfb);
  load_b(b1);
  advance(pfb);
  ring_barrier();
  read_a(af, 0, 0);
  read_a(af, 0, 1);

  // ---- one stage.  bc: activations of this stage (landed); bp: register set the stage-after-next is requested into;
  //      sc / sp / sn: ring slots of this stage, of the slice parked now (stage + 2), of the next stage.
  //   G0 | park slice g + 2 (requested one stage ago), request slice g + 3, fragments i4 = 2, 3 of this stage
  //   G1 | request the activations of stage g + 2
  //   G2 | advance the cursors; barrier (publishes the slice parked in this stage; everybody has left slot sp's old
  //      | reads); fragments i4 = 0, 1 of the NEXT stage (its slice was published by the previous stage's barrier)
  //   G3 | end of a tile: epilogue
  // Slot sp = (g + 2) % 3 was last read in stage g - 1 (fragments i4 = 2, 3, before that stage's barrier).
  auto stage = [&](const float4 (&bc)[4], float4 (&bp)[4], int sc, int sp, int sn) {
    mfma_group(af, bc[0], 0);
    SMOS_FENCE();
    park(sp);
    load_a();
    read_a(af, sc, 2);
    read_a(af, sc, 3);
    SMOS_FENCE();
    mfma_group(af, bc[1], 1);
    SMOS_FENCE();
    load_b(bp);
    SMOS_FENCE();
    mfma_group(af, bc[2], 2);
    SMOS_FENCE();
    advance(pfa);
    advance(pfb);
    ring_barrier();
    read_a(af, sn, 0);
    read_a(af, sn, 1);
    SMOS_FENCE();
    mfma_group(af, bc[3], 3);
    SMOS_FENCE();
    if (cur.s + 1 == a.nstage) epilogue();
    advance(cur);
  };

#pragma unroll 1
  for (int g = 0; g < total; g += 3) {
    stage(b0, b2, 0, 2, 1);
    if (g + 1 < total) stage(b1, b0, 1, 0, 2);
    if (g + 2 < total) stage(b2, b1, 2, 1, 0);
  }
}

}  // namespace smos

using namespace smos;

template <int MT>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
  const size_t lds = (size_t)3 * 256 * MT * sizeof(float4);
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&conv_igemm<MT>), lds, 256, &ks, "conv_cl")) return rc;
  static const int want_per_cu = [] {                       // tuning knob (tools/ubench_conv.py); default below
    const char* e = getenv("SMOS_CONV_BLOCKS_PER_CU");
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 8 ? v : 2;
  }();
  const int per_cu = ks.per_cu < want_per_cu ? ks.per_cu : want_per_cu;   // default two blocks per CU = 2 waves per SIMD
  const int64_t cap = (int64_t)ks.cus * per_cu;
  const unsigned grid = (unsigned)(a.n_items < cap ? a.n_items : cap);
  hipLaunchKernelGGL((conv_igemm<MT>), dim3(grid), dim3(256), lds, s, a);
  return check_launch("conv_cl");
}

// w: [Cout][Cin][KH][KW] reordered by ops.conv_prepare for the given MT.  x / res / out: channels-last rows with the given
// pitches (floats), 16-byte aligned.  Replaces conv2d -> BatchNorm (folded) -> ReLU / LeakyReLU (-> + residual -> ReLU) of
// networks/backbone.py:136-159 and multi_view_encoder.py:460-497 in one launch.
extern "C" int smos_conv_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, const float* res,
                            int64_t res_pitch, float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t Cin,
                            int64_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_h, int32_t pad_w, int32_t mt,
                            int32_t act, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && (mt == 1 || mt == 2 || mt == 4) &&
                   Cout % (32 * mt) == 0, "conv_cl: Cin must be a multiple of 32 and Cout of 32 * mt (mt in {1, 2, 4})");
  SMOS_REQUIRE(KH >= 1 && KW >= 1 && KH <= 7 && KW <= 7 && (stride == 1 || stride == 2) && pad_h >= 0 && pad_w >= 0 &&
                   act >= 0 && act <= 2, "conv_cl: kernel up to 7 x 7, stride 1 or 2");
  const int64_t Ho = (H + 2 * pad_h - KH) / stride + 1, Wo = (W + 2 * pad_w - KW) / stride + 1;
  SMOS_REQUIRE(Ho > 0 && Wo > 0, "conv_cl: empty output");
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
  a.x = x; a.w = reinterpret_cast<const float4*>(wprep); a.bias = bias; a.res = res; a.out = out;
  a.xp = x_pitch; a.rp = res_pitch; a.op = out_pitch;
  a.B = (int)B; a.H = (int)H; a.W = (int)W; a.Ho = (int)Ho; a.Wo = (int)Wo;
  a.KH = KH; a.KW = KW; a.S = stride; a.PH = pad_h; a.PW = pad_w;
  a.nch = (int)(Cin / 32); a.nstage = KH * KW * a.nch; a.nct = (int)nct;
  a.hq = (int)hq; a.xt = (int)xt; a.n_items = (int)(B * hq * xt * nct);
  a.slope = act == 0 ? 1.0f : act == 1 ? 0.0f : 0.01f;
  a.x_bytes = (int)(B * H * W * x_pitch * 4);
  a.r_bytes = res ? (int)(B * Ho * Wo * res_pitch * 4) : 0;
  a.o_bytes = (int)(B * Ho * Wo * out_pitch * 4);
  a.cout_bytes = (int)(Cout * 4);
  if (mt == 1) return launch_conv<1>(a, (hipStream_t)stream);
  if (mt == 2) return launch_conv<2>(a, (hipStream_t)stream);
  return launch_conv<4>(a, (hipStream_t)stream);
}
