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

struct ConvTile {      // the 32-pixel row segment a wave works on
  int b, y, x0, ct;
  bool valid;
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

template <int MT>
__global__ __launch_bounds__(256, 2) void conv_igemm(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 ring[];     // 3 slots x 256 * MT float4
  constexpr int kSlot = 256 * MT;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int iters = (a.n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = iters * a.nstage;
  if (total <= 0) return;

  auto tile_of = [&](int it) {
    ConvTile t;
    const int item = (int)blockIdx.x + it * (int)gridDim.x;
    t.valid = it < iters;
    const int q = t.valid ? item : 0;
    t.ct = q % a.nct;
    int u = q / a.nct;
    const int xt = u % a.xt;
    u /= a.xt;
    const int yq = u % a.hq;
    t.b = u / a.hq;
    t.y = yq * 4 + wave;
    t.x0 = xt * 32;
    t.valid = t.valid && t.y < a.Ho;
    return t;
  };

  // activations as a raw buffer: [0, x_bytes) readable, everything else reads as zero
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);

  // ---- prefetch cursor: the stage whose operands are being requested (two ahead of the one being computed) ----
  int pf_it = 0, pf_s = 0, pf_dy = 0, pf_dx = 0, pf_ch = 0;
  ConvTile pf = tile_of(0);
  auto issue = [&](float4 (&bset)[4], float4 (&areg)[MT]) {
    // weights first: they are the first thing this stage waits for (vmcnt retires in order).  No load sits under a
    // lane-dependent branch -- hipcc would stop counting and wait vmcnt(0) at the next use, draining this prefetch every
    // stage: the activations come through a buffer descriptor whose range check returns zeros for the lanes outside the
    // image (their offset is pushed past the end), which is the convolution's zero padding for free.
    const float4* wsrc = a.w + ((int64_t)(pf.ct * a.nstage + pf_s)) * kSlot + tid;      // past the last tile: slice of item 0
#pragma unroll
    for (int m = 0; m < MT; ++m) areg[m] = wsrc[256 * m];
    const int yy = pf.y * a.S - a.PH + pf_dy;
    const int xx = (pf.x0 + p) * a.S - a.PW + pf_dx;
    const bool ok = pf.valid && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
    const unsigned off = (unsigned)(((pf.b * a.H + yy) * a.W + xx) * (int)a.xp + pf_ch * 32 + 4 * h) * 4u;
    const unsigned voff = ok ? off : 0x80000000u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrd, voff + 32u * j, 0, 0);
      bset[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    // advance: channel chunk fastest, then the tap column, the tap row, the tile
    ++pf_s;
    if (++pf_ch == a.nch) {
      pf_ch = 0;
      if (++pf_dx == a.KW) {
        pf_dx = 0;
        ++pf_dy;
      }
    }
    if (pf_s == a.nstage) {
      pf_s = pf_dy = pf_dx = 0;
      ++pf_it;
      pf = tile_of(pf_it);
    }
  };

  // ---- compute cursor ----
  int c_it = 0, c_s = 0;
  ConvTile cur = tile_of(0);
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

  // bias / residual / output as raw buffers too: a missing operand is a zero-length buffer (reads as zero), lanes past
  // the image edge use an offset past the end (loads return zero, stores are dropped) -- no load or store of the
  // epilogue sits under a branch, so the eight loads of a 32-channel block are in flight together.
  const __amdgpu_buffer_rsrc_t bsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? a.cout_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? a.r_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.o_bytes, 0x00020000);

  auto epilogue = [&]() {
    const bool store = cur.valid && cur.x0 + p < a.Wo;
    const int pix = (cur.b * a.Ho + cur.y) * a.Wo + cur.x0 + p;
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

  auto compute = [&](const float4 (&bset)[4], int slot) {
    const float4* as = ring + slot * kSlot + lane;
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 av = as[(i4 * MT + mt) * 64];
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bset[i4].x, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bset[i4].y, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bset[i4].z, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bset[i4].w, acc[mt], 0, 0, 0);
      }
    }
  };

  auto park = [&](const float4 (&areg)[MT], int slot) {
#pragma unroll
    for (int m = 0; m < MT; ++m) ring[slot * kSlot + tid + 256 * m] = areg[m];
  };

  auto finish_stage = [&]() {
    if (++c_s == a.nstage) {
      epilogue();
      c_s = 0;
      ++c_it;
      cur = tile_of(c_it);
    }
  };

  float4 b0[4], b1[4], b2[4], ar[MT];
  // prologue: stages 0 and 1 requested, their weight slices in ring slots 0 and 1
  issue(b0, ar);
  park(ar, 0);
  issue(b1, ar);
  park(ar, 1);
  ring_barrier();

  // one stage = { request stage g + 2; MFMAs of stage g; park the slice of g + 2; barrier }.  Slot (g + 2) % 3 was last
  // read in stage g - 1, which every wave left through that stage's barrier; it is read again in stage g + 2, two
  // barriers from now.
#pragma unroll 1
  for (int g = 0; g < total; g += 3) {
    issue(b2, ar);
    compute(b0, 0);
    park(ar, 2);
    ring_barrier();
    finish_stage();
    if (g + 1 < total) {
      issue(b0, ar);
      compute(b1, 1);
      park(ar, 0);
      ring_barrier();
      finish_stage();
    }
    if (g + 2 < total) {
      issue(b1, ar);
      compute(b2, 2);
      park(ar, 1);
      ring_barrier();
      finish_stage();
    }
  }
}

}  // namespace smos

using namespace smos;

template <int MT>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
  const size_t lds = (size_t)3 * 256 * MT * sizeof(float4);
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&conv_igemm<MT>), lds, 256, &ks, "conv_cl")) return rc;
  const int per_cu = ks.per_cu < 2 ? ks.per_cu : 2;        // two blocks per CU: 2 waves per SIMD, room left for the other stream
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
