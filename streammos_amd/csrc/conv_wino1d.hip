// Stride-1 K x 3 / 3 x K convolution (K = 5, 7; channels-last fp32, "same" padding) in the 1-D Winograd F(2, 3) form along
// the 3-tap axis on the matrix cores of gfx950, bias + activation fused -- the two parallel branches of the network's
// Unbalance blocks (multi_view_encoder.py:478-497: 7x3 / 3x7 at 32 channels, 5x3 / 3x5 at 64) do 4 instead of 6
// multiply-adds per pair of outputs, tap of the long axis and channel pair:
//
//     (y0, y1) = A^T [ sum_{cin, kL} (G g_kL) . (B^T d_kL) ]        d_kL: 4 input pixels along the short axis S in the row
//                                                                   the long-axis tap kL selects, "." elementwise
//
// U = G g (per long-axis tap) is computed in float64 on the host and rounded once (ops.conv_wino1d_prepare); B^T d and A^T m
// are additions only, so the arithmetic stays plain fp32 (tools/winograd_numerics.py: error below the direct form's).
//
// B^T d of an input row does not depend on the tap: it is computed ONCE per (input row, k-step) and feeds every output row
// of the wave that reaches it (kL = input row - output row).  With 4 output rows per wave and K = 7 a k-step is 10 patch
// reads (4 ds_read_b32 + 4 additions each) for 224 MFMAs -- the matrix pipe is the only thing that has work, which is why
// this kernel runs one block (four waves) per CU with a 102 KB input region instead of two small ones.
//
//   logical axes = L (long: K taps) and S (short: 3 taps); a 3 x K layer is the K x 3 layer of the transposed image -- pixel
//                  strides are arguments, nothing is transposed in memory (a pixel's 16-channel chunk is one 64-byte segment
//                  either way).
//   work item    = 16 L-rows x 32 S-columns x 16 * MB output channels; wave w = L-rows 4 w .. 4 w + 3, 16 tiles of 1 x 2 outputs
//                  side by side along S (the MFMA column); 16 accumulators (4 rows x 4 positions) per m-block.
//   chunk        = 16 input channels = 4 k-steps; lane (q, tx) supplies channel 4 q + i at k-step i (as csrc/conv_wino.hip).
//   B operand    = region of (16 + K - 1) x 34 pixels x 16 channels staged once per chunk, transposed into the same LDS image
//                  as conv_wino (pixel pitch 17 words: conflict-free ds_read_b32), two buffers; a quarter of it is requested
//                  at the head of each k-step and stored at its tail.
//   A operand    = U [kL][mb][lane][position]: one ds_read_b128 per (kL, m-block) yields the four positions; all K * MB
//                  fragments of a k-step live in registers and are refilled from the next k-step's slice as soon as their
//                  last output row has used them.  Slices travel by LDS-DMA through a three-slot ring, requested at the head
//                  of the k-step before the one that reads them.
#include "conv_wino_common.h"

namespace smos {

struct Wino1dArgs {
  const float* x;      // [B, H, W, *] row pitch xp (floats)
  const float4* w;     // [cout tile][chunk][k-step][kL][mb][lane = q * 16 + m][position]  (ops.conv_wino1d_prepare)
  const float* bias;   // [Cout] or null
  float* out;          // [B, H, W, *] row pitch op
  int64_t xp, op;
  int B, nL, nS;       // image extent along the long / short axis
  int sL, sS;          // pixel strides of the two axes (L = y: sL = W, sS = 1; L = x: sL = 1, sS = W)
  int nchunk;          // Cin / 16
  int nct;             // Cout / (16 * MB)
  int lb, sb;          // ceil(nL / 16), ceil(nS / 32)
  int n_items;         // B * lb * sb * nct
  float slope;
  int x_bytes, o_bytes, cout;
};

struct Wino1dItem {
  int b, l0, s0, ct;
};

template <int KL, int MB>
__global__ __launch_bounds__(256, 1) void conv_wino1d(Wino1dArgs a) {
  constexpr int kRows = 16 + KL - 1;             // staged L-rows
  constexpr int kRegPix = kRows * 34;
  constexpr int kInWords = kRegPix * kWPP;
  constexpr int kSlot = KL * MB * 64;            // float4 per k-step slice
  constexpr int kPad = KL / 2;
  constexpr int kRounds = (kRegPix + 63) / 64;   // staging rounds of 64 pixels (thread = pixel slot x channel group)
  constexpr int kPerStep = (kRounds + 3) / 4;    // rounds per k-step (3 for K = 7 and K = 5)
  static_assert(kPerStep == 3, "staging schedule assumes three rounds per k-step");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4* w_lds = reinterpret_cast<float4*>(lds + 2 * kInWords);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, tx = lane & 15;

  const int per_block = (a.n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nb = (int)gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = (int)blockIdx.x & 7;
  const int lblock = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
  const int first = lblock * per_block;
  const int iters = a.n_items - first < per_block ? a.n_items - first : per_block;
  if (iters <= 0) return;
  const int total = iters * a.nchunk;

  auto item_at = [&](int u) {
    Wino1dItem t;
    t.ct = u % a.nct;
    u /= a.nct;
    t.s0 = (u % a.sb) * 32;
    u /= a.sb;
    t.l0 = (u % a.lb) * 16;
    t.b = u / a.lb;
    return t;
  };

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? a.cout * 4 : 0, 0x00020000);

  // ---- staging: region pixel p = round * 64 + (tid >> 2), channel group tid & 3; p -> (row l = p / 34, column s = p % 34) ----
  const int xp = (int)a.xp;
  const int sc4 = tid & 3, slot = tid >> 2;
  u32x4 st[kPerStep];
  auto stage_load_one = [&](const Wino1dItem& t, int c, int step, int j, bool valid) {
    const int p = (step * kPerStep + j) * 64 + slot;
    const int l = p / 34, s = p - l * 34;
    const int gl = t.l0 - kPad + l, gs = t.s0 - 1 + s;
    const bool ok = valid & (p < kRegPix) & ((unsigned)gl < (unsigned)a.nL) & ((unsigned)gs < (unsigned)a.nS);
    const unsigned off = ok ? (unsigned)(((t.b * a.nL * a.nS + gl * a.sL + gs * a.sS) * xp + 16 * c + 4 * sc4) * 4) : 0x80000000u;
    st[j] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, off, 0, 0);
  };
  auto stage_write_one = [&](float* buf, int step, int j) {
    const int p = (step * kPerStep + j) * 64 + slot;
    if (p < kRegPix) {
      float* d = buf + p * kWPP + sc4;        // channel 4 c4 + i of pixel p at word p * 17 + 4 i + c4
      d[0] = __uint_as_float(st[j].x);
      d[4] = __uint_as_float(st[j].y);
      d[8] = __uint_as_float(st[j].z);
      d[12] = __uint_as_float(st[j].w);
    }
  };
  auto stage_load = [&](const Wino1dItem& t, int c, int step, bool valid) {
#pragma unroll
    for (int j = 0; j < kPerStep; ++j) stage_load_one(t, c, step, j, valid);
  };
  auto stage_write = [&](float* buf, int step) {
#pragma unroll
    for (int j = 0; j < kPerStep; ++j) stage_write_one(buf, step, j);
  };

  // ---- weights: consecutive k-steps' slices are consecutive, cyclically over the block's items (cout tile fastest) ----
  const int n_slices = a.nct * a.nchunk * 4;
  int pa_slice = (first % a.nct) * a.nchunk * 4;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int kPieces = (kSlot + 255) / 256;   // 1 KB pieces per wave quarter
  auto w_piece = [&](int so, int k) {               // piece k of the slice pa_slice names; the last piece advances it
    const float4* s_ = a.w + (int64_t)__builtin_amdgcn_readfirstlane(pa_slice) * kSlot;
    const int e0 = (k * 4 + wave) * 64;            // this wave's 64 float4 of piece k (wave-uniform)
    if (e0 < kSlot) __builtin_amdgcn_global_load_lds((gptr_t)(s_ + e0 + lane), (lptr_t)(w_lds + so + e0), 16, 0, 0);
    if (k == kPieces - 1) pa_slice = pa_slice + 1 == n_slices ? 0 : pa_slice + 1;
  };
  auto w_load = [&](int so) {
#pragma unroll
    for (int k = 0; k < kPieces; ++k) w_piece(so, k);
  };

  f32x4 acc[4][MB][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[r][mb][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 af[KL][MB];

  // this lane's patch origin: region row 4 * wave (+ input row), column 2 * tx, word q
  const int in_base = ((4 * wave) * 34 + 2 * tx) * kWPP + q;

  // the four pixels of input row li (this lane's tile, channel 4 q + i) / B^T d in place
  auto load_patch = [&](const float* buf, int li, int i, float (&d)[4]) {
    const float* p = buf + in_base + (li * 34) * kWPP + 4 * i;
    d[0] = p[0];
    d[1] = p[kWPP];
    d[2] = p[2 * kWPP];
    d[3] = p[3 * kWPP];
  };
  auto transform = [&](float (&d)[4]) {
    const float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
    d[0] = d0 - d2;
    d[1] = d1 + d2;
    d[2] = d2 - d1;
    d[3] = d1 - d3;
  };
  auto a_read = [&](int so, int kl) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) af[kl][mb] = w_lds[so + (kl * MB + mb) * 64 + lane];
  };

  // ---- one k-step: the wave's 4 + KL - 1 input rows in turn.  The fragments of this k-step are in af on entry; so1: the next
  //      k-step's slice, from which af is refilled tap by tap.  The pixels of row li + 1 are requested before the MFMAs of row
  //      li and transformed behind them.  With one wave per SIMD nothing but the wave's own instruction stream can fill the
  //      matrix pipe's shadow, so everything else a k-step has to do rides INSIDE the MFMA stream, one piece behind the first
  //      eight MFMAs of an input row: the weight DMA of k-step + 2 (rows 0 ..), the requests for a quarter of the next chunk's
  //      region (three rows in the middle) and its stores (the last three rows, three rows = ~3000 cycles behind the request). ----
  constexpr int kLi = 4 + KL - 1;
  auto kstep = [&](const float* buf, float* nbuf, int i, int so1, int so2, const Wino1dItem& nt, int nc, bool more) {
    float v[3][4];                              // row li's B operands, row li + 1 (transformed during li), row li + 2 (in flight)
    load_patch(buf, 0, i, v[0]);
    load_patch(buf, 1, i, v[1]);
    transform(v[0]);
    SMOS_FENCE();
#pragma unroll
    for (int li = 0; li < kLi; ++li) {
      int n_mfma = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kl = li - r;
        if (kl < 0 || kl >= KL) continue;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          acc[r][mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kl][mb].x, v[li % 3][0], acc[r][mb][0], 0, 0, 0);
          acc[r][mb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kl][mb].y, v[li % 3][1], acc[r][mb][1], 0, 0, 0);
          acc[r][mb][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kl][mb].z, v[li % 3][2], acc[r][mb][2], 0, 0, 0);
          acc[r][mb][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kl][mb].w, v[li % 3][3], acc[r][mb][3], 0, 0, 0);
        }
        n_mfma += 4 * MB;
      }
      // everything else of this row: independent of its MFMAs, to be issued one instruction per MFMA shadow
      if (li + 1 < kLi) transform(v[(li + 1) % 3]);
      if (li + 2 < kLi) load_patch(buf, li + 2, i, v[(li + 2) % 3]);
      if (li < kPieces) w_piece(so2, li);
      if (li >= kLi - 6 && li < kLi - 3) stage_load_one(nt, nc, i, li - (kLi - 6), more);
      if (li >= kLi - 3) stage_write_one(nbuf, i, li - (kLi - 3));
      if (li >= 3) a_read(so1, li - 3);        // tap li - 3 has served its last output row: next k-step's fragment
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (k < n_mfma) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x096, 1, 0);     // one VALU / SALU / VMEM / DS instruction
        }
      }
      SMOS_FENCE();
    }
  };

  auto epilogue = [&](const Wino1dItem& t) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int c0 = (t.ct * MB + mb) * 16 + 4 * q;
      const u32x4 braw = __builtin_amdgcn_raw_buffer_load_b128(bsrd, (unsigned)c0 * 4u, 0, 0);
      const f32x4 bv = {__uint_as_float(braw.x), __uint_as_float(braw.y), __uint_as_float(braw.z), __uint_as_float(braw.w)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gl = t.l0 + 4 * wave + r;
        f32x4 y[2];
        y[0] = (acc[r][mb][0] + acc[r][mb][1]) + acc[r][mb][2];
        y[1] = (acc[r][mb][1] - acc[r][mb][2]) - acc[r][mb][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[r][mb][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int gs = t.s0 + 2 * tx + e;
          const bool ok = (gl < a.nL) & (gs < a.nS);
          const f32x4 vv = y[e] + bv;
          u32x4 ov;
#pragma unroll
          for (int k = 0; k < 4; ++k) ov[k] = __float_as_uint(__builtin_fmaf(a.slope, fminf(vv[k], 0.f), fmaxf(vv[k], 0.f)));
          const unsigned off = ok ? (unsigned)(((t.b * a.nL * a.nS + gl * a.sL + gs * a.sS) * (int)a.op + c0) * 4) : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(ov, osrd, off, 0, 0);
        }
      }
    }
  };

  // ---- prologue: region of chunk 0 in buffer 0, slices 0 and 1 in slots 0 and 1, fragments of k-step 0 ----
  int so0 = 0, so1 = kSlot, so2 = 2 * kSlot;
  w_load(so0);
  w_load(so1);
  {
    // the whole first region in one flight (12 requests per thread; the steady state moves 3 per k-step)
    const Wino1dItem t0 = item_at(first);
    u32x4 all[4][kPerStep];
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      stage_load(t0, 0, step, true);
#pragma unroll
      for (int j = 0; j < kPerStep; ++j) all[step][j] = st[j];
    }
#pragma unroll
    for (int step = 0; step < 4; ++step) {
#pragma unroll
      for (int j = 0; j < kPerStep; ++j) st[j] = all[step][j];
      stage_write(lds, step);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ring_barrier();
#pragma unroll
  for (int kl = 0; kl < KL; ++kl) a_read(so0, kl);

  float* buf_cur = lds;
  float* buf_nxt = lds + kInWords;
  int it = 0, c = 0;
#pragma unroll 1
  for (int g = 0; g < total; ++g) {
    int it1 = it, c1 = c + 1;
    if (c1 == a.nchunk) {
      c1 = 0;
      ++it1;
    }
    const bool more = g + 1 < total;
    const Wino1dItem nxt = item_at(first + (more ? it1 : it));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // inside the k-step: the slice of k-step + 2 (this k-step refills the fragments from the slice of k-step + 1, complete
      // since the previous barrier) and a quarter of the next chunk's region.  a wave's wait counter knows nothing of the other waves' DMA:
      // explicit wait, then barrier.
      kstep(buf_cur, buf_nxt, i, so1, so2, nxt, c1, more);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ring_barrier();
      const int r_ = so0;
      so0 = so1;
      so1 = so2;
      so2 = r_;
    }
    if (c1 == 0) epilogue(item_at(first + it));
    it = it1;
    c = c1;
    float* sw = buf_cur;
    buf_cur = buf_nxt;
    buf_nxt = sw;
  }
}

}  // namespace smos

using namespace smos;

template <int KL, int MB>
static int launch_wino1d(const Wino1dArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (16 + KL - 1) * 34 * kWPP * sizeof(float) + (size_t)3 * KL * MB * 64 * sizeof(float4);
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&conv_wino1d<KL, MB>), lds, 256, &ks, "conv_wino1d_cl")) return rc;
  const int64_t cap = conv_grid_cap((int64_t)ks.cus);          // one block per CU: the region and the ring take 145 KB of LDS at K = 7
  const unsigned grid = (unsigned)(a.n_items < cap ? a.n_items : cap);
  hipLaunchKernelGGL((conv_wino1d<KL, MB>), dim3(grid), dim3(256), lds, s, a);
  return check_launch("conv_wino1d_cl");
}

// act(conv(x, w) + bias) for a stride-1 KH x KW kernel with one extent 3 and the other 5 or 7 ("same" padding), 1-D Winograd
// F(2, 3) along the 3-tap axis.  wprep = ops.conv_wino1d_prepare(w, mb); pointers 16-byte aligned, channels-last with row
// pitches in floats; mb in {1, 2}: 16 * mb output channels per block.  Replaces nn.Conv2d + BatchNorm2d + ReLU of the two
// parallel branches of multi_view_encoder.py:478-497 (Unbalance_BasicBlock).
extern "C" int smos_conv_wino1d_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, float* out, int64_t out_pitch,
                                   int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, int32_t mb,
                                   int32_t act, smos_stream_t stream) {
  const bool l_is_y = KW == 3 && (KH == 5 || KH == 7), l_is_x = KH == 3 && (KW == 5 || KW == 7);
  SMOS_REQUIRE(l_is_y || l_is_x, "conv_wino1d_cl: kernel must be 5x3, 7x3, 3x5 or 3x7");
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && (mb == 1 || mb == 2) && Cout % (16 * mb) == 0 &&
                   act >= 0 && act <= 2, "conv_wino1d_cl: Cin must be a multiple of 16 and Cout of 16 * mb (mb in {1, 2})");
  SMOS_REQUIRE(Cout <= 2048, "conv_wino1d_cl: more than 2048 output channels");
  SMOS_REQUIRE(x && wprep && out && x_pitch >= Cin && out_pitch >= Cout && x_pitch % 4 == 0 && out_pitch % 4 == 0,
               "conv_wino1d_cl: null pointer / bad pitch");
  SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias) |
                 reinterpret_cast<uintptr_t>(wprep)) & 15) == 0, "conv_wino1d_cl: pointers must be 16-byte aligned");
  SMOS_REQUIRE(B * H * W * x_pitch * 4 < (1LL << 31) && B * H * W * out_pitch * 4 < (1LL << 31),
               "conv_wino1d_cl: a tensor larger than 2 GiB (32-bit buffer offsets)");
  const int64_t nL = l_is_y ? H : W, nS = l_is_y ? W : H, KL = l_is_y ? KH : KW;
  const int64_t lb = (nL + 15) / 16, sb = (nS + 31) / 32, nct = Cout / (16 * mb);
  SMOS_REQUIRE(B * lb * sb * nct < (1LL << 30) && nct * (Cin / 16) < (1LL << 24), "conv_wino1d_cl: too many tiles");
  Wino1dArgs a;
  a.x = x; a.w = reinterpret_cast<const float4*>(wprep); a.bias = bias; a.out = out;
  a.xp = x_pitch; a.op = out_pitch;
  a.B = (int)B; a.nL = (int)nL; a.nS = (int)nS;
  a.sL = l_is_y ? (int)W : 1; a.sS = l_is_y ? 1 : (int)W;
  a.nchunk = (int)(Cin / 16); a.nct = (int)nct; a.lb = (int)lb; a.sb = (int)sb; a.n_items = (int)(B * lb * sb * nct);
  a.slope = act == 0 ? 1.0f : act == 1 ? 0.0f : 0.01f;
  a.x_bytes = (int)(B * H * W * x_pitch * 4);
  a.o_bytes = (int)(B * H * W * out_pitch * 4);
  a.cout = (int)Cout;
  hipStream_t s = (hipStream_t)stream;
  if (KL == 7) return mb == 1 ? launch_wino1d<7, 1>(a, s) : launch_wino1d<7, 2>(a, s);
  return mb == 1 ? launch_wino1d<5, 1>(a, s) : launch_wino1d<5, 2>(a, s);
}
