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
  float4* w_lds = reinterpret_cast<float4*>(lds + 2 * kWInWords);       // three slots of 256 * MB float4
  constexpr int kSlot = 256 * MB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, tx = lane & 15;

  const int per_block = (a.n_items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nb = (int)gridDim.x, xq = nb >> 3, xr = nb & 7, xcd = (int)blockIdx.x & 7;
  const int lblock = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + ((int)blockIdx.x >> 3);
  // Two item orders.  a.group == 0: the block owns a contiguous item range, cout tile fastest (a block stages a region once per
  // cout tile, nct times in a row).  a.group == 1 (grid a multiple of nct): nct consecutive blocks -- same XCD, dispatched
  // together, equal work -- walk the SAME regions in step, one cout tile each: the region's second .. nct-th reader finds it in
  // the L2 the first one just filled instead of fetching it again many chunks later, and a block cycles through one cout
  // tile's weight slices only.  With one item per block the two orders are the same assignment.
  const int regions = a.n_items / a.nct;
  const int ngrp = nb / a.nct;
  const int grp = lblock / a.nct, ctm = lblock - grp * a.nct;
  const int per_group = a.group ? (regions + ngrp - 1) / ngrp : 0;
  const int first = a.group ? grp * per_group : lblock * per_block;          // first region / first item
  const int left = (a.group ? regions : a.n_items) - first, mine = a.group ? per_group : per_block;
  const int iters = left < mine ? left : mine;
  if (iters <= 0) return;
  const int total = iters * a.nchunk;

  // the block's items in order: cout tile fastest, then the 32-column block, the 8-row block, the sample.  Only the first one is
  // located with divisions; the walk is incremental (scalar selects; past the block's last item the coordinates are unused:
  // every request for such a chunk is masked off)
  auto first_item = [&]() {
    WinoItem t;
    int u = first;
    if (a.group) {
      t.ct = ctm;
    } else {
      t.ct = u % a.nct;
      u /= a.nct;
    }
    t.x0 = (u % a.xb) * 32;
    u /= a.xb;
    t.y0 = (u % a.yb) * 8;
    t.b = u / a.yb;
    return t;
  };
  auto next_item = [&](const WinoItem& t) {
    WinoItem n = t;
    const bool w_ct = a.group ? true : t.ct + 1 == a.nct;
    const bool w_x = w_ct & (t.x0 + 32 >= a.xb * 32);
    const bool w_y = w_x & (t.y0 + 8 >= a.yb * 8);
    n.ct = a.group ? t.ct : (w_ct ? 0 : t.ct + 1);
    n.x0 = w_x ? 0 : t.x0 + (w_ct ? 32 : 0);
    n.y0 = w_y ? 0 : t.y0 + (w_x ? 8 : 0);
    n.b = t.b + (w_y ? 1 : 0);
    return n;
  };

  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? a.r_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias ? a.cout * 4 : 0, 0x00020000);

  // ---- staging of a chunk's input region (10 rows x 34 pixels x 16 channels = 1360 float4, 5.3 per thread).  Thread =
  // (slot = tid / 4, channel group c4 = tid % 4).  Rounds 0..4 cover columns 0..31 of two region rows each: row 2 k + (slot >> 5)
  // -- wave-uniform: slot >> 5 == wave >> 1 -- and column slot & 31; round 5 covers columns 32, 33 of all ten rows with its
  // first 20 slots.  The address of rounds 0..4 is then a per-thread term plus a SCALAR per (item, chunk, round), the row test
  // is scalar and the column test is shared by the rounds: a few vector instructions per request instead of a dozen and no
  // exec-masked branch.  The per-thread terms are recomputed from an opaque copy of the slot where they are used: kept across
  // the loop they would cost the registers mb = 2 does not have. ----
  const int xp = (int)a.xp;
  const int sc4 = tid & 3;
  const int s_pyw = wave >> 1;
  const int row2_bytes = 2 * a.W * xp * 4;
  u32x4 st0, st1, st2;
#define WINO_SLOT(name)     \
  int name = tid >> 2;      \
  asm volatile("" : "+v"(name))
  // origin = byte offset of region pixel (0, 0), channel 16 c -- "negative" in the first row / column of the image: unsigned
  // wrap-around, the sum with the per-thread term is exact modulo 2^32 for every pixel inside the image
#define WINO_STAGE_ORIGIN(t, c) ((unsigned)(((((t).b * a.H + (t).y0 - 1) * a.W + (t).x0 - 1) * xp + 16 * (c)) * 4))
#define WINO_STAGE_ROW(dst, k, t, org, valid, trel, okx)                                                           \
  do {                                                                                                             \
    const bool oky_ = (valid) & ((unsigned)((t).y0 - 1 + 2 * (k) + s_pyw) < (unsigned)a.H);      /* scalar */        \
    const unsigned off_ = (oky_ & (okx)) ? (trel) + ((org) + (unsigned)((k) * row2_bytes)) : 0x80000000u;          \
    if (WINO_AB(1)) dst = u32x4{off_, 0u, 0u, 0u};                                                                 \
    else dst = __builtin_amdgcn_raw_buffer_load_b128(xsrd, off_, 0, 0);                                            \
  } while (0)
#define WINO_STAGE_TERMS(t)                                                                                        \
  WINO_SLOT(sl_);                                                                                                  \
  const int px_ = sl_ & 31;                                                                                        \
  const unsigned trel_ = (unsigned)((s_pyw * a.W + px_) * xp + 4 * sc4) * 4u;                                      \
  const bool okx_ = (unsigned)((t).x0 - 1 + px_) < (unsigned)a.W
  // the region travels in two halves through the SAME three registers (rounds 0..2, then 3..5): 12 instead of 24 live
#define WINO_STAGE_LOAD_A(t, c, valid)                       \
  do {                                                       \
    const unsigned org_ = WINO_STAGE_ORIGIN(t, c);           \
    WINO_STAGE_TERMS(t);                                     \
    WINO_STAGE_ROW(st0, 0, t, org_, valid, trel_, okx_);     \
    WINO_STAGE_ROW(st1, 1, t, org_, valid, trel_, okx_);     \
    WINO_STAGE_ROW(st2, 2, t, org_, valid, trel_, okx_);     \
  } while (0)
#define WINO_STAGE_LOAD_B(t, c, valid)                                                                             \
  do {                                                                                                             \
    const unsigned org_ = WINO_STAGE_ORIGIN(t, c);                                                                 \
    WINO_STAGE_TERMS(t);                                                                                           \
    WINO_STAGE_ROW(st0, 3, t, org_, valid, trel_, okx_);                                                           \
    WINO_STAGE_ROW(st1, 4, t, org_, valid, trel_, okx_);                                                           \
    const int rpy_ = sl_ >> 1, rpx_ = 32 + (sl_ & 1);               /* round 5: columns 32, 33 of the ten rows */  \
    const bool okr_ = (valid) & (sl_ < 20) & ((unsigned)((t).y0 - 1 + rpy_) < (unsigned)a.H) &                     \
                      ((unsigned)((t).x0 - 1 + rpx_) < (unsigned)a.W);                                              \
    const unsigned offr_ = okr_ ? (unsigned)((rpy_ * a.W + rpx_) * xp + 4 * sc4) * 4u + org_ : 0x80000000u;        \
    if (WINO_AB(1)) st2 = u32x4{offr_, 0u, 0u, 0u};                                                                \
    else st2 = __builtin_amdgcn_raw_buffer_load_b128(xsrd, offr_, 0, 0);                                           \
  } while (0)
  // channel 4 c4 + i of pixel p goes to word p * 17 + 4 i + c4 (k-step i reads word 4 i + q)
#define WINO_PARK_AT(dptr, src)                               \
  do {                                                        \
    float* d_ = (dptr);                                       \
    if (WINO_AB(2) && src.x != 0x7fc12345u) break;            \
    d_[0] = __uint_as_float(src.x);                           \
    d_[4] = __uint_as_float(src.y);                           \
    d_[8] = __uint_as_float(src.z);                           \
    d_[12] = __uint_as_float(src.w);                          \
  } while (0)
#define WINO_STAGE_WRITE_A(buf)                                                     \
  do {                                                                              \
    WINO_SLOT(sl_);                                                                 \
    float* b_ = (buf) + ((s_pyw * kWRegW) + (sl_ & 31)) * kWPP + sc4;               \
    WINO_PARK_AT(b_ + 0 * 2 * kWRegW * kWPP, st0);                                  \
    WINO_PARK_AT(b_ + 1 * 2 * kWRegW * kWPP, st1);                                  \
    WINO_PARK_AT(b_ + 2 * 2 * kWRegW * kWPP, st2);                                  \
  } while (0)
#define WINO_STAGE_WRITE_B(buf)                                                     \
  do {                                                                              \
    WINO_SLOT(sl_);                                                                 \
    float* b_ = (buf) + ((s_pyw * kWRegW) + (sl_ & 31)) * kWPP + sc4;               \
    WINO_PARK_AT(b_ + 3 * 2 * kWRegW * kWPP, st0);                                  \
    WINO_PARK_AT(b_ + 4 * 2 * kWRegW * kWPP, st1);                                  \
    if (sl_ < 20) WINO_PARK_AT((buf) + ((sl_ >> 1) * kWRegW + 32 + (sl_ & 1)) * kWPP + sc4, st2); \
  } while (0)

  // ---- weights: the slices of consecutive k-steps are consecutive, cyclically over the block's items (cout tile fastest).
  // They travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write; a wave's 64 lanes fill 1 KB
  // of the slot, which is exactly the slice's lane-linear layout): the slice of k-step s + 2 is requested at the head of
  // k-step s into ring slot (s + 2) % 3.  hipcc's own ordering of LDS-DMA against LDS reads is not what the ring needs (it puts
  // s_waitcnt vmcnt(0) in front of the first LDS read that may alias a pending DMA, once per k-step; another wave's DMA it cannot
  // see at all), so the wait that matters is explicit: before the
  // barrier that ends k-step s, everything but the requests issued during k-step s itself has landed (vmcnt retires in
  // order), i.e. the slice k-step s + 1 reads ----
  // (group order: the block's one cout tile, cyclically)
  const int slice_lo = a.group ? ctm * a.nchunk * 4 : 0;
  const int n_slices = a.group ? slice_lo + a.nchunk * 4 : a.nct * a.nchunk * 4;
  int pa_slice = a.group ? slice_lo : (first % a.nct) * a.nchunk * 4;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
#define WINO_W_LOAD(so)                                                                                          \
  do {                                                                                                           \
    const float4* s_ = a.w + (int64_t)__builtin_amdgcn_readfirstlane(pa_slice) * kSlot + tid;                     \
    float4* d_ = w_lds + (so) + wave * 64;                                                                       \
    if (!WINO_AB(4)) {                                                                                           \
      __builtin_amdgcn_global_load_lds((gptr_t)s_, (lptr_t)d_, 16, 0, 0);                                        \
      if constexpr (MB > 1) __builtin_amdgcn_global_load_lds((gptr_t)(s_ + 256), (lptr_t)(d_ + 256), 16, 0, 0);  \
    }                                                                                                            \
    pa_slice = pa_slice + 1 == n_slices ? slice_lo : pa_slice + 1;                                               \
  } while (0)
#define WINO_VM_WAIT(n)                                                   \
  do {                                                                    \
    if (!WINO_AB(128)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n) : "memory"); \
  } while (0)
#define WINO_BARRIER()                                   \
  do {                                                   \
    if (WINO_AB(8)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    else ring_barrier();                                 \
  } while (0)

  f32x4 acc[MB][16];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[mb][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's patch origin inside a region buffer: tile (row = wave, column = tx) -> region pixel (2 wave, 2 tx), word q
  const int in_base = ((2 * wave) * kWRegW + 2 * tx) * kWPP + q;

  // ---- pieces of a k-step ----
  float va[16], vb[16];                  // B operands of the current / next k-step (the next patch is transformed in place)
  float4 afa[MB], afb[MB];               // A operands of two consecutive (xi) groups
#ifdef SMOS_WINO_ABLATE
  for (int k = 0; k < 16; ++k) va[k] = vb[k] = (float)(lane + k);
  for (int k = 0; k < MB; ++k) afa[k] = afb[k] = make_float4((float)lane, 1.f, 2.f, (float)k);
#endif
#define WINO_D_READ(v, buf, i)                                                                          \
  do {                                                                                                  \
    if (WINO_AB(16)) break;                                                                             \
    const float* pin_ = (buf) + in_base + 4 * (i);                                                      \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_)                                                    \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) v[4 * r_ + c_] = pin_[(r_ * kWRegW + c_) * kWPP]; \
  } while (0)
  // B^T d (rows), in place: row 0 <- d0 - d2, 1 <- d1 + d2, 2 <- d2 - d1, 3 <- d1 - d3
#define WINO_T_ROWS(v)                                         \
  do {                                                         \
    if (WINO_AB(16)) break;                                    \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {         \
      const float d0_ = v[c_], d1_ = v[4 + c_], d2_ = v[8 + c_], d3_ = v[12 + c_]; \
      v[c_] = d0_ - d2_;                                       \
      v[4 + c_] = d1_ + d2_;                                   \
      v[8 + c_] = d2_ - d1_;                                   \
      v[12 + c_] = d1_ - d3_;                                  \
    }                                                          \
  } while (0)
  // (B^T d) B (columns), in place -> the 16 B operands
#define WINO_T_COLS(v)                                         \
  do {                                                         \
    if (WINO_AB(16)) break;                                    \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {         \
      const float t0_ = v[4 * r_], t1_ = v[4 * r_ + 1], t2_ = v[4 * r_ + 2], t3_ = v[4 * r_ + 3]; \
      v[4 * r_] = t0_ - t2_;                                   \
      v[4 * r_ + 1] = t1_ + t2_;                               \
      v[4 * r_ + 2] = t2_ - t1_;                               \
      v[4 * r_ + 3] = t1_ - t3_;                               \
    }                                                          \
  } while (0)
#define WINO_A_READ(af, so, g)                                                                          \
  do {                                                                                                  \
    if (WINO_AB(32)) break;                                                                             \
    _Pragma("unroll") for (int mb_ = 0; mb_ < MB; ++mb_) af[mb_] = w_lds[(so) + (mb_ * 4 + (g)) * 64 + lane]; \
  } while (0)
  // the MFMAs of one (xi) group for m-block mb: nu = 0..3
#define WINO_MFMA4(v, af, g, mb_)                                                                                             \
  do {                                                                                                                        \
    acc[mb_][4 * (g) + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb_].x, v[4 * (g) + 0], acc[mb_][4 * (g) + 0], 0, 0, 0);  \
    acc[mb_][4 * (g) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb_].y, v[4 * (g) + 1], acc[mb_][4 * (g) + 1], 0, 0, 0);  \
    acc[mb_][4 * (g) + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb_].z, v[4 * (g) + 2], acc[mb_][4 * (g) + 2], 0, 0, 0);  \
    acc[mb_][4 * (g) + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb_].w, v[4 * (g) + 3], acc[mb_][4 * (g) + 3], 0, 0, 0);  \
  } while (0)
#define WINO_MFMA_GROUP(v, af, g)                                            \
  do {                                                                       \
    _Pragma("unroll") for (int mb_ = 0; mb_ < MB; ++mb_) WINO_MFMA4(v, af, g, mb_); \
  } while (0)

  // ---- one k-step.  v: B operands of this k-step (ready); vn: receives those of the next one, whose patch is read from
  //      (nbuf, channel ni); so0 / so1 / so2: ring slots (float4 offsets) of this k-step, the next one and the one the slice
  //      requested here is parked in; head / tail: extra pieces (region requests / region stores).  afa holds group 0 of this
  //      k-step on entry and of the next one on exit; the slots rotate at the end. ----
#define WINO_KSTEP(v, vn, nbuf, ni, HEAD, TAIL, NVM)                   \
  do {                                                                 \
    WINO_A_READ(afb, so0, 1);                                          \
    HEAD;                                                              \
    WINO_D_READ(vn, nbuf, ni);                                         \
    SMOS_FENCE();                                                      \
    WINO_MFMA_GROUP(v, afa, 0);                                        \
    SMOS_FENCE();                                                      \
    WINO_A_READ(afa, so0, 2);                                          \
    WINO_T_ROWS(vn);                                                   \
    SMOS_FENCE();                                                      \
    WINO_MFMA_GROUP(v, afb, 1);                                        \
    SMOS_FENCE();                                                      \
    WINO_A_READ(afb, so0, 3);                                          \
    WINO_T_COLS(vn);                                                   \
    SMOS_FENCE();                                                      \
    WINO_MFMA_GROUP(v, afa, 2);                                        \
    SMOS_FENCE();                                                      \
    TAIL;                                                              \
    SMOS_FENCE();                                                      \
    WINO_MFMA4(v, afb, 3, 0);                                          \
    SMOS_FENCE();                                                      \
    WINO_VM_WAIT(NVM);                                                 \
    WINO_BARRIER();                                                    \
    WINO_A_READ(afa, so1, 0);                                          \
    SMOS_FENCE();                                                      \
    if constexpr (MB > 1) WINO_MFMA4(v, afb, 3, MB - 1);               \
    SMOS_FENCE();                                                      \
    {                                                                  \
      const int r_ = so0;                                              \
      so0 = so1;                                                       \
      so1 = so2;                                                       \
      so2 = r_;                                                        \
    }                                                                  \
  } while (0)

  // ---- epilogue of an item: A^T M A per (cout, tile) in the lane, bias / residual / activation, 16-byte stores ----
  auto epilogue = [&](const WinoItem& t) {
    const int y = t.y0 + 2 * wave, x = t.x0 + 2 * tx;
    float* srow = nullptr;
    if constexpr (SUMS) {
      const int chunk = (((t.y0 >> 3) * a.xb + (t.x0 >> 5)) << 2) + wave;
      srow = a.sums + ((int64_t)t.b * (a.yb * a.xb * 4) + chunk) * a.cout;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int c0 = (t.ct * MB + mb) * 16 + 4 * q;
      const u32x4 braw = __builtin_amdgcn_raw_buffer_load_b128(bsrd, (unsigned)c0 * 4u, 0, 0);
      const f32x4 bv = {__uint_as_float(braw.x), __uint_as_float(braw.y), __uint_as_float(braw.z), __uint_as_float(braw.w)};
      unsigned ooff[4];
      bool ok[4];
      u32x4 rr[RES ? 4 : 1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = y + (k >> 1), xx = x + (k & 1);
        ok[k] = (yy < a.H) & (xx < a.W);
        const int pix = (t.b * a.H + yy) * a.W + xx;
        ooff[k] = ok[k] ? (unsigned)(pix * (int)a.op + c0) * 4u : 0x80000000u;
        if constexpr (RES) rr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrd, ok[k] ? (unsigned)(pix * (int)a.rp + c0) * 4u : 0x80000000u, 0, 0);
      }
      f32x4 s0[4], s1[4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        s0[nu] = (acc[mb][nu] + acc[mb][4 + nu]) + acc[mb][8 + nu];
        s1[nu] = (acc[mb][4 + nu] - acc[mb][8 + nu]) - acc[mb][12 + nu];
      }
      f32x4 yv[4];
      yv[0] = (s0[0] + s0[1]) + s0[2];
      yv[1] = (s0[1] - s0[2]) - s0[3];
      yv[2] = (s1[0] + s1[1]) + s1[2];
      yv[3] = (s1[1] - s1[2]) - s1[3];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[mb][k] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 ssum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x4 v = yv[k] + bv;
        if constexpr (RES) {
          v[0] += __uint_as_float(rr[k].x); v[1] += __uint_as_float(rr[k].y);
          v[2] += __uint_as_float(rr[k].z); v[3] += __uint_as_float(rr[k].w);
        }
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float o = __builtin_fmaf(a.slope, fminf(v[e], 0.f), fmaxf(v[e], 0.f));
          ov[e] = __float_as_uint(o);
          if constexpr (SUMS) ssum[e] += ok[k] ? o : 0.f;
        }
        if (!WINO_AB(64) || ov.x == 0x7fc12345u) __builtin_amdgcn_raw_buffer_store_b128(ov, osrd, ooff[k], 0, 0);
      }
      if constexpr (SUMS) {
        float4 sv;
        sv.x = row16_sum(ssum[0]);
        sv.y = row16_sum(ssum[1]);
        sv.z = row16_sum(ssum[2]);
        sv.w = row16_sum(ssum[3]);
        if (tx == 15) *reinterpret_cast<float4*>(srow + c0) = sv;
      }
    }
  };

  // ---- prologue: region of chunk 0 in buffer 0, slices 0 and 1 in slots 0 and 1, B operands of k-step 0, first half of
  //      chunk 1's region on its way ----
  auto advance = [&](int& it_, int& c_) {
    if (++c_ == a.nchunk) {
      c_ = 0;
      ++it_;
    }
  };
  int it = 0, c = 0;                 // chunk g
  int it1 = 0, c1 = 0;               // chunk g + 1
  advance(it1, c1);
  WinoItem cur = first_item();
  WinoItem nxt = c1 == 0 ? next_item(cur) : cur;
  int so0 = 0, so1 = kSlot, so2 = 2 * kSlot;
  constexpr bool kEarly = MB == 1;   // region requests one k-step earlier (needs the registers mb = 2 does not have)
  WINO_STAGE_LOAD_A(cur, 0, true);
  WINO_W_LOAD(so0);
  WINO_W_LOAD(so1);
  WINO_STAGE_WRITE_A(lds);
  WINO_STAGE_LOAD_B(cur, 0, true);
  WINO_STAGE_WRITE_B(lds);
  if constexpr (kEarly) WINO_STAGE_LOAD_A(nxt, c1, 1 < total);
  WINO_VM_WAIT(kEarly ? 3 : 0);      // everything but the three requests just issued
  ring_barrier();
  WINO_D_READ(va, lds, 0);
  WINO_T_ROWS(va);
  WINO_T_COLS(va);
  WINO_A_READ(afa, so0, 0);

  float* buf_cur = lds;
  float* buf_nxt = lds + kWInWords;
#pragma unroll 1
  for (int g = 0; g < total; ++g) {
    int it2 = it1, c2 = c1;          // chunk g + 2
    advance(it2, c2);
    const WinoItem nn = c2 == 0 ? next_item(nxt) : nxt;            // past the last item: unused, loads masked off
    // Region of chunk g + 1, through ONE set of three registers: first half requested at the head of k-step 3 of chunk g - 1
    // and stored at the tail of k-step 0, second half requested right behind that store and stored at the tail of k-step 2
    // (about two k-steps of latency each); the barrier that ends k-step 2 publishes the buffer, so that k-step 3 can already
    // read the first patch of chunk g + 1.  hipcc waits vmcnt(0) at the use of an ordinary load while an LDS-DMA is in
    // flight, so in the k-steps with a region store the weight DMA is issued BEHIND the store, otherwise at the head.
    // Last argument: the VMEM requests the wave issues in the k-step itself, i.e. what may still be in flight at its barrier.
    if constexpr (kEarly) {
      WINO_KSTEP(va, vb, buf_cur, 1, (void)0,
                 WINO_STAGE_WRITE_A(buf_nxt); WINO_W_LOAD(so2); WINO_STAGE_LOAD_B(nxt, c1, g + 1 < total), MB + 3);
      WINO_KSTEP(vb, va, buf_cur, 2, WINO_W_LOAD(so2), (void)0, MB + 3);
      WINO_KSTEP(va, vb, buf_cur, 3, (void)0, WINO_STAGE_WRITE_B(buf_nxt); WINO_W_LOAD(so2), MB);
      // (when an item ends with this chunk, the region request goes out behind the epilogue instead of across it; the k-step's
      // HEAD then issues the weight DMA only, so its barrier may leave MB requests in flight, not MB + 3 -- with MB + 3 the
      // slice the next k-step reads could still be on its way)
      if (c1 != 0) {
        WINO_KSTEP(vb, va, buf_nxt, 0, WINO_W_LOAD(so2); WINO_STAGE_LOAD_A(nn, c2, g + 2 < total), (void)0, MB + 3);
      } else {
        WINO_KSTEP(vb, va, buf_nxt, 0, WINO_W_LOAD(so2), (void)0, MB);
        epilogue(cur);
        cur = nxt;
        WINO_STAGE_LOAD_A(nn, c2, g + 2 < total);
      }
    } else {
      // mb = 2 has no register to spare for the longer flight: first half requested at the head of k-step 0 and stored at the
      // tail of k-step 1, second half requested behind that store and stored at the tail of k-step 2
      WINO_KSTEP(va, vb, buf_cur, 1, WINO_W_LOAD(so2); WINO_STAGE_LOAD_A(nxt, c1, g + 1 < total), (void)0, MB + 3);
      WINO_KSTEP(vb, va, buf_cur, 2, (void)0,
                 WINO_STAGE_WRITE_A(buf_nxt); WINO_W_LOAD(so2); WINO_STAGE_LOAD_B(nxt, c1, g + 1 < total), MB + 3);
      WINO_KSTEP(va, vb, buf_cur, 3, (void)0, WINO_STAGE_WRITE_B(buf_nxt); WINO_W_LOAD(so2), MB);
      WINO_KSTEP(vb, va, buf_nxt, 0, WINO_W_LOAD(so2), (void)0, MB);
      if (c1 == 0) {
        epilogue(cur);
        cur = nxt;
      }
    }
    nxt = nn;
    it = it1;
    c = c1;
    it1 = it2;
    c1 = c2;
    float* sw = buf_cur;
    buf_cur = buf_nxt;
    buf_nxt = sw;
  }
  (void)it;
  (void)c;
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
