// Temporal fusion (DeformAttnLayer, networks/multi_view_encoder.py:285-321; MSDeformAttn's projections,
// deformattn/modules/ms_deform_attn.py:94-115) as own MFMA kernels for gfx950.  Per frame and layer the reference runs
//
//     value = value_proj(src);  qp = [sampling_offsets | attention_weights](query);  att = output_proj(sampler(value, qp))
//     query = norm1(query + att);  query = norm2(query + linear2(relu(linear1(query))))
//
// on a token matrix of B * H * W = 16 384 rows x 128 channels: six small GEMMs, two residual + LayerNorm passes and the
// sampler per layer (round 3: thirteen library GEMM launches of ~14 us for 0.5 GFLOP each).  Here:
//
//   tfusion_project   every token-wise Linear that does NOT depend on the previous layer -- value_proj of BOTH layers (they
//                     read the same src) and the first layer's offset / logit projection -- as jobs of ONE launch;
//   (msda_fwd_qp)     the sampler, reading value and qp in place (csrc/msda.hip);
//   tfusion_layer     output_proj -> + query -> LayerNorm -> linear1 -> ReLU -> linear2 -> + -> LayerNorm (-> the NEXT layer's
//                     offset / logit projection) in one kernel: 128 -> 128 -> 512 -> 128 (-> 48); the 512-wide hidden
//                     activations never leave the registers.
//
// Five launches per frame instead of ~18.
//
// Mapping (the transposed chain of point_head.hip / conv_wino.hip on v_mfma_f32_16x16x4_f32): C = W * X with the output
// channel on the MFMA row and the TOKEN on the column.  A wave owns 16 tokens; lane (q = lane >> 4, n = lane & 15) holds
// channels 16 t + 4 q + r (r = 0..3: one float4 of the token's row) of token n for every 16-channel tile t -- which is at
// the same time (a) what a 16-byte row load delivers, (b) the B operand of the four MFMAs (r) that consume tile t and (c)
// the accumulator layout of a 16-channel OUTPUT tile, so a layer's result tiles feed the next layer with no lane movement
// and no LDS.  16 384 tokens = 1 024 waves of 16: one wave per SIMD of the chip, 256 blocks of four waves.
//
// Weights.  A "pair" P(o, t) = the 16 x 16 block W[16 o .. +16][16 t .. +16] as 64 lanes x float4 (lane (q, m) holds
// W[16 o + m][16 t + 4 q + 0..3]): one ds_read_b128 per lane feeds 4 MFMAs (128 matrix cycles).  The host packs each
// layer's weights as the STREAM of pairs in the order the kernel consumes them (ops.tfusion_prepare), cut into slots of 8
// pairs (8 KB): output_proj o = 0..7 (slot = the 8 k-tiles of an output tile), then the FFN one hidden tile of 16 at a time --
// linear1(0), then linear1(j + 1) and linear2(j) alternating (linear1 slot = the tile's 8 k-tiles, linear2 slot = hidden tile j
// as the k-tile of the 8 output tiles), linear2(last) -- then the next layer's query projection.  All four waves of a block consume the same stream in lock step: slot s + 1 is read from
// LDS into registers while the 32 MFMAs of slot s run from registers; slot s + 2 is written to LDS from registers that
// were loaded from global memory two slots earlier.  Two LDS buffers, one barrier per slot (32 MFMAs = 1 024 matrix
// cycles).  600 KB of weights per layer stream once per block: 8 B/clk/CU from L2, 32 B/clk/CU from LDS.
//
// Arithmetic: plain fp32 (exact products, fp32 accumulation in k order per tile), LayerNorm two-pass (mean, then centred
// variance; biased; eps inside the root) like torch's.  Against the float64 formulation: <= 2e-6 of the output range
// (tests/test_gpu_ops.py::test_tfusion_*).
#include "conv_common.h"

// Diagnostic build only (tools/tf_stamps.py, -DSMOS_TF_STAMPS): s_memtime stamps of wave 0 of every block at the phase
// borders of tfusion_layer, written to the buffer whose address the script passes in SMOS_TF_STAMP_PTR.
#ifdef SMOS_TF_STAMPS
#include <stdlib.h>
#define TF_STAMP(k)                                                                             \
  do {                                                                                          \
    if (a.stamps && threadIdx.x == 0) a.stamps[(int64_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define TF_STAMP(k) ((void)0)
#endif

namespace smos {

typedef float tf4 __attribute__((ext_vector_type(4)));

constexpr int kTfC = 128;              // model width (d_model)
constexpr int kTfT = kTfC / 16;        // k-tiles / output tiles of a 128-wide layer
constexpr int kTfSlot = 8 * 64;        // float4 per slot: 8 pairs x 64 lanes
constexpr int kTfSlotBytes = kTfSlot * 16;

// ---- the weight stream: global -> registers (four slots in flight) -> LDS (two buffers) -> fragment registers ----
struct TfStream {
  __amdgpu_buffer_rsrc_t srd;
  unsigned next;        // byte offset of the next slot to request (this thread's first float4 of it)
  u32x4 g[4][2];        // staging registers: set = slot index & 3
};

#define TF_GLOAD(st, set)                                                                        \
  do {                                                                                           \
    (st).g[set][0] = __builtin_amdgcn_raw_buffer_load_b128((st).srd, (st).next, 0, 0);           \
    (st).g[set][1] = __builtin_amdgcn_raw_buffer_load_b128((st).srd, (st).next + 4096u, 0, 0);   \
    (st).next += (unsigned)kTfSlotBytes;                                                         \
  } while (0)
#define TF_PARK(st, set, buf)                                                                    \
  do {                                                                                           \
    float4* d_ = (buf) + threadIdx.x;                                                            \
    d_[0] = make_float4(__uint_as_float((st).g[set][0].x), __uint_as_float((st).g[set][0].y),    \
                        __uint_as_float((st).g[set][0].z), __uint_as_float((st).g[set][0].w));   \
    d_[256] = make_float4(__uint_as_float((st).g[set][1].x), __uint_as_float((st).g[set][1].y),  \
                          __uint_as_float((st).g[set][1].z), __uint_as_float((st).g[set][1].w)); \
  } while (0)

// One slot: barrier (slot s + 1 visible, everybody done with the buffer slot s + 2 goes to), then per pair p: read pair p
// of slot s + 1 into the other fragment set and run the four MFMAs of pair p of slot s (ACC_OF / B_OF name accumulator and B
// tile); then park slot s + 2 and request slot s + 6 (four slots = 4 x 1 024 matrix cycles in flight: an L2 hit takes longer
// than two).  IDX = s & 3 (compile-time): fragment set and LDS buffer = IDX & 1, staging set of slot s + 2 = (IDX + 2) & 3.
// (v_mfma_f32_16x16x4_f32 issues every 32 cycles but a DEPENDENT one needs 40: consecutive MFMAs never share an accumulator.)
// ROW slot: all eight pairs accumulate ONE output tile (output_proj, linear1, the projections): the k sum is split over two
// accumulators by the parity of r, consecutive MFMAs alternate between them; the caller adds the two.
#define TF_SLOT_HEAD(IDX)                                                                        \
  ring_barrier();                                                                                \
  const float4* rd_ = ring + (((IDX) & 1) ^ 1) * kTfSlot + lane
#define TF_SLOT_TAIL(IDX)                                                                        \
  TF_PARK(st, ((IDX) + 2) & 3, ring + ((IDX) & 1) * kTfSlot);                                    \
  TF_GLOAD(st, ((IDX) + 2) & 3)
#define TF_ROW_PAIR(IDX, EA, EB, B_OF, p_)                                                      \
  do {                                                                                           \
    frag[((IDX) & 1) ^ 1][p_] = rd_[(p_) * 64];                                                  \
    const float4 f_ = frag[(IDX) & 1][p_];                                                       \
    EA = __builtin_amdgcn_mfma_f32_16x16x4f32(f_.x, B_OF(p_)[0], EA, 0, 0, 0);                   \
    EB = __builtin_amdgcn_mfma_f32_16x16x4f32(f_.y, B_OF(p_)[1], EB, 0, 0, 0);                   \
    EA = __builtin_amdgcn_mfma_f32_16x16x4f32(f_.z, B_OF(p_)[2], EA, 0, 0, 0);                   \
    EB = __builtin_amdgcn_mfma_f32_16x16x4f32(f_.w, B_OF(p_)[3], EB, 0, 0, 0);                   \
  } while (0)
// (the park + request sit EARLY in the slot: at its end the LDS writes would still be in flight at the next slot's barrier,
// whose s_waitcnt lgkmcnt(0) then exposes their latency once per slot)
#define TF_SLOT_ROW(IDX, EA, EB, B_OF)                                                           \
  do {                                                                                           \
    TF_SLOT_HEAD(IDX);                                                                           \
    TF_ROW_PAIR(IDX, EA, EB, B_OF, 0);                                                           \
    TF_ROW_PAIR(IDX, EA, EB, B_OF, 1);                                                           \
    SMOS_FENCE();                                                                                \
    TF_SLOT_TAIL(IDX);                                                                           \
    SMOS_FENCE();                                                                                \
    _Pragma("unroll") for (int p_ = 2; p_ < 8; ++p_) TF_ROW_PAIR(IDX, EA, EB, B_OF, p_);         \
  } while (0)
// COLUMN slot: pair p accumulates output tile p from ONE B tile (linear2): r-major, so that the eight MFMAs of an r go to eight
// different accumulators.
#define TF_SLOT_COL(IDX, ACC_OF, BT)                                                             \
  do {                                                                                           \
    TF_SLOT_HEAD(IDX);                                                                           \
    frag[((IDX) & 1) ^ 1][0] = rd_[0 * 64];                                                      \
    frag[((IDX) & 1) ^ 1][1] = rd_[1 * 64];                                                      \
    _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_)                                             \
        ACC_OF(p_) = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[(IDX) & 1][p_].x, (BT)[0], ACC_OF(p_), 0, 0, 0); \
    SMOS_FENCE();                                                                                \
    TF_SLOT_TAIL(IDX);                                                                           \
    SMOS_FENCE();                                                                                \
    frag[((IDX) & 1) ^ 1][2] = rd_[2 * 64];                                                      \
    frag[((IDX) & 1) ^ 1][3] = rd_[3 * 64];                                                      \
    _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_)                                             \
        ACC_OF(p_) = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[(IDX) & 1][p_].y, (BT)[1], ACC_OF(p_), 0, 0, 0); \
    frag[((IDX) & 1) ^ 1][4] = rd_[4 * 64];                                                      \
    frag[((IDX) & 1) ^ 1][5] = rd_[5 * 64];                                                      \
    _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_)                                             \
        ACC_OF(p_) = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[(IDX) & 1][p_].z, (BT)[2], ACC_OF(p_), 0, 0, 0); \
    frag[((IDX) & 1) ^ 1][6] = rd_[6 * 64];                                                      \
    frag[((IDX) & 1) ^ 1][7] = rd_[7 * 64];                                                      \
    _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_)                                             \
        ACC_OF(p_) = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[(IDX) & 1][p_].w, (BT)[3], ACC_OF(p_), 0, 0, 0); \
  } while (0)

// prologue of a stream, in two halves so that the caller can put its own loads between them (everything the prologue needs
// is then in flight together instead of one round trip after the other): ISSUE requests slots 0 .. 3; FINISH parks slots 0
// and 1, requests 4 and 5, and leaves the fragments of slot 0 in frag[0] behind a barrier.
#define TF_STREAM_ISSUE(wptr, wbytes)                                                            \
  do {                                                                                           \
    st.srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wptr), 0, (int)(wbytes), 0x00020000); \
    st.next = threadIdx.x * 16u;                                                                 \
    TF_GLOAD(st, 0);                                                                             \
    TF_GLOAD(st, 1);                                                                             \
    TF_GLOAD(st, 2);                                                                             \
    TF_GLOAD(st, 3);                                                                             \
  } while (0)
#define TF_STREAM_FINISH()                                                                       \
  do {                                                                                           \
    TF_PARK(st, 0, ring);                                                                        \
    TF_GLOAD(st, 0);                                                                             \
    TF_PARK(st, 1, ring + kTfSlot);                                                              \
    TF_GLOAD(st, 1);                                                                             \
    ring_barrier();                                                                              \
    _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) frag[0][p_] = ring[p_ * 64 + lane];         \
  } while (0)

// sum over the four lanes that hold one token (n, n + 16, n + 32, n + 48)
__device__ __forceinline__ float token_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// LayerNorm over the 128 channels of each token, in place on the B-operand tiles; gamma / beta: per lane the float4 at
// channel 16 t + 4 q of tile t (LDS)
__device__ __forceinline__ void layer_norm_tiles(tf4 (&x)[kTfT], const float* gamma, const float* beta, int q, float eps) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < kTfT; ++t) s += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
  const float mean = token_sum(s) * (1.0f / kTfC);
  float v = 0.f;
#pragma unroll
  for (int t = 0; t < kTfT; ++t) {
    x[t] -= mean;
    v += (x[t][0] * x[t][0] + x[t][1] * x[t][1]) + (x[t][2] * x[t][2] + x[t][3] * x[t][3]);
  }
  const float rstd = 1.0f / sqrtf(token_sum(v) * (1.0f / kTfC) + eps);
#pragma unroll
  for (int t = 0; t < kTfT; ++t) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + 16 * t + 4 * q);
    const float4 b = *reinterpret_cast<const float4*>(beta + 16 * t + 4 * q);
    x[t][0] = x[t][0] * rstd * g.x + b.x;
    x[t][1] = x[t][1] * rstd * g.y + b.y;
    x[t][2] = x[t][2] * rstd * g.z + b.z;
    x[t][3] = x[t][3] * rstd * g.w + b.w;
  }
}

__device__ __forceinline__ tf4 as_tf4(u32x4 v) {
  return tf4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ u32x4 as_u32x4(tf4 v) {
  return u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
}

// ------------------------------------------------------------------------------------------------------------------
// tfusion_layer
// ------------------------------------------------------------------------------------------------------------------
// params (floats): bo[128] g1[128] be1[128] b1[F] b2[128] g2[128] be2[128] bq[64 (48 used)]
struct TfLayerArgs {
  const float* sampled;    // [tokens, 128] sampler output (pitch 128)
  const float* query;      // [tokens, *] pitch qp_in
  const float4* wstream;   // ops.tfusion_prepare: 8 + 2 * F / 16 + (has_next ? 4 : 0) slots
  const float* params;
  float* out;              // [tokens, *] pitch op: the layer's output query
  float* qp_next;          // [tokens, 48] or null
  int64_t q_pitch, o_pitch;
  int tokens, ffn_tiles, nq;   // F / 16; channels of the next projection (<= 64)
  int w_bytes;
  float eps1, eps2;
#ifdef SMOS_TF_STAMPS
  unsigned long long* stamps;
#endif
};

__global__ __launch_bounds__(256, 1) void tfusion_layer(TfLayerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4* ring = reinterpret_cast<float4*>(lds);                 // 2 x 8 KB
  float* prm = lds + 2 * kTfSlot * 4;                            // parameters
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int F = a.ffn_tiles * 16;
  const int n_prm = 6 * kTfC + F + 64;
  const float* p_bo = prm;
  const float* p_g1 = prm + kTfC;
  const float* p_be1 = prm + 2 * kTfC;
  const float* p_b1 = prm + 3 * kTfC;
  const float* p_b2 = p_b1 + F;
  const float* p_g2 = p_b2 + kTfC;
  const float* p_be2 = p_g2 + kTfC;
  const float* p_bq = p_be2 + kTfC;

  // prologue: the first weight slots, the two input tiles and the parameter block are requested together
  TF_STAMP(0);
  TfStream st;
  float4 frag[2][8];
  TF_STREAM_ISSUE(a.wstream, a.w_bytes);
  const int row = (int)blockIdx.x * 64 + wave * 16 + n;
  const bool live = row < a.tokens;
  const __amdgpu_buffer_rsrc_t ssrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sampled), 0, a.tokens * kTfC * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t qsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.query), 0, (int)(a.tokens * a.q_pitch * 4), 0x00020000);
  const unsigned s_off = live ? (unsigned)(row * kTfC + 4 * q) * 4u : 0x80000000u;
  const unsigned q_off = live ? (unsigned)(row * (int)a.q_pitch + 4 * q) * 4u : 0x80000000u;
  tf4 xs[kTfT], xq[kTfT];                 // B tiles: sampled (then q1), query
#pragma unroll
  for (int t = 0; t < kTfT; ++t) xs[t] = as_tf4(__builtin_amdgcn_raw_buffer_load_b128(ssrd, s_off + 64u * t, 0, 0));
#pragma unroll
  for (int t = 0; t < kTfT; ++t) xq[t] = as_tf4(__builtin_amdgcn_raw_buffer_load_b128(qsrd, q_off + 64u * t, 0, 0));
  {
    const __amdgpu_buffer_rsrc_t psrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.params), 0, n_prm * 4, 0x00020000);
    u32x4 pv[2];                          // n_prm <= 6 * 128 + 4096 + 64 floats: up to 5 float4 per thread; two rounds of up to 2048 floats
    for (int base = 0; base < n_prm; base += 2048) {
      pv[0] = __builtin_amdgcn_raw_buffer_load_b128(psrd, (unsigned)(base + 4 * (int)threadIdx.x) * 4u, 0, 0);
      pv[1] = __builtin_amdgcn_raw_buffer_load_b128(psrd, (unsigned)(base + 1024 + 4 * (int)threadIdx.x) * 4u, 0, 0);
      if (base + 4 * (int)threadIdx.x < n_prm) *reinterpret_cast<float4*>(prm + base + 4 * threadIdx.x) =
          make_float4(__uint_as_float(pv[0].x), __uint_as_float(pv[0].y), __uint_as_float(pv[0].z), __uint_as_float(pv[0].w));
      if (base + 1024 + 4 * (int)threadIdx.x < n_prm) *reinterpret_cast<float4*>(prm + base + 1024 + 4 * threadIdx.x) =
          make_float4(__uint_as_float(pv[1].x), __uint_as_float(pv[1].y), __uint_as_float(pv[1].z), __uint_as_float(pv[1].w));
    }
  }
  TF_STREAM_FINISH();                     // its barrier also publishes prm
  TF_STAMP(1);

  // ---- output_proj: acc1[o] = sum_t P(o, t) xs[t] ----
  tf4 acc1[kTfT], ea, eb;
  const tf4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define TF_B_T(p) xs[p]
#define TF_ROW_OUT(IDX, dst)            \
  do {                                  \
    ea = zero4;                         \
    eb = zero4;                         \
    TF_SLOT_ROW(IDX, ea, eb, TF_B_T);   \
    dst = ea + eb;                      \
  } while (0)
#pragma unroll
  for (int o_ = 0; o_ < kTfT; o_ += 4) {
    TF_ROW_OUT(0, acc1[o_]);
    TF_ROW_OUT(1, acc1[o_ + 1]);
    TF_ROW_OUT(2, acc1[o_ + 2]);
    TF_ROW_OUT(3, acc1[o_ + 3]);
  }
  TF_STAMP(2);
  // + bias + query -> LayerNorm 1 -> q1 (kept in xs: linear1's B operand and the second residual)
#pragma unroll
  for (int t = 0; t < kTfT; ++t) {
    const float4 b = *reinterpret_cast<const float4*>(p_bo + 16 * t + 4 * q);
    xs[t][0] = (acc1[t][0] + b.x) + xq[t][0];
    xs[t][1] = (acc1[t][1] + b.y) + xq[t][1];
    xs[t][2] = (acc1[t][2] + b.z) + xq[t][2];
    xs[t][3] = (acc1[t][3] + b.w) + xq[t][3];
  }
  layer_norm_tiles(xs, p_g1, p_be1, q, a.eps1);

  // ---- FFN: per hidden tile j: h_j = relu(W1[j] q1 + b1[j]) (8 pairs, two interleaved accumulators), acc2[o] += W2[o][j] h_j (8
  //      pairs).  Software-pipelined by one tile: the stream brings linear1(j + 1) BEFORE linear2(j), so that the ReLU of a tile
  //      is taken a whole slot after its last MFMA was issued (no wait for the matrix result in front of linear2) ----
  tf4 acc2[kTfT];
#pragma unroll
  for (int o = 0; o < kTfT; ++o) acc2[o] = tf4{0.f, 0.f, 0.f, 0.f};
  tf4 hsum, hb;
#define TF_ACC_2(p) acc2[p]
#define TF_H_RELU(j)                                                                   \
  do {                                                                                 \
    const float4 b_ = *reinterpret_cast<const float4*>(p_b1 + 16 * (j) + 4 * q);        \
    hb[0] = fmaxf(hsum[0] + b_.x, 0.f);                                                 \
    hb[1] = fmaxf(hsum[1] + b_.y, 0.f);                                                 \
    hb[2] = fmaxf(hsum[2] + b_.z, 0.f);                                                 \
    hb[3] = fmaxf(hsum[3] + b_.w, 0.f);                                                 \
  } while (0)
  TF_STAMP(3);
  TF_ROW_OUT(0, hsum);                           // linear1(0)
  TF_H_RELU(0);
#pragma unroll 1
  for (int j = 0; j + 2 < a.ffn_tiles; j += 2) {   // ffn_tiles is even (host check): slot index & 3 is static
    TF_ROW_OUT(1, hsum);                         // linear1(j + 1)
    TF_SLOT_COL(2, TF_ACC_2, hb);                // linear2(j) from hb
    TF_H_RELU(j + 1);
    TF_ROW_OUT(3, hsum);                         // linear1(j + 2)
    TF_SLOT_COL(0, TF_ACC_2, hb);                // linear2(j + 1)
    TF_H_RELU(j + 2);
  }
  TF_ROW_OUT(1, hsum);                           // linear1(last)
  TF_SLOT_COL(2, TF_ACC_2, hb);                  // linear2(last - 1)
  TF_H_RELU(a.ffn_tiles - 1);
  TF_SLOT_COL(3, TF_ACC_2, hb);                  // linear2(last)
  TF_STAMP(4);
#undef TF_ACC_2
#undef TF_H_RELU
  // + bias + q1 -> LayerNorm 2 -> the layer's output
#pragma unroll
  for (int t = 0; t < kTfT; ++t) {
    const float4 b = *reinterpret_cast<const float4*>(p_b2 + 16 * t + 4 * q);
    xs[t][0] = (acc2[t][0] + b.x) + xs[t][0];
    xs[t][1] = (acc2[t][1] + b.y) + xs[t][1];
    xs[t][2] = (acc2[t][2] + b.z) + xs[t][2];
    xs[t][3] = (acc2[t][3] + b.w) + xs[t][3];
  }
  layer_norm_tiles(xs, p_g2, p_be2, q, a.eps2);
  {
    const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(a.tokens * a.o_pitch * 4), 0x00020000);
    const unsigned o_off = live ? (unsigned)(row * (int)a.o_pitch + 4 * q) * 4u : 0x80000000u;
#pragma unroll
    for (int t = 0; t < kTfT; ++t) __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(xs[t]), osrd, o_off + 64u * t, 0, 0);
  }

  TF_STAMP(5);
  // ---- the next layer's offset / logit projection on the fresh output (4 slots: up to 64 channels, nq stored) ----
  if (a.qp_next) {
    tf4 accq[4];
    TF_ROW_OUT(0, accq[0]);
    TF_ROW_OUT(1, accq[1]);
    TF_ROW_OUT(2, accq[2]);
    TF_ROW_OUT(3, accq[3]);
    const __amdgpu_buffer_rsrc_t nsrd = __builtin_amdgcn_make_buffer_rsrc(a.qp_next, 0, a.tokens * a.nq * 4, 0x00020000);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float4 b = *reinterpret_cast<const float4*>(p_bq + 16 * o + 4 * q);
      const tf4 v = {accq[o][0] + b.x, accq[o][1] + b.y, accq[o][2] + b.z, accq[o][3] + b.w};
      const unsigned off = (live && 16 * o + 4 * q < a.nq) ? (unsigned)(row * a.nq + 16 * o + 4 * q) * 4u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(v), nsrd, off, 0, 0);
    }
  }
  TF_STAMP(6);
#undef TF_B_T
#undef TF_ROW_OUT
}

// ------------------------------------------------------------------------------------------------------------------
// tfusion_project: up to eight independent token-wise Linear jobs (128 -> 16 * tiles) in one launch, blockIdx.y = job
// ------------------------------------------------------------------------------------------------------------------
struct TfJob {
  const float* x;          // [tokens, *] pitch xp
  const float4* wstream;   // tiles slots (cout rounded up to a multiple of 64): slot o = P(o, t = 0..7)
  const float* bias;       // [cout]
  float* out;              // [tokens, *] rows of pitch op
  int64_t xp, op;
  int cout, tiles;         // channels stored; slots streamed (a multiple of 4)
  int tokens;
};
struct TfProjectArgs {
  TfJob job[8];
};

__global__ __launch_bounds__(256, 1) void tfusion_project(TfProjectArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4* ring = reinterpret_cast<float4*>(lds);
  const TfJob& jb = a.job[blockIdx.y];
  if ((int)blockIdx.x * 64 >= jb.tokens) return;          // a shorter job of the launch: the whole block leaves
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, n = lane & 15;
  const int row = (int)blockIdx.x * 64 + wave * 16 + n;
  const bool live = row < jb.tokens;
  const __amdgpu_buffer_rsrc_t xsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.x), 0, (int)(jb.tokens * jb.xp * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(jb.out, 0, (int)(jb.tokens * jb.op * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t bsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.bias), 0, jb.bias ? jb.cout * 4 : 0, 0x00020000);
  const unsigned x_off = live ? (unsigned)(row * (int)jb.xp + 4 * q) * 4u : 0x80000000u;
  tf4 xs[kTfT];
#pragma unroll
  for (int t = 0; t < kTfT; ++t) xs[t] = as_tf4(__builtin_amdgcn_raw_buffer_load_b128(xsrd, x_off + 64u * t, 0, 0));
  TfStream st;
  float4 frag[2][8];
  TF_STREAM_ISSUE(jb.wstream, jb.tiles * kTfSlotBytes);
  TF_STREAM_FINISH();
  tf4 acc, ea, eb;
  const tf4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define TF_ACC_P(p) acc
#define TF_B_T(p) xs[p]
#define TF_EMIT(o)                                                                                                     \
  do {                                                                                                                 \
    const tf4 b_ = as_tf4(__builtin_amdgcn_raw_buffer_load_b128(bsrd, (unsigned)(16 * (o) + 4 * q) * 4u, 0, 0));       \
    const unsigned off_ = (live && 16 * (o) + 4 * q < jb.cout) ? (unsigned)(row * (int)jb.op + 16 * (o) + 4 * q) * 4u : 0x80000000u; \
    __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(acc + b_), osrd, off_, 0, 0);                                      \
  } while (0)
#define TF_ROW_EMIT(IDX, o)             \
  do {                                  \
    ea = zero4;                         \
    eb = zero4;                         \
    TF_SLOT_ROW(IDX, ea, eb, TF_B_T);   \
    acc = ea + eb;                      \
    TF_EMIT(o);                         \
  } while (0)
#pragma unroll 1
  for (int o = 0; o < jb.tiles; o += 4) {
    TF_ROW_EMIT(0, o);
    TF_ROW_EMIT(1, o + 1);
    TF_ROW_EMIT(2, o + 2);
    TF_ROW_EMIT(3, o + 3);
  }
#undef TF_ROW_EMIT
#undef TF_ACC_P
#undef TF_B_T
#undef TF_EMIT
}

}  // namespace smos

using namespace smos;

// Token-wise Linear layers y = W x (+ b) on 128-channel token rows, up to eight jobs in one launch: the projections of a frame's
// temporal fusion that do not depend on a previous layer (value_proj of every DeformAttnLayer and the first layer's
// [sampling_offsets | attention_weights], deformattn/modules/ms_deform_attn.py:94-103), and the decoder's tap products
// (csrc/upconv.hip: [B Hs Ws, 128] x [128, 9 * 128] for the two coarse maps).  Per job: x [tokens[j], *] (row pitch x_pitch
// floats, >= 128), wstream = ops.tfusion_pack_linear(W) (cout rounded up to a multiple of 64, zero padded), bias [cout] or
// NULL, out [tokens[j], *] rows of pitch out_pitch[j] >= cout (a job may write a column range of a wider matrix).  cout a
// multiple of 4, <= 2048; operands < 2 GiB.
extern "C" int smos_tfusion_project(int32_t n_jobs, const float* const* x, const int64_t* x_pitch, const float* const* wstream,
                                    const float* const* bias, float* const* out, const int64_t* out_pitch, const int64_t* cout,
                                    const int64_t* tokens, smos_stream_t stream) {
  SMOS_REQUIRE(n_jobs >= 1 && n_jobs <= 8 && x && x_pitch && wstream && bias && out && out_pitch && cout && tokens,
               "tfusion_project: 1..8 jobs");
  TfProjectArgs a;
  int64_t most = 0;
  for (int j = 0; j < n_jobs; ++j) {
    SMOS_REQUIRE(x[j] && wstream[j] && out[j] && x_pitch[j] >= kTfC && x_pitch[j] % 4 == 0 && cout[j] > 0 && cout[j] % 4 == 0 &&
                     cout[j] <= 2048 && out_pitch[j] >= cout[j] && out_pitch[j] % 4 == 0 && tokens[j] > 0 && tokens[j] < (1LL << 22),
                 "tfusion_project: bad job (pitch >= 128, cout a multiple of 4 and <= 2048, out pitch >= cout, 0 < tokens < 2^22)");
    SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(x[j]) | reinterpret_cast<uintptr_t>(wstream[j]) | reinterpret_cast<uintptr_t>(bias[j]) |
                   reinterpret_cast<uintptr_t>(out[j])) & 15) == 0, "tfusion_project: pointers must be 16-byte aligned");
    SMOS_REQUIRE(tokens[j] * x_pitch[j] * 4 < (1LL << 31) && tokens[j] * out_pitch[j] * 4 < (1LL << 31),
                 "tfusion_project: an operand larger than 2 GiB");
    a.job[j].x = x[j]; a.job[j].wstream = reinterpret_cast<const float4*>(wstream[j]); a.job[j].bias = bias[j]; a.job[j].out = out[j];
    a.job[j].xp = x_pitch[j]; a.job[j].op = out_pitch[j]; a.job[j].cout = (int)cout[j]; a.job[j].tiles = (int)((cout[j] + 63) / 64 * 4);
    a.job[j].tokens = (int)tokens[j];
    most = tokens[j] > most ? tokens[j] : most;
  }
  for (int j = n_jobs; j < 8; ++j) a.job[j] = a.job[0];
  const size_t lds = (size_t)2 * kTfSlotBytes;
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&tfusion_project), lds, 0, &ks, "tfusion_project")) return rc;
  hipLaunchKernelGGL(tfusion_project, dim3((unsigned)((most + 63) / 64), (unsigned)n_jobs), dim3(256), lds, (hipStream_t)stream, a);
  return check_launch("tfusion_project");
}

extern "C" int64_t smos_tfusion_layer_param_floats(int64_t ffn) { return 6 * kTfC + ffn + 64; }
extern "C" int64_t smos_tfusion_layer_stream_floats(int64_t ffn, int32_t has_next) {
  return (int64_t)(8 + 2 * (ffn / 16) + (has_next ? 4 : 0)) * kTfSlot * 4;
}

// One DeformAttnLayer behind its sampler (multi_view_encoder.py:314-320): out = norm2(q1 + linear2(relu(linear1(q1)))) with
// q1 = norm1(query + output_proj(sampled)); optionally also qp_next = [sampling_offsets | attention_weights](out) of the
// NEXT layer (nq channels, <= 64).  sampled [tokens, 128] dense, query [tokens, *] pitch q_pitch, out pitch o_pitch;
// wstream / params = ops.tfusion_prepare(...); d_model 128, ffn a multiple of 16.
extern "C" int smos_tfusion_layer(const float* sampled, const float* query, int64_t q_pitch, const float* wstream, const float* params,
                                  float* out, int64_t o_pitch, float* qp_next, int64_t nq, int64_t tokens, int64_t ffn, float eps1,
                                  float eps2, smos_stream_t stream) {
  SMOS_REQUIRE(tokens > 0 && tokens < (1LL << 22) && ffn >= 32 && ffn % 32 == 0 && ffn <= 4096, "tfusion_layer: bad sizes (ffn a multiple of 32)");
  SMOS_REQUIRE(sampled && query && wstream && params && out && q_pitch >= kTfC && o_pitch >= kTfC && q_pitch % 4 == 0 &&
                   o_pitch % 4 == 0, "tfusion_layer: null pointer / bad pitch");
  SMOS_REQUIRE(!qp_next || (nq > 0 && nq <= 64 && nq % 4 == 0), "tfusion_layer: the next projection has 4..64 channels");
  SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(sampled) | reinterpret_cast<uintptr_t>(query) | reinterpret_cast<uintptr_t>(wstream) |
                 reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(qp_next)) & 15) == 0,
               "tfusion_layer: pointers must be 16-byte aligned");
  SMOS_REQUIRE(tokens * q_pitch * 4 < (1LL << 31) && tokens * o_pitch * 4 < (1LL << 31), "tfusion_layer: a tensor larger than 2 GiB");
  TfLayerArgs a;
  a.sampled = sampled; a.query = query; a.wstream = reinterpret_cast<const float4*>(wstream); a.params = params; a.out = out;
  a.qp_next = qp_next; a.q_pitch = q_pitch; a.o_pitch = o_pitch; a.tokens = (int)tokens; a.ffn_tiles = (int)(ffn / 16);
  a.nq = (int)nq; a.eps1 = eps1; a.eps2 = eps2;
  a.w_bytes = (int)(smos_tfusion_layer_stream_floats(ffn, qp_next != nullptr) * 4);
#ifdef SMOS_TF_STAMPS
  {
    const char* e = getenv("SMOS_TF_STAMP_PTR");
    a.stamps = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr;
  }
#endif
  const size_t lds = (size_t)2 * kTfSlotBytes + (size_t)smos_tfusion_layer_param_floats(ffn) * sizeof(float);
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&tfusion_layer), lds, 0, &ks, "tfusion_layer")) return rc;
  hipLaunchKernelGGL(tfusion_layer, dim3((unsigned)((tokens + 63) / 64)), dim3(256), lds, (hipStream_t)stream, a);
  return check_launch("tfusion_layer");
}
