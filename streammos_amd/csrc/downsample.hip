// DownSample2D's pool branch in one kernel for gfx950 (networks/backbone.py:105-134):
//
//     out = relu( a + bias + maxpool3x3( W x ; stride, pad 1 ) )        W: the 1x1 pool-branch convolution (BatchNorm folded;
//                                                                          both branches' biases sit behind the pool, in `bias`)
//
// with a = the conv-branch result (3x3, stride 1 or 2: smos_conv_cl / smos_conv_wino_cl).  Until round 4 this was two launches --
// the 1x1 convolution at FULL resolution (smos_conv_cl) writing q = W x to HBM, then smos_downsample_epilogue_cl reading q back
// through nine taps per output: for the 64-channel block at 256 x 256 that is 67 MB written and re-read for 17 MB of output.
// Here a block owns a tile of 2 x 16 (stride 2) or 4 x 32 (stride 1) output pixels:
//   phase 1  q of the tile's input region ((2 - 1) * 2 + 3 = 5 rows x 33 columns resp. 6 x 34; 1.29x / 1.59x the pixels a
//            non-overlapping cut would touch) on the matrix cores, pixels as MFMA columns in the transposed form of tfusion.hip
//            (lane (q, n) holds channels 16 t + 4 q + 0..3 of pixel n = one float4 of its row = the B operand of four
//            v_mfma_f32_16x16x4_f32; W as 16 x 16 "pairs" resident in LDS for the block's lifetime), written to an LDS tile
//            [region pixel][COUT + 4];
//   phase 2  every (output pixel, 4 channels) takes the maximum over its window from LDS (taps outside the image do not take
//            part: PyTorch pads a max pool with -inf), adds a and the bias, applies the ReLU and stores 16 bytes.
// q never leaves the CU.  Persistent blocks walk the tiles; x is read 1.3x - 1.6x, a once, out written once.
#include "conv_common.h"

namespace smos {

typedef float pb4 __attribute__((ext_vector_type(4)));

struct PbArgs {
  const float* x;      // [B, H, W, *] pitch xp
  const float4* w;     // pairs [COUT / 16][CIN / 16][64 lanes] float4 (ops._tf_pairs)
  const float* a;      // [B, Ho, Wo, *] pitch ap
  const float* bias;   // [COUT]
  float* out;          // [B, Ho, Wo, *] pitch op
  int64_t xp, ap, op;
  int B, H, W, Ho, Wo, stride;
  int tro, tco;        // output rows / columns per tile
  int rr, rc;          // region rows / columns = (tro - 1) * stride + 3, (tco - 1) * stride + 3
  int tiles_y, tiles_x, n_tiles;
};

template <int KT, int OT>
__global__ __launch_bounds__(256) void pool_branch(PbArgs a) {
  constexpr int CIN = 16 * KT, COUT = 16 * OT, kPitch = COUT + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4* wl = reinterpret_cast<float4*>(lds);                 // OT * KT * 64 float4
  float* qt = lds + OT * KT * 64 * 4;                          // [rr * rc][kPitch]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, n = lane & 15;
  for (int i = threadIdx.x; i < OT * KT * 64; i += 256) wl[i] = a.w[i];
  const __amdgpu_buffer_rsrc_t xsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((int64_t)a.B * a.H * a.W * a.xp * 4), 0x00020000);
  const int rp = a.rr * a.rc, n_tt = (rp + 15) >> 4;
  const pb4 zero4 = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int tile = (int)blockIdx.x; tile < a.n_tiles; tile += (int)gridDim.x) {
    const int tx = tile % a.tiles_x;
    const int ty = (tile / a.tiles_x) % a.tiles_y;
    const int b = tile / (a.tiles_x * a.tiles_y);
    const int ho0 = ty * a.tro, wo0 = tx * a.tco;
    const int y0 = ho0 * a.stride - 1, x0 = wo0 * a.stride - 1;

    // ---- phase 1: q = W x on the region's pixels, 16 at a time per wave; the next token tile's rows are requested before
    //      the current one's MFMAs
    auto x_offset = [&](int tt) {
      const int p = tt * 16 + n;
      const int ry = p / a.rc, rx = p - ry * a.rc;
      const int y = y0 + ry, xx = x0 + rx;
      const bool ok = (p < rp) & (y >= 0) & (y < a.H) & (xx >= 0) & (xx < a.W);
      return ok ? (unsigned)((((int64_t)b * a.H + y) * a.W + xx) * a.xp + 4 * q) * 4u : 0x80000000u;
    };
    pb4 xs[KT], xn[KT];
    {
      const unsigned off = wave < n_tt ? x_offset(wave) : 0x80000000u;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrd, off + 64u * t, 0, 0);
        xs[t] = pb4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    }
    for (int tt = wave; tt < n_tt; tt += 4) {
      {
        const unsigned off = tt + 4 < n_tt ? x_offset(tt + 4) : 0x80000000u;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xsrd, off + 64u * t, 0, 0);
          xn[t] = pb4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
      }
      const int p = tt * 16 + n;
      float* qrow = qt + p * kPitch + 4 * q;
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        pb4 ea = zero4, eb = zero4;          // consecutive MFMAs never share an accumulator (40-cycle dependent latency)
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const float4 f = wl[(o * KT + t) * 64 + lane];
          ea = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, xs[t][0], ea, 0, 0, 0);
          eb = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, xs[t][1], eb, 0, 0, 0);
          ea = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, xs[t][2], ea, 0, 0, 0);
          eb = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, xs[t][3], eb, 0, 0, 0);
        }
        const pb4 r = ea + eb;
        if (p < rp) *reinterpret_cast<float4*>(qrow + 16 * o) = make_float4(r[0], r[1], r[2], r[3]);
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) xs[t] = xn[t];
    }
    __syncthreads();

    // ---- phase 2: window maximum from LDS + conv branch + bias, ReLU
    const int n_items = a.tro * a.tco * (COUT / 4);
    for (int it = threadIdx.x; it < n_items; it += 256) {
      const int c4 = it % (COUT / 4);
      const int op_ = it / (COUT / 4);
      const int orow = op_ / a.tco, ocol = op_ - orow * a.tco;
      const int ho = ho0 + orow, wo = wo0 + ocol;
      if (ho >= a.Ho || wo >= a.Wo) continue;
      float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int ry = orow * a.stride + k / 3, rx = ocol * a.stride + k % 3;
        const int y = y0 + ry, xx = x0 + rx;
        const bool in = (y >= 0) & (y < a.H) & (xx >= 0) & (xx < a.W);
        const float4 v = *reinterpret_cast<const float4*>(qt + (ry * a.rc + rx) * kPitch + 4 * c4);
        m.x = fmaxf(m.x, in ? v.x : -INFINITY); m.y = fmaxf(m.y, in ? v.y : -INFINITY);
        m.z = fmaxf(m.z, in ? v.z : -INFINITY); m.w = fmaxf(m.w, in ? v.w : -INFINITY);
      }
      const int64_t o = ((int64_t)b * a.Ho + ho) * a.Wo + wo;
      const float4 av = *reinterpret_cast<const float4*>(a.a + o * a.ap + 4 * c4);
      const float4 bv = *reinterpret_cast<const float4*>(a.bias + 4 * c4);
      float4 r;
      r.x = fmaxf((av.x + m.x) + bv.x, 0.f); r.y = fmaxf((av.y + m.y) + bv.y, 0.f);
      r.z = fmaxf((av.z + m.z) + bv.z, 0.f); r.w = fmaxf((av.w + m.w) + bv.w, 0.f);
      *reinterpret_cast<float4*>(a.out + o * a.op + 4 * c4) = r;
    }
    __syncthreads();
  }
}

}  // namespace smos

using namespace smos;

template <int KT, int OT>
static int launch_pool_branch(const PbArgs& a, hipStream_t s) {
  const size_t lds = (size_t)OT * KT * 64 * 16 + (size_t)a.rr * a.rc * (16 * OT + 4) * sizeof(float);
  // the dynamic-LDS opt-in is set once per (kernel, device): ask for the largest tile this instantiation can be given (the
  // stride-1 region, 6 x 34 pixels), capped at the CU's 160 KB
  size_t optin = (size_t)OT * KT * 64 * 16 + (size_t)6 * 34 * (16 * OT + 4) * sizeof(float);
  if (optin > 160 * 1024) optin = 160 * 1024;
  if (lds > optin) {
    set_error("downsample_pool_branch: %zu bytes of LDS needed (%d channels at stride %d)", lds, 16 * OT, a.stride);
    return SMOS_ERR_UNSUPPORTED;
  }
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(&pool_branch<KT, OT>), optin, 0, &ks, "downsample_pool_branch")) return rc;
  const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);
  const int64_t cap = (int64_t)ks.cus * per_cu;
  hipLaunchKernelGGL((pool_branch<KT, OT>), dim3((unsigned)(a.n_tiles < cap ? a.n_tiles : cap)), dim3(256), lds, s, a);
  return check_launch("downsample_pool_branch");
}

// out = relu(a + bias + maxpool3x3(conv1x1(x, w); stride, pad 1)): the DownSample2D tail with its pool branch computed on the
// fly (networks/backbone.py:105-134).  x [B, H, W, *] (pitch x_pitch >= Cin), wpairs = the 1x1 weights [Cout, Cin] as 16 x 16
// blocks in MFMA operand order (streammos_amd.ops._tf_pairs), a / out [B, Ho, Wo, *] with Ho = (H - 1) / stride + 1; Cin = Cout
// in {32, 64, 128}; stride 1 or 2; pitches multiples of 4, pointers 16-byte aligned, x below 2 GiB.
extern "C" int smos_downsample_pool_branch(const float* x, int64_t x_pitch, const float* wpairs, const float* a, int64_t a_pitch,
                                           const float* bias, float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W,
                                           int64_t Cin, int64_t Cout, int32_t stride, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && Cin == Cout && (Cin == 32 || Cin == 64 || Cin == 128),
               "downsample_pool_branch: Cin == Cout in {32, 64, 128}, stride 1 or 2");
  SMOS_REQUIRE(x && wpairs && a && bias && out && x_pitch >= Cin && a_pitch >= Cout && out_pitch >= Cout && x_pitch % 4 == 0 &&
                   a_pitch % 4 == 0 && out_pitch % 4 == 0, "downsample_pool_branch: null pointer / bad pitch");
  SMOS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpairs) | reinterpret_cast<uintptr_t>(a) |
                 reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
               "downsample_pool_branch: pointers must be 16-byte aligned");
  SMOS_REQUIRE(B * H * W * x_pitch * 4 < (1LL << 31), "downsample_pool_branch: input larger than 2 GiB");
  PbArgs p;
  p.x = x; p.w = reinterpret_cast<const float4*>(wpairs); p.a = a; p.bias = bias; p.out = out;
  p.xp = x_pitch; p.ap = a_pitch; p.op = out_pitch;
  p.B = (int)B; p.H = (int)H; p.W = (int)W; p.stride = stride;
  p.Ho = (int)((H + 2 - 3) / stride + 1); p.Wo = (int)((W + 2 - 3) / stride + 1);
  p.tro = stride == 2 ? 2 : 4; p.tco = stride == 2 ? 16 : 32;
  p.rr = (p.tro - 1) * stride + 3; p.rc = (p.tco - 1) * stride + 3;
  p.tiles_y = (p.Ho + p.tro - 1) / p.tro; p.tiles_x = (p.Wo + p.tco - 1) / p.tco;
  const int64_t n_tiles = B * p.tiles_y * p.tiles_x;
  SMOS_REQUIRE(n_tiles < (1LL << 31) && B * p.Ho * p.Wo < (1LL << 31), "downsample_pool_branch: too many tiles");
  p.n_tiles = (int)n_tiles;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 32) return launch_pool_branch<2, 2>(p, s);
  if (Cin == 64) return launch_pool_branch<4, 4>(p, s);
  return launch_pool_branch<8, 8>(p, s);
}
