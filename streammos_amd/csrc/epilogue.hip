// Fused elementwise epilogues of the BEV / range-view encoder for gfx950.
//
// The reference runs every conv as conv -> BatchNorm -> ReLU (-> add -> ReLU), each its own pass over
// the activation (networks/backbone.py:14-34,136-159, networks/multi_view_encoder.py:460-497); at
// inference BatchNorm is a per-channel affine map, so its scale is folded into the conv weights on the
// host and what remains of all those passes is ONE read-modify-write per conv output:
//
//   bias_act            out = act(x + bias[c] (+ residual))           BasicBlock / Unbalance / BasicConv2d
//   downsample_epilogue out = relu(a + bias[c] + maxpool3x3_s(p))     DownSample2D (backbone.py:29-34)
//   plane_sum + gate_residual                                          BasicBlock with ChannelAtt (:87-102,151-159)
//   upsample_concat     out = cat(up(x0), up(x1), up(x2))              decoder input (multi_view_encoder.py:441-447)
//
// All of them are pure HBM streaming: float4 accesses, one (batch, channel) plane chunk per block so the
// per-channel constants are wave-uniform, outputs addressed through explicit batch/channel strides so a
// result can land directly inside a channel slice of a concatenation buffer (torch.cat disappears).
#include "smos_common.h"

namespace smos {

enum { kActNone = 0, kActRelu = 1, kActLeaky = 2 };

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == kActRelu) return fmaxf(v, 0.0f);
  if (act == kActLeaky) return v > 0.0f ? v : v * 0.01f;  // nn.LeakyReLU() default slope
  return v;
}

// planes are contiguous (H*W elements); x / res / out may have different batch and channel strides
template <bool kVec>
__global__ __launch_bounds__(kBlock) void bias_act_planes(const float* __restrict__ x, int64_t xs_b, int64_t xs_c,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ res, int64_t rs_b, int64_t rs_c,
                                                          float* __restrict__ out, int64_t os_b, int64_t os_c, int C,
                                                          int64_t HW, int act) {
  const int plane = blockIdx.y;
  const int b = plane / C, c = plane - b * C;
  const float bv = bias ? bias[c] : 0.0f;
  const float* xp = x + b * xs_b + c * xs_c;
  const float* rp = res ? res + b * rs_b + c * rs_c : nullptr;
  float* op = out + b * os_b + c * os_c;
  if (kVec) {
    const int64_t n4 = HW >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      float4 v = reinterpret_cast<const float4*>(xp)[i];
      if (rp) {
        const float4 r = reinterpret_cast<const float4*>(rp)[i];
        v.x = (v.x + bv) + r.x; v.y = (v.y + bv) + r.y; v.z = (v.z + bv) + r.z; v.w = (v.w + bv) + r.w;
      } else {
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
      }
      v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
      reinterpret_cast<float4*>(op)[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
      float v = xp[i] + bv;
      if (rp) v += rp[i];
      op[i] = act_apply(v, act);
    }
  }
}

// out[b,c,ho,wo] = relu(a[b,c,ho,wo] + bias[c] + max over the 3x3 window (stride s, pad 1) of p[b,c,:,:])
// a and p are addressed with full strides (NCHW or channels-last); out planes are contiguous.
__global__ __launch_bounds__(kBlock) void downsample_epilogue(const float* __restrict__ a, Dims4 as,
                                                              const float* __restrict__ p, Dims4 ps,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int64_t os_b, int64_t os_c, int C, int H, int W, int Ho,
                                                              int Wo, int stride) {
  const int plane = blockIdx.y;
  const int b = plane / C, c = plane - b * C;
  const float bv = bias[c];
  const float* ap = a + b * as.v[0] + c * as.v[1];
  const float* pp = p + b * ps.v[0] + c * ps.v[1];
  float* op = out + b * os_b + c * os_c;
  const int total = Ho * Wo;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ho = i / Wo, wo = i - ho * Wo;
    const int h0 = ho * stride - 1, w0 = wo * stride - 1;
    float m = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int h = h0 + dy;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int w = w0 + dx;
        if (w < 0 || w >= W) continue;
        m = fmaxf(m, pp[(int64_t)h * ps.v[2] + (int64_t)w * ps.v[3]]);
      }
    }
    const float v = (ap[(int64_t)ho * as.v[2] + (int64_t)wo * as.v[3]] + m) + bv;
    op[i] = fmaxf(v, 0.0f);
  }
}


// Channels-last variant of downsample_epilogue: a [B,Ho,Wo,C] and p [B,H,W,C] are read with lane = channel
// (contiguous C-float rows), the result is transposed through LDS and written as NCHW planes.
// One block = one output row segment of 64 pixels x 32 channels.
__global__ __launch_bounds__(kBlock) void downsample_epilogue_cl(const float* __restrict__ a, const float* __restrict__ p,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int64_t os_b, int64_t os_c, int C, int H, int W, int Ho,
                                                                 int Wo, int stride) {
  __shared__ float tile[32][65];
  const int c_blocks = C / 32;
  const int cb = blockIdx.y % c_blocks;
  const int b = blockIdx.y / c_blocks;
  const int seg_per_row = (Wo + 63) / 64;
  const int ho = blockIdx.x / seg_per_row;
  const int wo0 = (blockIdx.x - ho * seg_per_row) * 64;
  const int c = cb * 32 + (threadIdx.x & 31);
  const float bv = bias[c];
  const float* ab = a + ((int64_t)b * Ho + ho) * Wo * C + c;
  const float* pb = p + (int64_t)b * H * W * C + c;
  const int h0 = ho * stride - 1;
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int px = pass * 8 + (threadIdx.x >> 5);
    const int wo = wo0 + px;
    float v = 0.0f;
    if (wo < Wo) {
      const int w0 = wo * stride - 1;
      float m = -INFINITY;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int h = h0 + dy;
        if (h < 0 || h >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int w = w0 + dx;
          if (w < 0 || w >= W) continue;
          m = fmaxf(m, pb[((int64_t)h * W + w) * C]);
        }
      }
      v = fmaxf((ab[(int64_t)wo * C] + m) + bv, 0.0f);
    }
    tile[threadIdx.x & 31][px] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wo0 + lane < Wo) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int cc = wv * 8 + r;
      out[(int64_t)b * os_b + (int64_t)(cb * 32 + cc) * os_c + (int64_t)ho * Wo + wo0 + lane] = tile[cc][lane];
    }
  }
}

// sums[b*C + c] = sum over the plane of x (one block per plane; float accumulation per thread, then a
// wave shuffle + LDS reduction)
__global__ __launch_bounds__(kBlock) void plane_sum(const float* __restrict__ x, int64_t xs_b, int64_t xs_c, int C,
                                                    int64_t HW, float* __restrict__ sums) {
  const int plane = blockIdx.x;
  const int b = plane / C, c = plane - b * C;
  const float* xp = x + b * xs_b + c * xs_c;
  float acc = 0.0f;
  const int64_t n4 = HW >> 2;
  for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(xp)[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
  for (int64_t i = (n4 << 2) + threadIdx.x; i < HW; i += blockDim.x) acc += xp[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  __shared__ float part[kBlock / kWave];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int w = 0; w < kBlock / kWave; ++w) s += part[w];
    sums[plane] = s;
  }
}

// Channel attention + residual: g = sigmoid(W2 relu(W1 mean + b1) + b2), out = relu((y + bias) * g + x)
// where mean[c] = sums[b,c]/HW + bias[c].  Every block recomputes the tiny gate MLP for its own channel.
__global__ __launch_bounds__(kBlock) void gate_residual(const float* __restrict__ y, int64_t ys_b, int64_t ys_c,
                                                        const float* __restrict__ bias, const float* __restrict__ sums,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* __restrict__ xres, int64_t rs_b, int64_t rs_c,
                                                        float* __restrict__ out, int64_t os_b, int64_t os_c, int C,
                                                        int Cr, int64_t HW) {
  const int plane = blockIdx.y;
  const int b = plane / C, c = plane - b * C;
  __shared__ float hidden[64];
  __shared__ float gate;
  const float inv = 1.0f / (float)HW;
  for (int j = threadIdx.x; j < Cr; j += blockDim.x) {
    float h = b1[j];
    for (int k = 0; k < C; ++k) h += w1[j * C + k] * (sums[b * C + k] * inv + bias[k]);
    hidden[j] = fmaxf(h, 0.0f);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float g = b2[c];
    for (int j = 0; j < Cr; ++j) g += w2[c * Cr + j] * hidden[j];
    gate = 1.0f / (1.0f + expf(-g));
  }
  __syncthreads();
  const float g = gate, bv = bias[c];
  const float* yp = y + b * ys_b + c * ys_c;
  const float* rp = xres + b * rs_b + c * rs_c;
  float* op = out + b * os_b + c * os_c;
  const int64_t n4 = HW >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(yp)[i];
    const float4 r = reinterpret_cast<const float4*>(rp)[i];
    v.x = fmaxf((v.x + bv) * g + r.x, 0.0f);
    v.y = fmaxf((v.y + bv) * g + r.y, 0.0f);
    v.z = fmaxf((v.z + bv) * g + r.z, 0.0f);
    v.w = fmaxf((v.w + bv) * g + r.w, 0.0f);
    reinterpret_cast<float4*>(op)[i] = v;
  }
}

// Bilinear resize (align_corners=True, ATen's upsample_bilinear2d formula) of up to three NCHW sources to a
// common (Ho, Wo), written side by side along the channel axis of `out` (B, C0+C1+C2, Ho, Wo).
struct UpSrc {
  const float* p;
  int C, H, W;
  int64_t s_b, s_c;
};

__global__ __launch_bounds__(kBlock) void upsample_concat(UpSrc s0, UpSrc s1, UpSrc s2, float* __restrict__ out,
                                                          int Ctot, int Ho, int Wo) {
  const int plane = blockIdx.y;
  const int b = plane / Ctot;
  int c = plane - b * Ctot;
  UpSrc s = s0;
  if (c >= s0.C) {
    c -= s0.C;
    s = s1;
    if (c >= s1.C) {
      c -= s1.C;
      s = s2;
    }
  }
  const float* sp = s.p + b * s.s_b + c * s.s_c;
  float* op = out + (int64_t)plane * Ho * Wo;
  const float rh = Ho > 1 ? (float)(s.H - 1) / (float)(Ho - 1) : 0.0f;
  const float rw = Wo > 1 ? (float)(s.W - 1) / (float)(Wo - 1) : 0.0f;
  const int total = Ho * Wo;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int h2 = i / Wo, w2 = i - h2 * Wo;
    const float h1r = rh * h2, w1r = rw * w2;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = (h1 < s.H - 1) ? 1 : 0, w1p = (w1 < s.W - 1) ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.0f - h1l, w1l = w1r - w1, w0l = 1.0f - w1l;
    const float* r0 = sp + (int64_t)h1 * s.W;
    const float* r1 = r0 + (int64_t)h1p * s.W;
    op[i] = h0l * (w0l * r0[w1] + w1l * r0[w1 + w1p]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + w1p]);
  }
}

}  // namespace smos

using namespace smos;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int smos_bias_act(const float* x, int64_t xs_b, int64_t xs_c, const float* bias, const float* res,
                             int64_t rs_b, int64_t rs_c, float* out, int64_t os_b, int64_t os_c, int64_t B, int64_t C,
                             int64_t HW, int32_t act, smos_stream_t stream) {
  SMOS_REQUIRE(B >= 0 && C >= 0 && HW >= 0 && act >= 0 && act <= 2, "bias_act: bad arguments");
  if (B * C * HW == 0) return SMOS_OK;
  SMOS_REQUIRE(x && out, "bias_act: null pointer");
  SMOS_REQUIRE(B * C <= 65535, "bias_act: more than 65535 planes");
  const bool vec = (HW % 4 == 0) && aligned16(x) && aligned16(out) && (!res || aligned16(res)) && xs_b % 4 == 0 &&
                   xs_c % 4 == 0 && os_b % 4 == 0 && os_c % 4 == 0 && (!res || (rs_b % 4 == 0 && rs_c % 4 == 0));
  const int64_t per_block = kBlock * (vec ? 16 : 4);
  dim3 grid((unsigned)((HW + per_block - 1) / per_block), (unsigned)(B * C));
  if (vec)
    hipLaunchKernelGGL((bias_act_planes<true>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, xs_b, xs_c, bias, res,
                       rs_b, rs_c, out, os_b, os_c, (int)C, HW, act);
  else
    hipLaunchKernelGGL((bias_act_planes<false>), grid, dim3(kBlock), 0, (hipStream_t)stream, x, xs_b, xs_c, bias, res,
                       rs_b, rs_c, out, os_b, os_c, (int)C, HW, act);
  return check_launch("bias_act");
}

extern "C" int smos_downsample_epilogue(const float* a, const int64_t* a_stride, const float* p,
                                        const int64_t* p_stride, const float* bias, float* out, int64_t os_b,
                                        int64_t os_c, int64_t B, int64_t C, int64_t H, int64_t W, int32_t stride,
                                        smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "downsample_epilogue: bad arguments");
  SMOS_REQUIRE(a && p && bias && out && a_stride && p_stride, "downsample_epilogue: null pointer");
  SMOS_REQUIRE(B * C <= 65535, "downsample_epilogue: more than 65535 planes");
  const int Ho = (int)((H + 2 - 3) / stride + 1), Wo = (int)((W + 2 - 3) / stride + 1);
  Dims4 as, ps;
  for (int i = 0; i < 4; ++i) {
    as.v[i] = a_stride[i];
    ps.v[i] = p_stride[i];
  }
  const bool a_cl = a_stride[1] == 1 && a_stride[3] == C && a_stride[2] == (int64_t)Wo * C && a_stride[0] == (int64_t)Ho * Wo * C;
  const bool p_cl = p_stride[1] == 1 && p_stride[3] == C && p_stride[2] == W * C && p_stride[0] == H * W * C;
  if (a_cl && p_cl && C % 32 == 0) {
    dim3 gcl((unsigned)(Ho * ((Wo + 63) / 64)), (unsigned)(B * (C / 32)));
    hipLaunchKernelGGL(downsample_epilogue_cl, gcl, dim3(kBlock), 0, (hipStream_t)stream, a, p, bias, out, os_b, os_c, (int)C,
                       (int)H, (int)W, Ho, Wo, (int)stride);
    return check_launch("downsample_epilogue_cl");
  }
  dim3 grid((unsigned)((Ho * Wo + kBlock * 4 - 1) / (kBlock * 4)), (unsigned)(B * C));
  hipLaunchKernelGGL(downsample_epilogue, grid, dim3(kBlock), 0, (hipStream_t)stream, a, as, p, ps, bias, out, os_b, os_c,
                     (int)C, (int)H, (int)W, Ho, Wo, (int)stride);
  return check_launch("downsample_epilogue");
}

extern "C" int smos_channel_gate_residual(const float* y, int64_t ys_b, int64_t ys_c, const float* bias,
                                          const float* w1, const float* b1, const float* w2, const float* b2,
                                          const float* xres, int64_t rs_b, int64_t rs_c, float* out, int64_t os_b,
                                          int64_t os_c, float* sums_ws, int64_t B, int64_t C, int64_t Cr, int64_t HW,
                                          smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && C > 0 && Cr > 0 && Cr <= 64 && HW > 0 && HW % 4 == 0, "channel_gate_residual: bad arguments");
  SMOS_REQUIRE(y && bias && w1 && b1 && w2 && b2 && xres && out && sums_ws, "channel_gate_residual: null pointer");
  SMOS_REQUIRE(B * C <= 65535, "channel_gate_residual: more than 65535 planes");
  SMOS_REQUIRE(aligned16(y) && aligned16(xres) && aligned16(out) && ys_b % 4 == 0 && ys_c % 4 == 0 && rs_b % 4 == 0 &&
                   rs_c % 4 == 0 && os_b % 4 == 0 && os_c % 4 == 0,
               "channel_gate_residual: planes must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(plane_sum, dim3((unsigned)(B * C)), dim3(kBlock), 0, s, y, ys_b, ys_c, (int)C, HW, sums_ws);
  const int64_t per_block = kBlock * 16;
  dim3 grid((unsigned)((HW + per_block - 1) / per_block), (unsigned)(B * C));
  hipLaunchKernelGGL(gate_residual, grid, dim3(kBlock), 0, s, y, ys_b, ys_c, bias, (const float*)sums_ws, w1, b1, w2, b2,
                     xres, rs_b, rs_c, out, os_b, os_c, (int)C, (int)Cr, HW);
  return check_launch("channel_gate_residual");
}

extern "C" int smos_upsample_concat(const float* const* src, const int64_t* src_c, const int64_t* src_h,
                                    const int64_t* src_w, const int64_t* src_sb, const int64_t* src_sc, int32_t n_src,
                                    float* out, int64_t B, int64_t Ho, int64_t Wo, smos_stream_t stream) {
  SMOS_REQUIRE(n_src >= 1 && n_src <= 3 && B > 0 && Ho > 0 && Wo > 0, "upsample_concat: bad arguments");
  SMOS_REQUIRE(src && src_c && src_h && src_w && src_sb && src_sc && out, "upsample_concat: null pointer");
  UpSrc s[3];
  int ctot = 0;
  for (int i = 0; i < 3; ++i) {
    if (i < n_src) {
      SMOS_REQUIRE(src[i] && src_c[i] > 0 && src_h[i] > 0 && src_w[i] > 0, "upsample_concat: bad source %d", i);
      s[i] = UpSrc{src[i], (int)src_c[i], (int)src_h[i], (int)src_w[i], src_sb[i], src_sc[i]};
      ctot += (int)src_c[i];
    } else {
      s[i] = UpSrc{nullptr, 0, 1, 1, 0, 0};
    }
  }
  SMOS_REQUIRE(B * ctot <= 65535, "upsample_concat: more than 65535 planes");
  dim3 grid((unsigned)((Ho * Wo + kBlock * 4 - 1) / (kBlock * 4)), (unsigned)(B * ctot));
  hipLaunchKernelGGL(upsample_concat, grid, dim3(kBlock), 0, (hipStream_t)stream, s[0], s[1], s[2], out, ctot, (int)Ho,
                     (int)Wo);
  return check_launch("upsample_concat");
}

// ---------------------------------------------------------------------------------------------
// out = LayerNorm(x + res) over the last dimension (DeformAttnLayer, multi_view_encoder.py:314-320: the two residual
// + norm steps of a layer), one wave per token row, C a multiple of 64 up to 512.  Two-pass moments in registers.
// ---------------------------------------------------------------------------------------------
namespace smos {

template <int kPer>
__global__ __launch_bounds__(kBlock) void add_layer_norm(const float* __restrict__ x, const float* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ out, int64_t rows, float eps) {
  constexpr int C = kPer * kWave;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave);
  float g[kPer], bt[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    g[k] = gamma[k * kWave + lane];
    bt[k] = beta[k * kWave + lane];
  }
  for (int64_t r = wave; r < rows; r += n_waves) {
    float v[kPer];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      v[k] = x[r * C + k * kWave + lane] + (res ? res[r * C + k * kWave + lane] : 0.0f);
      s += v[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < kPer; ++k) q += (v[k] - mean) * (v[k] - mean);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int k = 0; k < kPer; ++k) out[r * C + k * kWave + lane] = (v[k] - mean) * rstd * g[k] + bt[k];
  }
}

}  // namespace smos

extern "C" int smos_add_layer_norm(const float* x, const float* res, const float* gamma, const float* beta, float* out,
                                   int64_t rows, int64_t C, float eps, smos_stream_t stream) {
  using namespace smos;
  SMOS_REQUIRE(rows >= 0 && C > 0 && C % kWave == 0 && C <= 512, "add_layer_norm: C must be a multiple of 64 up to 512");
  if (rows == 0) return SMOS_OK;
  SMOS_REQUIRE(x && gamma && beta && out, "add_layer_norm: null pointer");
  const dim3 grid(grid_for(rows * kWave, kBlock, 256 * 16)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  switch (C / kWave) {
    case 1: hipLaunchKernelGGL(add_layer_norm<1>, grid, block, 0, s, x, res, gamma, beta, out, rows, eps); break;
    case 2: hipLaunchKernelGGL(add_layer_norm<2>, grid, block, 0, s, x, res, gamma, beta, out, rows, eps); break;
    case 4: hipLaunchKernelGGL(add_layer_norm<4>, grid, block, 0, s, x, res, gamma, beta, out, rows, eps); break;
    case 8: hipLaunchKernelGGL(add_layer_norm<8>, grid, block, 0, s, x, res, gamma, beta, out, rows, eps); break;
    default:
      set_error("add_layer_norm: C = %lld is not built (64, 128, 256, 512)", (long long)C);
      return SMOS_ERR_UNSUPPORTED;
  }
  return check_launch("add_layer_norm");
}
