// Channels-last ("cl", [B, H, W, C] with C innermost) variants of the engine's elementwise and point kernels
// for gfx950.  Under MIOpen's solver search every conv of the network is as fast or faster in channels-last
// (tools/ubench_conv_layout.py: 4.82 vs 5.63 ms per scan summed over the layer list), the scatter targets are
// channels-last by construction (one contiguous row of C floats per cell = one full-rate row atomic), and a
// [B, H*W, C] map IS the token layout of the deformable-attention block, so the engine keeps every feature map in
// this layout and nothing is ever transposed.
//
// A "row" below is the C channels of one pixel / point; every tensor argument carries its own row pitch in
// elements, so a tensor may be a channel slice of a wider buffer (concatenations are written in place).
#include <stdlib.h>

#include "smos_common.h"

namespace smos {

__device__ __forceinline__ float act1(float v, int act) {
  if (act == 1) return fmaxf(v, 0.0f);
  if (act == 2) return v > 0.0f ? v : v * 0.01f;
  return v;
}
__device__ __forceinline__ float4 act4(float4 v, int act) {
  return make_float4(act1(v.x, act), act1(v.y, act), act1(v.z, act), act1(v.w, act));
}

// out[p, c] = act(x[p, c] + bias[c] (+ res[p, c]));  C % 4 == 0, pitches % 4 == 0
__global__ __launch_bounds__(kBlock) void bias_act_cl(const float* __restrict__ x, int64_t xp, const float* __restrict__ bias,
                                                      const float* __restrict__ res, int64_t rp, float* __restrict__ out,
                                                      int64_t op, int64_t P, int C4, int act) {
  const int64_t total = P * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / C4;
    const int q = (int)(i - p * C4) * 4;
    float4 v = *reinterpret_cast<const float4*>(x + p * xp + q);
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + q);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + p * rp + q);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    *reinterpret_cast<float4*>(out + p * op + q) = act4(v, act);
  }
}

// DownSample2D tail, all channels-last: out = relu(a + bias + maxpool3x3(p; stride, pad 1))
__global__ __launch_bounds__(kBlock) void downsample_epilogue_cl2(const float* __restrict__ a, int64_t ap,
                                                                  const float* __restrict__ p, int64_t pp,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int64_t op, int B, int C4, int H, int W, int Ho, int Wo,
                                                                  int stride) {
  // the nine window loads are unconditional (a tap outside the map reads a clamped address and is replaced by -inf):
  // under "if (outside) continue" each one was an exec-masked branch followed by s_waitcnt vmcnt(0) -- nine serialised
  // memory round trips per output.  32-bit index arithmetic (element count checked by the host).
  const int total = B * Ho * Wo * C4;
  for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < total; i += (int)(gridDim.x * blockDim.x)) {
    const int q = (i % C4) * 4;
    int t = i / C4;
    const int wo = t % Wo;
    t /= Wo;
    const int ho = t % Ho;
    const int b = t / Ho;
    const int h0 = ho * stride - 1, w0 = wo * stride - 1;
    float4 v[9];
    bool in[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int h = h0 + k / 3, w = w0 + k % 3;
      in[k] = (h >= 0) & (h < H) & (w >= 0) & (w < W);
      v[k] = *reinterpret_cast<const float4*>(p + ((int64_t)(b * H + min(max(h, 0), H - 1)) * W + min(max(w, 0), W - 1)) * pp + q);
    }
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      m.x = fmaxf(m.x, in[k] ? v[k].x : -INFINITY); m.y = fmaxf(m.y, in[k] ? v[k].y : -INFINITY);
      m.z = fmaxf(m.z, in[k] ? v[k].z : -INFINITY); m.w = fmaxf(m.w, in[k] ? v[k].w : -INFINITY);
    }
    const int64_t o = ((int64_t)b * Ho + ho) * Wo + wo;
    const float4 av = *reinterpret_cast<const float4*>(a + o * ap + q);
    const float4 bv = *reinterpret_cast<const float4*>(bias + q);
    float4 r;
    r.x = fmaxf((av.x + m.x) + bv.x, 0.f); r.y = fmaxf((av.y + m.y) + bv.y, 0.f);
    r.z = fmaxf((av.z + m.z) + bv.z, 0.f); r.w = fmaxf((av.w + m.w) + bv.w, 0.f);
    *reinterpret_cast<float4*>(out + o * op + q) = r;
  }
}

// ---- channel attention: deterministic column sums -> gate MLP -> gated residual -------------------------
// partial[b][chunk][c] = sum over the chunk's pixels of y[b, p, c].  A block is (256 / (C/4)) pixel lanes x C/4
// float4 columns; fixed chunking and a fixed in-block order keep the sums run-to-run deterministic.
__global__ __launch_bounds__(kBlock) void colsum_partial(const float* __restrict__ y, int64_t yp, int C4, int64_t HW,
                                                         int64_t per, int chunks, float* __restrict__ partial) {
  __shared__ float4 red[kBlock];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int q = threadIdx.x % C4, pl = threadIdx.x / C4, npl = kBlock / C4;
  const int64_t p0 = (int64_t)chunk * per, p1 = min(HW, p0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t p = p0 + pl; p < p1; p += npl) {
    const float4 v = *reinterpret_cast<const float4*>(y + ((int64_t)b * HW + p) * yp + q * 4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < npl; ++k) {
      const float4 v = red[k * C4 + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + ((int64_t)b * chunks + chunk) * C4 * 4 + q * 4) = s;
  }
}

// gate[b][c] = sigmoid(w2 relu(w1 (mean + bias) + b1) + b2)   (one block per batch element)
__global__ __launch_bounds__(kBlock) void gate_mlp(const float* __restrict__ partial, int chunks, const float* __restrict__ bias,
                                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2, int C, int Cr,
                                                   int64_t HW, float* __restrict__ gate) {
  __shared__ float mean[256];
  __shared__ float part_sum[kBlock];
  __shared__ float hidden[64];
  const int b = blockIdx.x;
  {
    // the chunk sums of a channel are added by 256 / C threads in a fixed interleaved order (deterministic), instead of
    // one thread walking all chunks serially (that was 15 us of dependent loads on the block's critical path)
    const int parts = kBlock / C, c = threadIdx.x % C, part = threadIdx.x / C;
    float s = 0.0f;
    if (part < parts)
      for (int k = part; k < chunks; k += parts) s += partial[((int64_t)b * chunks + k) * C + c];
    part_sum[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
      float t = 0.0f;
      for (int q = 0; q < parts; ++q) t += part_sum[q * C + threadIdx.x];
      mean[threadIdx.x] = t / (float)HW + bias[threadIdx.x];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < Cr; j += blockDim.x) {
    float h = b1[j];
    for (int k = 0; k < C; ++k) h += w1[j * C + k] * mean[k];
    hidden[j] = fmaxf(h, 0.0f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float g = b2[c];
    for (int j = 0; j < Cr; ++j) g += w2[c * Cr + j] * hidden[j];
    gate[b * C + c] = 1.0f / (1.0f + expf(-g));
  }
}

// The same gate from the per-row-segment channel sums that smos_conv_cl leaves behind (chan_sums: [B][chunks][C], four times
// as many chunks as colsum_partial makes): 1024 threads per sample, the chunks of a channel shared by 1024 / C of them in a
// fixed interleaved order, then the same two-layer MLP.
__global__ __launch_bounds__(1024) void gate_mlp_wide(const float* __restrict__ partial, int chunks, const float* __restrict__ bias,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2, int C, int Cr,
                                                      int64_t HW, float* __restrict__ gate) {
  __shared__ float mean[256];
  __shared__ float part_sum[1024];
  __shared__ float hidden[64];
  const int b = blockIdx.x;
  const int parts = 1024 / C, c = threadIdx.x % C, part = threadIdx.x / C;
  // eight independent chains per thread, their loads issued together: one block reads up to 256 KB here, and a single
  // chain per thread is a string of dependent memory latencies (measured: the step got SLOWER than with the extra pass)
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* src = partial + (int64_t)b * chunks * C + c;
  int k = part;
  for (; k + 7 * parts < chunks; k += 8 * parts) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u * parts) * C];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += v[u];
  }
  for (int u = 0; k < chunks; k += parts, ++u) s[u & 7] += src[(int64_t)k * C];
  part_sum[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.0f;
    for (int q = 0; q < parts; ++q) t += part_sum[q * C + threadIdx.x];
    mean[threadIdx.x] = t / (float)HW + bias[threadIdx.x];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < Cr; j += blockDim.x) {
    float h = b1[j];
    for (int q = 0; q < C; ++q) h += w1[j * C + q] * mean[q];
    hidden[j] = fmaxf(h, 0.0f);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < C; q += blockDim.x) {
    float g = b2[q];
    for (int j = 0; j < Cr; ++j) g += w2[q * Cr + j] * hidden[j];
    gate[b * C + q] = 1.0f / (1.0f + expf(-g));
  }
}

// out = relu((y + bias) * gate[b] + xres)
__global__ __launch_bounds__(kBlock) void gate_apply_cl(const float* __restrict__ y, int64_t yp, const float* __restrict__ bias,
                                                        const float* __restrict__ gate, const float* __restrict__ xres,
                                                        int64_t rp, float* __restrict__ out, int64_t op, int64_t HW, int B,
                                                        int C4) {
  const int64_t total = (int64_t)B * HW * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / C4;
    const int q = (int)(i - p * C4) * 4;
    const int b = (int)(p / HW);
    const float4 v = *reinterpret_cast<const float4*>(y + p * yp + q);
    const float4 bv = *reinterpret_cast<const float4*>(bias + q);
    const float4 g = *reinterpret_cast<const float4*>(gate + (int64_t)b * C4 * 4 + q);
    const float4 r = *reinterpret_cast<const float4*>(xres + p * rp + q);
    float4 o;
    o.x = fmaxf((v.x + bv.x) * g.x + r.x, 0.f); o.y = fmaxf((v.y + bv.y) * g.y + r.y, 0.f);
    o.z = fmaxf((v.z + bv.z) * g.z + r.z, 0.f); o.w = fmaxf((v.w + bv.w) * g.w + r.w, 0.f);
    *reinterpret_cast<float4*>(out + p * op + q) = o;
  }
}

// ---- decoder input: bilinear (align_corners=True, ATen formula) resize + concat, channels-last --------
struct UpSrcCl {
  const float* p;
  int C, H, W;
  int64_t pitch;
};

__global__ __launch_bounds__(kBlock) void upsample_concat_cl(UpSrcCl s0, UpSrcCl s1, UpSrcCl s2, float* __restrict__ out,
                                                             int B, int Ctot, int Ho, int Wo) {
  const int C4 = Ctot / 4;
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int q = (int)(i % C4) * 4;
    int64_t t = i / C4;
    const int w2 = (int)(t % Wo);
    t /= Wo;
    const int h2 = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float* o = out + (((int64_t)b * Ho + h2) * Wo + w2) * Ctot + q;
    UpSrcCl s = s0;
    if (q >= s0.C) {
      q -= s0.C;
      s = s1;
      if (q >= s1.C) {
        q -= s1.C;
        s = s2;
      }
    }
    const float rh = Ho > 1 ? (float)(s.H - 1) / (float)(Ho - 1) : 0.0f;
    const float rw = Wo > 1 ? (float)(s.W - 1) / (float)(Wo - 1) : 0.0f;
    const float h1r = rh * h2, w1r = rw * w2;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = (h1 < s.H - 1) ? 1 : 0, w1p = (w1 < s.W - 1) ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.0f - h1l, w1l = w1r - w1, w0l = 1.0f - w1l;
    const float* base = s.p + (((int64_t)b * s.H + h1) * s.W + w1) * s.pitch + q;
    const float4 v00 = *reinterpret_cast<const float4*>(base);
    const float4 v01 = *reinterpret_cast<const float4*>(base + (int64_t)w1p * s.pitch);
    const float4 v10 = *reinterpret_cast<const float4*>(base + (int64_t)h1p * s.W * s.pitch);
    const float4 v11 = *reinterpret_cast<const float4*>(base + ((int64_t)h1p * s.W + w1p) * s.pitch);
    float4 r;
    r.x = h0l * (w0l * v00.x + w1l * v01.x) + h1l * (w0l * v10.x + w1l * v11.x);
    r.y = h0l * (w0l * v00.y + w1l * v01.y) + h1l * (w0l * v10.y + w1l * v11.y);
    r.z = h0l * (w0l * v00.z + w1l * v01.z) + h1l * (w0l * v10.z + w1l * v11.z);
    r.w = h0l * (w0l * v00.w + w1l * v01.w) + h1l * (w0l * v10.w + w1l * v11.w);
    *reinterpret_cast<float4*>(o) = r;
  }
}

// ---- bilinear gather from a channels-last map fused with the max scatter into another channels-last map ----
// A group of kC lanes (kC = C = 32 or 64) owns a run of kC consecutive points.  Phase A: lane j prepares point j
// (tap offsets, weights, target cell) -- coalesced coordinate reads, the float32 position arithmetic of
// bilinear_gather.hip done once per point.  Phase B: the group walks its points; lane = channel; per point four
// contiguous row reads, one optional row store of the gathered feature, and a running maximum that is flushed with one
// row atomic whenever the target cell changes (azimuth-ordered LiDAR points mostly stay in the same cell).
struct GsClArgs {
  const float* grid;    // [B, Hg, Wg, *] row pitch gp (channel offset already applied)
  const float* gcoord;  // sample b, point n at gcoord + b * gbs + n * Kg (dense [B, N, Kg]: gbs = N * Kg)
  const float* scoord;  // likewise with sbs, Ks; or null
  float* out;           // [B, Ho, Wo, *] row pitch op (channel offset applied), zero-filled; or null
  float* pts_out;       // [B, N, *] row pitch po_n (channel offset applied); or null
  const int32_t* n_live; // device, or null: point rows of the padding tail [*n_live, N) are not wanted (gather_scatter_cl4 only)
  int64_t gp, op, po_b, po_n, gbs, sbs;
  int B, N, Kg, Ks, Hg, Wg, Ho, Wo;
  float gsy, gsx, ssy, ssx;
};

__device__ __forceinline__ float pix_cl(float c, float s, int size) {
  const float sm1 = (float)(size - 1);
  const float gn = __fsub_rn(__fdiv_rn(__fmul_rn(__fmul_rn(2.0f, c), s), sm1), 1.0f);
  return __fmul_rn(__fdiv_rn(__fadd_rn(gn, 1.0f), 2.0f), sm1);
}

// value of lane j of the caller's group.  kC = 64: the group is the wave and j is wave-uniform, so a scalar lane read
// (v_readlane, result in an SGPR) replaces the LDS shuffle: 0.082 -> 0.070 ms per launch.  kC = 32 (two groups per wave)
// keeps the shuffle: two lane reads + a select measured slower (0.067 vs 0.045 ms).
template <int kC>
__device__ __forceinline__ int group_read(int v, int j) {
  if (kC == kWave) return __builtin_amdgcn_readlane(v, j);
  return __shfl(v, j, kC);
}

template <int kC>
__global__ __launch_bounds__(kBlock) void gather_scatter_cl(GsClArgs a) {
  constexpr int kGroups = kBlock / kC;
  const int lane = threadIdx.x % kC;
  const int runs_per_sample = (a.N + kC - 1) / kC;
  const int64_t n_runs = (int64_t)a.B * runs_per_sample;
  for (int64_t run = (int64_t)blockIdx.x * kGroups + threadIdx.x / kC; run < n_runs; run += (int64_t)gridDim.x * kGroups) {
    const int b = (int)(run / runs_per_sample);
    const int n0 = (int)(run - (int64_t)b * runs_per_sample) * kC;
    const int n = n0 + lane;
    // ---- phase A: lane j <-> point j
    int off[4] = {-1, -1, -1, -1};
    float wt[4] = {0.f, 0.f, 0.f, 0.f};
    int cell = -1;
    if (n < a.N) {
      const float* cr = a.gcoord + (int64_t)b * a.gbs + (int64_t)n * a.Kg;
      const float iy = pix_cl(cr[0], a.gsy, a.Hg), ix = pix_cl(cr[1], a.gsx, a.Wg);
      const float fy = floorf(iy), fx = floorf(ix);
      const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
      const bool fin = (iy > -2.0f) && (iy < (float)(a.Hg + 1)) && (ix > -2.0f) && (ix < (float)(a.Wg + 1));
      const int y0 = fin ? (int)fy : -5, x0 = fin ? (int)fx : -5;
      const float w4[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int y = y0 + (k >> 1), xx = x0 + (k & 1);
        const bool in = (y >= 0) && (y < a.Hg) && (xx >= 0) && (xx < a.Wg);
        off[k] = in ? y * a.Wg + xx : -1;
        wt[k] = in ? w4[k] : 0.0f;
      }
      if (a.scoord) {
        const float* sr = a.scoord + (int64_t)b * a.sbs + (int64_t)n * a.Ks;
        const float py = __fmul_rn(sr[0], a.ssy), px = __fmul_rn(sr[1], a.ssx);
        const bool ok = (py > -1.0f) && (py < (float)a.Ho) && (px > -1.0f) && (px < (float)a.Wo);
        cell = ok ? (int)py * a.Wo + (int)px : -1;
      }
    }
    // ---- phase B: lane = channel
    const float* gb = a.grid + (int64_t)b * a.Hg * a.Wg * a.gp + lane;
    float* ob = a.out ? a.out + (int64_t)b * a.Ho * a.Wo * a.op + lane : nullptr;
    float* pb = a.pts_out ? a.pts_out + (int64_t)b * a.po_b + (int64_t)n0 * a.po_n + lane : nullptr;
    const int n_valid = min(kC, a.N - n0);
    int cur = -1;
    float best = 0.0f;
    // fixed trip count, four points per iteration: the 16 row loads of a group are issued before the first use (the
    // loop used to be one dependent shuffle -> load -> fma -> store chain per point).  An absent tap reads row 0 and is
    // discarded by the select, so the value is exactly the sum over the valid taps, in tap order, as before.
    constexpr int kU = 4;
    for (int j0 = 0; j0 < kC; j0 += kU) {
      float g[kU][4], w[kU][4];
      int o[kU][4], c[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[u][k] = group_read<kC>(off[k], j0 + u);
          w[u][k] = __int_as_float(group_read<kC>(__float_as_int(wt[k]), j0 + u));
        }
        c[u] = group_read<kC>(cell, j0 + u);
      }
      // scatter-only launch: a group of points without a target cell (the padding tail of a scan) needs no gather
      if (!pb) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < kU; ++u) any |= c[u] >= 0;
        if (!any) continue;
      }
      // a group whose points all lie outside the source map (padding again) gathers nothing: its rows are zeros
      bool any_tap = false;
#pragma unroll
      for (int u = 0; u < kU; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) any_tap |= o[u][k] >= 0;
      if (any_tap) {
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) g[u][k] = gb[(int64_t)max(o[u][k], 0) * a.gp];
      } else {
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) g[u][k] = 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int j = j0 + u;
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) v = o[u][k] >= 0 ? v + g[u][k] * w[u][k] : v;
        if (j < n_valid) {
          if (pb) pb[(int64_t)j * a.po_n] = v;
          if (ob) {
            if (c[u] != cur) {
              if (cur >= 0 && best > 0.0f) atomicMax(reinterpret_cast<int*>(ob + (int64_t)cur * a.op), __float_as_int(best));
              cur = c[u];
              best = 0.0f;
            }
            best = fmaxf(best, v);
          }
        }
      }
    }
    if (ob && cur >= 0 && best > 0.0f) atomicMax(reinterpret_cast<int*>(ob + (int64_t)cur * a.op), __float_as_int(best));
  }
}


// The same function with 16-byte lanes for the GATHER half (round 4).  A row of C channels is read by C / 4 lanes as float4,
// so a wave instruction serves G = 256 / C points (C = 64: four 16-lane groups; C = 32: eight 8-lane groups); group g walks
// the C / 4 consecutive points [g * C / 4, (g + 1) * C / 4) of the wave's 64-point run.  Per point the one-lane-per-channel
// kernel above issues 9 scalar lane reads, their scalar address arithmetic (one scalar unit per CU) and 4 row loads per
// wave; here a wave instruction serves G points: 9 LDS-crossbar shuffles, 4 loads, 16 multiply-adds, one 16-byte row store.
// The SCATTER half stays one lane per channel -- a row atomic of 16-byte lanes is four instructions that each touch every
// fourth word of the row (measured: 2 - 3x slower than the scalar-lane kernel) -- so the gathered rows of the run go through
// a wave-private LDS tile ([64 points][C], written as float4, read back one word per lane: conflict-free both ways) and a
// second sweep with lane = channel keeps the running maximum per target cell and flushes it with ONE dense row atomic when
// the cell changes, exactly as above.  The value of a point is the same expression (taps in order, absent taps skipped)
// and the scatter is a maximum, so the results equal the old kernel's bit for bit.  (16-byte aligned rows required.)
template <int kC, bool kScatter>
__global__ __launch_bounds__(kBlock) void gather_scatter_cl4(GsClArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gs_lds[];
  constexpr int kL = kC / 4;            // lanes per row = points per group
  constexpr int kWavesPerBlock = kBlock / kWave;
  const int lane = threadIdx.x & 63;
  const int l = lane % kL, grp = lane / kL;
  float* tile = gs_lds + (kScatter ? (threadIdx.x >> 6) * 64 * kC : 0);      // this wave's [64][kC]
  const int runs_per_sample = (a.N + 63) / 64;
  const int64_t n_runs = (int64_t)a.B * runs_per_sample;
  const int n_live = a.n_live ? min(max(*a.n_live, 0), a.N) : a.N;
  for (int64_t run = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); run < n_runs; run += (int64_t)gridDim.x * kWavesPerBlock) {
    const int b = (int)(run / runs_per_sample);
    const int n0 = (int)(run - (int64_t)b * runs_per_sample) * 64;
    const int n = n0 + lane;
    // ---- phase A: lane j <-> point j (identical to gather_scatter_cl)
    int off[4] = {-1, -1, -1, -1};
    float wt[4] = {0.f, 0.f, 0.f, 0.f};
    int cell = -1;
    if (n < a.N) {
      const float* cr = a.gcoord + (int64_t)b * a.gbs + (int64_t)n * a.Kg;
      const float iy = pix_cl(cr[0], a.gsy, a.Hg), ix = pix_cl(cr[1], a.gsx, a.Wg);
      const float fy = floorf(iy), fx = floorf(ix);
      const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
      const bool fin = (iy > -2.0f) && (iy < (float)(a.Hg + 1)) && (ix > -2.0f) && (ix < (float)(a.Wg + 1));
      const int y0 = fin ? (int)fy : -5, x0 = fin ? (int)fx : -5;
      const float w4[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int y = y0 + (k >> 1), xx = x0 + (k & 1);
        const bool in = (y >= 0) && (y < a.Hg) && (xx >= 0) && (xx < a.Wg);
        off[k] = in ? y * a.Wg + xx : -1;
        wt[k] = in ? w4[k] : 0.0f;
      }
      if (kScatter) {
        const float* sr = a.scoord + (int64_t)b * a.sbs + (int64_t)n * a.Ks;
        const float py = __fmul_rn(sr[0], a.ssy), px = __fmul_rn(sr[1], a.ssx);
        const bool ok = (py > -1.0f) && (py < (float)a.Ho) && (px > -1.0f) && (px < (float)a.Wo);
        cell = ok ? (int)py * a.Wo + (int)px : -1;
      }
    }
    // a run without a tap inside the source map (the padding tail of a scan) gathers zeros: nothing to add to a zero-filled
    // target, and its point rows are zeros
    const bool any_tap = __builtin_amdgcn_ballot_w64((off[0] >= 0) | (off[1] >= 0) | (off[2] >= 0) | (off[3] >= 0)) != 0;
    const bool any_cell = !kScatter || __builtin_amdgcn_ballot_w64(cell >= 0) != 0;
    // ---- phase B: lane = (point group, 4 channels)
    const float* gb = a.grid + (int64_t)b * a.Hg * a.Wg * a.gp + 4 * l;
    float* pb = a.pts_out ? a.pts_out + (int64_t)b * a.po_b + (int64_t)n0 * a.po_n + 4 * l : nullptr;
    if (!any_tap) {
      if (pb && n0 < n_live) {
        for (int i = 0; i < kL; ++i) {
          const int j = grp * kL + i;
          if (n0 + j < a.N) *reinterpret_cast<float4*>(pb + (int64_t)j * a.po_n) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      continue;
    }
    if (!pb && !any_cell) continue;       // scatter-only launch, no point of the run has a target cell
    constexpr int kU = 4;
#pragma unroll 1
    for (int i0 = 0; i0 < kL; i0 += kU) {
      float4 g[kU][4];
      float w[kU][4];
      int o[kU][4];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int src = grp * kL + i0 + u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[u][k] = __shfl(off[k], src);
          w[u][k] = __shfl(wt[k], src);
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) g[u][k] = *reinterpret_cast<const float4*>(gb + (int64_t)max(o[u][k], 0) * a.gp);
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int j = grp * kL + i0 + u;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool has = o[u][k] >= 0;
          v.x = has ? v.x + g[u][k].x * w[u][k] : v.x;
          v.y = has ? v.y + g[u][k].y * w[u][k] : v.y;
          v.z = has ? v.z + g[u][k].z * w[u][k] : v.z;
          v.w = has ? v.w + g[u][k].w * w[u][k] : v.w;
        }
        if (pb && n0 + j < a.N) *reinterpret_cast<float4*>(pb + (int64_t)j * a.po_n) = v;
        if (kScatter) *reinterpret_cast<float4*>(tile + j * kC + 4 * l) = v;
      }
    }
    if (kScatter) {
      // ---- phase C: lane = channel; kC = 32: the two halves of the wave sweep the two halves of the run
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int kHalves = kWave / kC;                 // 1 or 2
      const int ch = lane % kC, half = lane / kC;
      float* ob = a.out + (int64_t)b * a.Ho * a.Wo * a.op + ch;
      const int n_valid = min(64, a.N - n0);
      int cur = -1;
      float best = 0.0f;
#pragma unroll 4
      for (int i = 0; i < 64 / kHalves; ++i) {
        const int j = half * (64 / kHalves) + i;
        const int c = kHalves == 1 ? __builtin_amdgcn_readlane(cell, i) : __shfl(cell, j);
        const float v = tile[j * kC + ch];
        if (j < n_valid) {
          if (c != cur) {
            if (cur >= 0 && best > 0.0f) atomicMax(reinterpret_cast<int*>(ob + (int64_t)cur * a.op), __float_as_int(best));
            cur = c;
            best = 0.0f;
          }
          best = fmaxf(best, v);
        }
      }
      if (cur >= 0 && best > 0.0f) atomicMax(reinterpret_cast<int*>(ob + (int64_t)cur * a.op), __float_as_int(best));
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the tile is rewritten by the next run
    }
  }
}

}  // namespace smos

using namespace smos;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int smos_bias_act_cl(const float* x, int64_t x_pitch, const float* bias, const float* res, int64_t res_pitch,
                                float* out, int64_t out_pitch, int64_t P, int64_t C, int32_t act, smos_stream_t stream) {
  SMOS_REQUIRE(P >= 0 && C > 0 && C % 4 == 0 && act >= 0 && act <= 2, "bias_act_cl: bad arguments (C %% 4 must be 0)");
  if (P == 0) return SMOS_OK;
  SMOS_REQUIRE(x && out && al16(x) && al16(out) && x_pitch % 4 == 0 && out_pitch % 4 == 0 && (!bias || al16(bias)) &&
                   (!res || (al16(res) && res_pitch % 4 == 0)), "bias_act_cl: pointers / pitches must be 16-byte aligned");
  hipLaunchKernelGGL(bias_act_cl, dim3(grid_for(P * (C / 4), kBlock, 256 * 16)), dim3(kBlock), 0, (hipStream_t)stream, x, x_pitch,
                     bias, res, res_pitch, out, out_pitch, P, (int)(C / 4), act);
  return check_launch("bias_act_cl");
}

// Zero fill of up to four channels-last views (rows x row_floats at a row pitch) in ONE launch: the scatter targets of a
// frame's two cross-view transfers (two dense range-view maps, two channel slices of concatenation buffers).
struct ZeroViews {
  float* p[4];
  int64_t rows[4], pitch[4];
  int f4[4];          // float4 per row
};

__global__ __launch_bounds__(kBlock) void zero_views(ZeroViews a) {
  const int v = blockIdx.y;
  const int64_t total = a.rows[v] * a.f4[v];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / a.f4[v];
    const int c = (int)(i - r * a.f4[v]);
    *reinterpret_cast<float4*>(a.p[v] + r * a.pitch[v] + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

extern "C" int smos_zero_views_cl(int32_t n, float* const* ptrs, const int64_t* rows, const int64_t* row_floats, const int64_t* pitches,
                                  smos_stream_t stream) {
  SMOS_REQUIRE(n >= 1 && n <= 4 && ptrs && rows && row_floats && pitches, "zero_views_cl: 1 .. 4 views");
  ZeroViews a;
  int64_t most = 0;
  for (int v = 0; v < 4; ++v) {
    const int k = v < n ? v : 0;
    SMOS_REQUIRE(ptrs[k] && al16(ptrs[k]) && rows[k] >= 0 && row_floats[k] > 0 && row_floats[k] % 4 == 0 && pitches[k] >= row_floats[k] &&
                     pitches[k] % 4 == 0, "zero_views_cl: views must be 16-byte aligned rows of a multiple of 4 floats");
    a.p[v] = ptrs[k];
    a.rows[v] = v < n ? rows[k] : 0;
    a.pitch[v] = pitches[k];
    a.f4[v] = (int)(row_floats[k] / 4);
    const int64_t t = a.rows[v] * a.f4[v];
    most = t > most ? t : most;
  }
  if (most == 0) return SMOS_OK;
  hipLaunchKernelGGL(zero_views, dim3(grid_for(most, kBlock, 256 * 8), n), dim3(kBlock), 0, (hipStream_t)stream, a);
  return check_launch("zero_views_cl");
}

extern "C" int smos_downsample_epilogue_cl(const float* a, int64_t a_pitch, const float* p, int64_t p_pitch, const float* bias,
                                           float* out, int64_t out_pitch, int64_t B, int64_t C, int64_t H, int64_t W,
                                           int32_t stride, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "downsample_epilogue_cl: bad arguments");
  SMOS_REQUIRE(a && p && bias && out && al16(a) && al16(p) && al16(bias) && al16(out) && a_pitch % 4 == 0 && p_pitch % 4 == 0 &&
                   out_pitch % 4 == 0, "downsample_epilogue_cl: pointers / pitches must be 16-byte aligned");
  const int Ho = (int)((H + 2 - 3) / stride + 1), Wo = (int)((W + 2 - 3) / stride + 1);
  SMOS_REQUIRE(B * Ho * Wo * (C / 4) < kMaxTotal32 && B * H < (1LL << 31), "downsample_epilogue_cl: too many elements for 32-bit indices");
  hipLaunchKernelGGL(downsample_epilogue_cl2, dim3(grid_for(B * Ho * Wo * (C / 4), kBlock, 256 * 16)), dim3(kBlock), 0,
                     (hipStream_t)stream, a, a_pitch, p, p_pitch, bias, out, out_pitch, (int)B, (int)(C / 4), (int)H, (int)W, Ho, Wo,
                     (int)stride);
  return check_launch("downsample_epilogue_cl");
}

extern "C" int smos_channel_gate_residual_cl(const float* y, int64_t y_pitch, const float* bias, const float* w1, const float* b1,
                                             const float* w2, const float* b2, const float* xres, int64_t res_pitch, float* out,
                                             int64_t out_pitch, float* ws, int64_t ws_floats, int64_t B, int64_t C, int64_t Cr,
                                             int64_t HW, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && C <= 256 && kBlock % C == 0 && Cr > 0 && Cr <= 64 && HW > 0,
               "channel_gate_residual_cl: bad sizes (C must divide 256)");
  SMOS_REQUIRE(y && bias && w1 && b1 && w2 && b2 && xres && out && ws, "channel_gate_residual_cl: null pointer");
  SMOS_REQUIRE(al16(y) && al16(xres) && al16(out) && al16(bias) && al16(ws) && y_pitch % 4 == 0 && res_pitch % 4 == 0 &&
                   out_pitch % 4 == 0, "channel_gate_residual_cl: 16-byte alignment required");
  const int64_t per = 512;
  const int chunks = (int)((HW + per - 1) / per);
  SMOS_REQUIRE(ws_floats >= B * C * (chunks + 1), "channel_gate_residual_cl: workspace too small (%lld floats needed)",
               (long long)(B * C * (chunks + 1)));
  float* partial = ws;
  float* gate = ws + B * C * chunks;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial, dim3(chunks, (unsigned)B), dim3(kBlock), 0, s, y, y_pitch, (int)(C / 4), HW, per, chunks, partial);
  hipLaunchKernelGGL(gate_mlp, dim3((unsigned)B), dim3(kBlock), 0, s, (const float*)partial, chunks, bias, w1, b1, w2, b2, (int)C,
                     (int)Cr, HW, gate);
  hipLaunchKernelGGL(gate_apply_cl, dim3(grid_for(B * HW * (C / 4), kBlock, 256 * 16)), dim3(kBlock), 0, s, y, y_pitch, bias,
                     (const float*)gate, xres, res_pitch, out, out_pitch, HW, (int)B, (int)(C / 4));
  return check_launch("channel_gate_residual_cl");
}

extern "C" int smos_channel_gate_apply_cl(const float* y, int64_t y_pitch, const float* bias, const float* w1, const float* b1,
                                          const float* w2, const float* b2, const float* xres, int64_t res_pitch, float* out,
                                          int64_t out_pitch, const float* chan_sums, int64_t chunks, float* gate_ws, int64_t B,
                                          int64_t C, int64_t Cr, int64_t HW, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && C <= 256 && 1024 % C == 0 && Cr > 0 && Cr <= 64 && HW > 0 && chunks > 0 &&
                   chunks < (1LL << 24), "channel_gate_apply_cl: bad sizes (C must divide 1024, C <= 256, Cr <= 64)");
  SMOS_REQUIRE(y && bias && w1 && b1 && w2 && b2 && xres && out && chan_sums && gate_ws, "channel_gate_apply_cl: null pointer");
  SMOS_REQUIRE(al16(y) && al16(xres) && al16(out) && al16(bias) && al16(gate_ws) && y_pitch % 4 == 0 && res_pitch % 4 == 0 &&
                   out_pitch % 4 == 0, "channel_gate_apply_cl: 16-byte alignment required");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gate_mlp_wide, dim3((unsigned)B), dim3(1024), 0, s, chan_sums, (int)chunks, bias, w1, b1, w2, b2, (int)C,
                     (int)Cr, HW, gate_ws);
  hipLaunchKernelGGL(gate_apply_cl, dim3(grid_for(B * HW * (C / 4), kBlock, 256 * 16)), dim3(kBlock), 0, s, y, y_pitch, bias,
                     (const float*)gate_ws, xres, res_pitch, out, out_pitch, HW, (int)B, (int)(C / 4));
  return check_launch("channel_gate_apply_cl");
}

extern "C" int smos_upsample_concat_cl(const float* const* src, const int64_t* src_c, const int64_t* src_h, const int64_t* src_w,
                                       const int64_t* src_pitch, int32_t n_src, float* out, int64_t B, int64_t Ho, int64_t Wo,
                                       smos_stream_t stream) {
  SMOS_REQUIRE(n_src >= 1 && n_src <= 3 && B > 0 && Ho > 0 && Wo > 0 && src && src_c && src_h && src_w && src_pitch && out,
               "upsample_concat_cl: bad arguments");
  UpSrcCl s[3];
  int ctot = 0;
  for (int i = 0; i < 3; ++i) {
    if (i < n_src) {
      SMOS_REQUIRE(src[i] && al16(src[i]) && src_c[i] > 0 && src_c[i] % 4 == 0 && src_pitch[i] % 4 == 0, "upsample_concat_cl: bad source %d", i);
      s[i] = UpSrcCl{src[i], (int)src_c[i], (int)src_h[i], (int)src_w[i], src_pitch[i]};
      ctot += (int)src_c[i];
    } else {
      s[i] = UpSrcCl{nullptr, 1 << 30, 1, 1, 0};
    }
  }
  hipLaunchKernelGGL(upsample_concat_cl, dim3(grid_for(B * Ho * Wo * (ctot / 4), kBlock, 256 * 16)), dim3(kBlock), 0,
                     (hipStream_t)stream, s[0], s[1], s[2], out, (int)B, ctot, (int)Ho, (int)Wo);
  return check_launch("upsample_concat_cl");
}

extern "C" int smos_gather_scatter_cl_view(const float* grid, int64_t grid_pitch, const float* gcoord, int32_t Kg, int64_t g_batch_stride,
                                           const float* gscale, const float* scoord, int32_t Ks, int64_t s_batch_stride,
                                           const float* sscale, float* out, int64_t out_pitch, float* pts_out, int64_t po_b,
                                           int64_t po_n, int64_t B, int64_t C, int64_t Hg, int64_t Wg, int64_t N, int64_t Ho,
                                           int64_t Wo, const int32_t* n_live, smos_stream_t stream);

extern "C" int smos_gather_scatter_cl(const float* grid, int64_t grid_pitch, const float* gcoord, int32_t Kg, const float* gscale,
                                      const float* scoord, int32_t Ks, const float* sscale, float* out, int64_t out_pitch,
                                      float* pts_out, int64_t po_b, int64_t po_n, int64_t B, int64_t C, int64_t Hg, int64_t Wg,
                                      int64_t N, int64_t Ho, int64_t Wo, smos_stream_t stream) {
  return smos_gather_scatter_cl_live(grid, grid_pitch, gcoord, Kg, gscale, scoord, Ks, sscale, out, out_pitch, pts_out, po_b, po_n, B, C,
                                     Hg, Wg, N, Ho, Wo, nullptr, stream);
}

// n_live (device int32, may be null): the first *n_live points of every sample are real; point rows of the padding tail that lie
// wholly outside the source map are not written (they would be zeros nobody reads).  Everything else as smos_gather_scatter_cl.
extern "C" int smos_gather_scatter_cl_live(const float* grid, int64_t grid_pitch, const float* gcoord, int32_t Kg, const float* gscale,
                                           const float* scoord, int32_t Ks, const float* sscale, float* out, int64_t out_pitch,
                                           float* pts_out, int64_t po_b, int64_t po_n, int64_t B, int64_t C, int64_t Hg, int64_t Wg,
                                           int64_t N, int64_t Ho, int64_t Wo, const int32_t* n_live, smos_stream_t stream) {
  return smos_gather_scatter_cl_view(grid, grid_pitch, gcoord, Kg, N * Kg, gscale, scoord, Ks, N * Ks, sscale, out, out_pitch, pts_out,
                                     po_b, po_n, B, C, Hg, Wg, N, Ho, Wo, n_live, stream);
}

// The coordinates as strided views: sample b, point n at coord + b * batch_stride + n * K (floats; the first two of a point's K
// values are read) -- the engine passes pcds_coord[:, 0, :, :, 0] of the reference's [B, T, N, 3, 1] tensor as it lies, without a
// compacting copy.  Everything else as smos_gather_scatter_cl_live.
extern "C" int smos_gather_scatter_cl_view(const float* grid, int64_t grid_pitch, const float* gcoord, int32_t Kg, int64_t g_batch_stride,
                                           const float* gscale, const float* scoord, int32_t Ks, int64_t s_batch_stride,
                                           const float* sscale, float* out, int64_t out_pitch, float* pts_out, int64_t po_b,
                                           int64_t po_n, int64_t B, int64_t C, int64_t Hg, int64_t Wg, int64_t N, int64_t Ho,
                                           int64_t Wo, const int32_t* n_live, smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && (C == 32 || C == 64) && N > 0 && Hg > 0 && Wg > 0 && Kg >= 2, "gather_scatter_cl: bad sizes (C must be 32 or 64)");
  SMOS_REQUIRE(g_batch_stride >= 0 && (!out || s_batch_stride >= 0), "gather_scatter_cl: negative coordinate stride");
  SMOS_REQUIRE(grid && gcoord && gscale && (out || pts_out) && grid_pitch >= C, "gather_scatter_cl: null pointer / bad pitch");
  SMOS_REQUIRE(!out || (scoord && sscale && Ks >= 2 && Ho > 0 && Wo > 0 && out_pitch >= C && Ho * Wo < (1LL << 31)),
               "gather_scatter_cl: bad scatter target");
  SMOS_REQUIRE(!pts_out || po_n >= C, "gather_scatter_cl: point row pitch smaller than C");
  SMOS_REQUIRE(Hg * Wg < (1LL << 31), "gather_scatter_cl: grid too large");
  GsClArgs a;
  a.grid = grid; a.gcoord = gcoord; a.scoord = out ? scoord : nullptr; a.out = out; a.pts_out = pts_out; a.n_live = n_live;
  a.gp = grid_pitch; a.op = out_pitch; a.po_b = po_b; a.po_n = po_n; a.gbs = g_batch_stride; a.sbs = s_batch_stride;
  a.B = (int)B; a.N = (int)N; a.Kg = Kg; a.Ks = Ks; a.Hg = (int)Hg; a.Wg = (int)Wg; a.Ho = (int)Ho; a.Wo = (int)Wo;
  a.gsy = gscale[0]; a.gsx = gscale[1]; a.ssy = out ? sscale[0] : 0.f; a.ssx = out ? sscale[1] : 0.f;
  // 16-byte lanes wherever the rows are 16-byte aligned (every call of the engine); SMOS_GS_VEC=0: the one-lane-per-channel
  // kernel (A/B, same results)
  static const bool want_vec = [] {
    const char* e = getenv("SMOS_GS_VEC");
    return !(e && e[0] == '0');
  }();
  const bool vec = want_vec && al16(grid) && grid_pitch % 4 == 0 && (!out || (al16(out) && out_pitch % 4 == 0)) &&
                   (!pts_out || (al16(pts_out) && po_n % 4 == 0 && po_b % 4 == 0));
  if (vec) {
    const int64_t runs = B * ((N + 63) / 64);
    const int64_t blocks = (runs + 3) / 4;
    dim3 g((unsigned)(blocks < 256 * 32 ? blocks : 256 * 32));
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = out ? (size_t)4 * 64 * C * sizeof(float) : 0;       // one [64 points][C] tile per wave
    KernelSetup ks;
    if (C == 32 && out) {
      if (int rc = kernel_setup(reinterpret_cast<const void*>(&gather_scatter_cl4<32, true>), lds, 0, &ks, "gather_scatter_cl")) return rc;
      hipLaunchKernelGGL((gather_scatter_cl4<32, true>), g, dim3(kBlock), lds, s, a);
    } else if (C == 32) {
      hipLaunchKernelGGL((gather_scatter_cl4<32, false>), g, dim3(kBlock), 0, s, a);
    } else if (out) {
      if (int rc = kernel_setup(reinterpret_cast<const void*>(&gather_scatter_cl4<64, true>), lds, 0, &ks, "gather_scatter_cl")) return rc;
      hipLaunchKernelGGL((gather_scatter_cl4<64, true>), g, dim3(kBlock), lds, s, a);
    } else {
      hipLaunchKernelGGL((gather_scatter_cl4<64, false>), g, dim3(kBlock), 0, s, a);
    }
    return check_launch("gather_scatter_cl");
  }
  const int64_t runs = B * ((N + C - 1) / C);
  const int64_t blocks = (runs * C + kBlock - 1) / kBlock;
  dim3 g((unsigned)(blocks < 256 * 32 ? blocks : 256 * 32));
  if (C == 32)
    hipLaunchKernelGGL((gather_scatter_cl<32>), g, dim3(kBlock), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((gather_scatter_cl<64>), g, dim3(kBlock), 0, (hipStream_t)stream, a);
  return check_launch("gather_scatter_cl");
}
