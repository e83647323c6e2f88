// Multi-scale deformable attention sampler (forward) for gfx950.
// Replaces ms_deformable_im2col_gpu_kernel (deformattn/src/cuda/ms_deform_im2col_cuda.cuh:237-299) and
// ms_deform_attn_im2col_bilinear (:33-84).
//
// Lane mapping: lane = channel inside one (batch, query, head).  With the model's D = 32 a wavefront
// covers two heads of a query; every tap is a contiguous 128-byte row of the value tensor
// [S, M, D].  The (location, weight) pairs of a head are the same for all of its D lanes: in the
// D == 32 specialisation each 32-lane half loads them once (lane p < L*P takes sample p) and the
// bilinear corner offsets / weights are broadcast with wavefront shuffles instead of being re-read
// and re-derived by every lane.  The kernel for any other D, L, P (and for float64) works the same way with a whole
// wave per head (msda_fwd_heads).
#include "smos_common.h"

namespace smos {

// Any D, L, P, float or double: one WAVE per (batch, query, head).  Lane s of the wave owns sample s of the head (its
// level, location, attention weight, the four corner offsets and bilinear weights -- computed once per head, not once per
// output channel), lane c owns channel c: the sample loop broadcasts a sample's eight numbers with wavefront shuffles
// and every tap is one contiguous row of D values of the [S, M, D] value tensor.  Heads wider than 64 channels take
// several passes over the channels, heads with more than 64 samples several chunks of samples.  Per output element the
// arithmetic is that of the reference kernel (ms_deform_im2col_cuda.cuh:237-299): ((w1 v1 + w2 v2) + w3 v3 + w4 v4) * a,
// summed over the samples in order; a sample outside (-1, H) x (-1, W) contributes nothing, a corner outside the map 0.
template <typename T>
__global__ __launch_bounds__(kBlock) void msda_fwd_heads(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                         const int64_t* __restrict__ lsi, const T* __restrict__ loc,
                                                         const T* __restrict__ attn, T* __restrict__ out,
                                                         int64_t n_heads_total, int S, int M, int D, int L, int Lq, int P) {
  const int lane = threadIdx.x & 63;
  const int LP = L * P;
  const int64_t wstride = (int64_t)M * D;
  for (int64_t hq = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; hq < n_heads_total;
       hq += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const int m = (int)(hq % M);
    const int64_t b = hq / ((int64_t)M * Lq);
    const T* vb = value + (b * S) * wstride + (int64_t)m * D;
    for (int c0 = 0; c0 < D; c0 += 64) {
      const int c = c0 + lane;
      T col = 0;
      for (int s0 = 0; s0 < LP; s0 += 64) {
        T cw[4] = {0, 0, 0, 0}, aw = 0;
        int co[4] = {-1, -1, -1, -1};
        const int smp = s0 + lane;
        if (smp < LP) {
          const int l = smp / P;
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
          const T loc_w = loc[(hq * LP + smp) * 2], loc_h = loc[(hq * LP + smp) * 2 + 1];
          aw = attn[hq * LP + smp];
          const T h_im = loc_h * H - (T)0.5, w_im = loc_w * W - (T)0.5;
          if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
            const int y0 = (int)floor(h_im), x0 = (int)floor(w_im);
            const T fy = h_im - y0, fx = w_im - x0, gy = 1 - fy, gx = 1 - fx;
            const int base = (int)lsi[l];
            const bool top = y0 >= 0, bot = y0 + 1 <= H - 1, lft = x0 >= 0, rgt = x0 + 1 <= W - 1;
            co[0] = (top && lft) ? base + y0 * W + x0 : -1;
            co[1] = (top && rgt) ? base + y0 * W + x0 + 1 : -1;
            co[2] = (bot && lft) ? base + (y0 + 1) * W + x0 : -1;
            co[3] = (bot && rgt) ? base + (y0 + 1) * W + x0 + 1 : -1;
            cw[0] = gy * gx; cw[1] = gy * fx; cw[2] = fy * gx; cw[3] = fy * fx;
          }
        }
        const int n = LP - s0 < 64 ? LP - s0 : 64;
        for (int q = 0; q < n; ++q) {
          T tap = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int o = __shfl(co[k], q);
            const T w = __shfl(cw[k], q);
            const T v = (o >= 0 && c < D) ? vb[(int64_t)o * wstride + c] : (T)0;
            tap += w * v;
          }
          col += tap * __shfl(aw, q);
        }
      }
      if (c < D) out[hq * D + c] = col;
    }
  }
}

// D == 32, L*P <= 8, float32: one 32-lane half per (b, q, m); shuffle-broadcast sample metadata.
// Arithmetic per output element is the same expression tree as msda_fwd_heads.
__global__ __launch_bounds__(kBlock) void msda_fwd_d32(const float* __restrict__ value,
                                                       const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                       const float* __restrict__ attn, float* __restrict__ out,
                                                       int64_t n_heads_total, int S, int M, int L, int Lq, int P) {
  const int lane32 = threadIdx.x & 31;
  const int LP = L * P;
  const int64_t wstride = (int64_t)M * 32;
  for (int64_t hq = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5; hq < n_heads_total;
       hq += ((int64_t)gridDim.x * blockDim.x) >> 5) {
    const int m = (int)(hq % M);
    const int64_t b = hq / ((int64_t)M * Lq);
    // lane p of the half owns sample p: its level, location, weight, corner offsets and weights
    float cw[4] = {0.f, 0.f, 0.f, 0.f};
    int co[4] = {-1, -1, -1, -1};
    float aw = 0.f;
    if (lane32 < LP) {
      const int l = lane32 / P;
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float loc_w = loc[(hq * LP + lane32) * 2], loc_h = loc[(hq * LP + lane32) * 2 + 1];
      aw = attn[hq * LP + lane32];
      const float h_im = loc_h * H - 0.5f, w_im = loc_w * W - 0.5f;
      if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
        const int base = (int)lsi[l];
        const bool t = h_low >= 0, bt = h_low + 1 <= H - 1, lf = w_low >= 0, rt = w_low + 1 <= W - 1;
        co[0] = (t && lf) ? base + h_low * W + w_low : -1;
        co[1] = (t && rt) ? base + h_low * W + w_low + 1 : -1;
        co[2] = (bt && lf) ? base + (h_low + 1) * W + w_low : -1;
        co[3] = (bt && rt) ? base + (h_low + 1) * W + w_low + 1 : -1;
        cw[0] = hh * hw; cw[1] = hh * lw; cw[2] = lh * hw; cw[3] = lh * lw;
      }
    }
    const float* vb = value + (b * S) * wstride + (int64_t)m * 32 + lane32;
    float col = 0.f;
    const int half_base = threadIdx.x & 32;  // shuffle sources live in this lane's half of the wave
    for (int p = 0; p < LP; ++p) {
      const int src = half_base + p;
      // same expression tree as the reference: (w1*v1 + w2*v2 + w3*v3 + w4*v4) * weight, absent taps = 0
      float tap = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o = __shfl(co[k], src);
        const float w = __shfl(cw[k], src);
        const float v = (o >= 0) ? vb[(int64_t)o * wstride] : 0.f;
        tap += w * v;
      }
      col += tap * __shfl(aw, src);
    }
    out[hq * 32 + lane32] = col;
  }
}

// The same sampler fed by the RAW query projection of DeformAttnLayer (multi_view_encoder.py:300-316): per token a row
// qp = [M*P*2 offsets | M*P attention logits]; single level H x W, reference points = cell centres, D == 32, P <= 8.
// Folds what the module does in five small launches -- softmax over the P logits, off / (W, H), + reference point --
// into the sampler: lane p of a 32-lane half owns sample p.  loc = ref + off / norm and (loc * size - 0.5) keep the
// module's operation order.
__global__ __launch_bounds__(kBlock) void msda_fwd_qp_d32(const float* __restrict__ value, const float* __restrict__ qp,
                                                          float* __restrict__ out, int64_t n_heads_total, int M, int H, int W,
                                                          int P) {
  const int lane32 = threadIdx.x & 31;
  const int S = H * W, row = M * P * 3;
  const int64_t wstride = (int64_t)M * 32;
  for (int64_t hq = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5; hq < n_heads_total;
       hq += ((int64_t)gridDim.x * blockDim.x) >> 5) {
    const int m = (int)(hq % M);
    const int64_t tok = hq / M;                      // b * S + q
    const int q = (int)(tok % S);
    const int64_t b = tok / S;
    const float* r = qp + tok * row;
    float logit = -INFINITY, off_x = 0.f, off_y = 0.f;
    if (lane32 < P) {
      off_x = r[(m * P + lane32) * 2];
      off_y = r[(m * P + lane32) * 2 + 1];
      logit = r[M * P * 2 + m * P + lane32];
    }
    // softmax over the P samples of this (token, head): lanes >= P hold -inf / 0
    float mx = logit;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 8));
    const float e = lane32 < P ? expf(logit - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 8);
    const float aw = e / sum;
    float cw[4] = {0.f, 0.f, 0.f, 0.f};
    int co[4] = {-1, -1, -1, -1};
    if (lane32 < P) {
      const float ref_x = ((float)(q % W) + 0.5f) / (float)W, ref_y = ((float)(q / W) + 0.5f) / (float)H;
      const float loc_w = ref_x + off_x / (float)W, loc_h = ref_y + off_y / (float)H;
      const float h_im = loc_h * H - 0.5f, w_im = loc_w * W - 0.5f;
      if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
        const bool t = h_low >= 0, bt = h_low + 1 <= H - 1, lf = w_low >= 0, rt = w_low + 1 <= W - 1;
        co[0] = (t && lf) ? h_low * W + w_low : -1;
        co[1] = (t && rt) ? h_low * W + w_low + 1 : -1;
        co[2] = (bt && lf) ? (h_low + 1) * W + w_low : -1;
        co[3] = (bt && rt) ? (h_low + 1) * W + w_low + 1 : -1;
        cw[0] = hh * hw; cw[1] = hh * lw; cw[2] = lh * hw; cw[3] = lh * lw;
      }
    }
    const float* vb = value + (b * S) * wstride + (int64_t)m * 32 + lane32;
    float col = 0.f;
    const int half_base = threadIdx.x & 32;
    // the four taps of a sample are loaded unconditionally and together (an absent tap reads row 0 and enters with weight
    // 0): as "(o >= 0) ? vb[..] : 0" each load sat under a lane-dependent branch (the two halves of a wave own different
    // heads) and was followed by s_waitcnt vmcnt(0) -- sixteen serialised round trips per (token, head)
    for (int p = 0; p < P; ++p) {
      const int src = half_base + p;
      float v[4], w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o = __shfl(co[k], src);
        w[k] = o >= 0 ? __shfl(cw[k], src) : 0.f;
        v[k] = vb[(int64_t)max(o, 0) * wstride];
      }
      float tap = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) tap += w[k] * v[k];
      col += tap * __shfl(aw, src);
    }
    out[hq * 32 + lane32] = col;
  }
}


// ---------------------------------------------------------------------------------------------
// backward (training row f2): replaces the col2im kernel family of the reference
// (ms_deform_im2col_cuda.cuh:87-159 bilinear part, :301-920 the per-channel-count reduction variants).
// One lane group (32 or 64 lanes) per (batch, query, head); lanes stride over the D channels, accumulate their
// partial d/d(loc) and d/d(attn) per sample in registers and combine them with wave shuffles -- no shared
// memory, no block-serial reduction; grad_value is accumulated with native float / double atomic adds.
template <typename T, int kGroup>
__global__ __launch_bounds__(kBlock) void msda_bwd(const T* __restrict__ grad_out, const T* __restrict__ value,
                                                   const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                                                   const T* __restrict__ loc, const T* __restrict__ attn,
                                                   T* __restrict__ grad_value, T* __restrict__ grad_loc,
                                                   T* __restrict__ grad_attn, int64_t n_heads_total, int S, int M, int D,
                                                   int L, int Lq, int P) {
  constexpr int kGroups = kBlock / kGroup;
  const int lane = threadIdx.x % kGroup;
  const int64_t wstride = (int64_t)M * D;
  for (int64_t hq = (int64_t)blockIdx.x * kGroups + threadIdx.x / kGroup; hq < n_heads_total;
       hq += (int64_t)gridDim.x * kGroups) {
    const int m = (int)(hq % M);
    const int64_t b = hq / ((int64_t)M * Lq);
    int64_t wptr = hq * L * P;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const int64_t base = (b * S + lsi[l]) * wstride + (int64_t)m * D;
      for (int p = 0; p < P; ++p, ++wptr) {
        const T loc_w = loc[2 * wptr], loc_h = loc[2 * wptr + 1], aw = attn[wptr];
        const T h_im = loc_h * H - (T)0.5, w_im = loc_w * W - (T)0.5;
        T g_w = 0, g_h = 0, g_a = 0;
        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
          const int h_low = (int)floor(h_im), w_low = (int)floor(w_im);
          const int h_high = h_low + 1, w_high = w_low + 1;
          const T lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;
          const bool t = h_low >= 0, bt = h_high <= H - 1, lf = w_low >= 0, rt = w_high <= W - 1;
          const int64_t o1 = base + ((int64_t)h_low * W + w_low) * wstride, o2 = o1 + wstride;
          const int64_t o3 = o1 + (int64_t)W * wstride, o4 = o3 + wstride;
          for (int c = lane; c < D; c += kGroup) {
            const T top = grad_out[hq * D + c];
            const T tgv = top * aw;
            T gh = 0, gw = 0, val = 0;
            if (t && lf) {
              const T v = value[o1 + c];
              gh -= hw * v; gw -= hh * v; val += hh * hw * v;
              atomicAdd(grad_value + o1 + c, hh * hw * tgv);
            }
            if (t && rt) {
              const T v = value[o2 + c];
              gh -= lw * v; gw += hh * v; val += hh * lw * v;
              atomicAdd(grad_value + o2 + c, hh * lw * tgv);
            }
            if (bt && lf) {
              const T v = value[o3 + c];
              gh += hw * v; gw -= lh * v; val += lh * hw * v;
              atomicAdd(grad_value + o3 + c, lh * hw * tgv);
            }
            if (bt && rt) {
              const T v = value[o4 + c];
              gh += lw * v; gw += lh * v; val += lh * lw * v;
              atomicAdd(grad_value + o4 + c, lh * lw * tgv);
            }
            g_a += top * val;
            g_w += (T)W * gw * tgv;
            g_h += (T)H * gh * tgv;
          }
        }
#pragma unroll
        for (int off = kGroup / 2; off > 0; off >>= 1) {
          g_w += __shfl_down(g_w, off, kGroup);
          g_h += __shfl_down(g_h, off, kGroup);
          g_a += __shfl_down(g_a, off, kGroup);
        }
        if (lane == 0) {
          grad_loc[2 * wptr] = g_w;
          grad_loc[2 * wptr + 1] = g_h;
          grad_attn[wptr] = g_a;
        }
      }
    }
  }
}

}  // namespace smos

using namespace smos;

extern "C" int smos_msda_fwd(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                             const void* sampling_loc, const void* attn_weight, void* out, int64_t N, int64_t S,
                             int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P, int32_t dtype,
                             smos_stream_t stream) {
  SMOS_REQUIRE(N >= 0 && S >= 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_fwd: bad sizes");
  if (dtype != SMOS_F32 && dtype != SMOS_F64) {
    set_error("msda_fwd: dtype code %d not implemented (the reference dispatches float/double only)", (int)dtype);
    return SMOS_ERR_UNSUPPORTED;
  }
  const int64_t total = N * Lq * M * D;
  if (total == 0) return SMOS_OK;
  SMOS_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
               "msda_fwd: null device pointer");
  SMOS_REQUIRE(S * M * D < (1LL << 31) && total < (1LL << 40), "msda_fwd: tensor too large for 32-bit tap offsets");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SMOS_F32) {
    if (D == 32 && L * P <= 8) {
      const int64_t heads = N * Lq * M;
      hipLaunchKernelGGL(msda_fwd_d32, dim3(grid_for(heads * 32, kBlock, 256 * 16)), dim3(kBlock), 0, s,
                         (const float*)value, spatial_shapes, level_start_index, (const float*)sampling_loc,
                         (const float*)attn_weight, (float*)out, heads, (int)S, (int)M, (int)L, (int)Lq, (int)P);
    } else {
      hipLaunchKernelGGL(msda_fwd_heads<float>, dim3(grid_for(N * Lq * M * 64, kBlock, 256 * 16)), dim3(kBlock), 0, s,
                         (const float*)value, spatial_shapes, level_start_index, (const float*)sampling_loc,
                         (const float*)attn_weight, (float*)out, N * Lq * M, (int)S, (int)M, (int)D, (int)L, (int)Lq, (int)P);
    }
  } else {
    hipLaunchKernelGGL(msda_fwd_heads<double>, dim3(grid_for(N * Lq * M * 64, kBlock, 256 * 16)), dim3(kBlock), 0, s,
                       (const double*)value, spatial_shapes, level_start_index, (const double*)sampling_loc,
                       (const double*)attn_weight, (double*)out, N * Lq * M, (int)S, (int)M, (int)D, (int)L, (int)Lq, (int)P);
  }
  return check_launch("msda_fwd");
}

extern "C" int smos_msda_fwd_qp(const float* value, const float* qp, float* out, int64_t N, int64_t H, int64_t W, int64_t M,
                                int64_t D, int64_t P, smos_stream_t stream) {
  SMOS_REQUIRE(N >= 0 && H > 0 && W > 0 && M > 0 && P > 0, "msda_fwd_qp: bad sizes");
  if (D != 32 || P > 8) {
    set_error("msda_fwd_qp: built for head width 32 and at most 8 points (got %lld, %lld)", (long long)D, (long long)P);
    return SMOS_ERR_UNSUPPORTED;
  }
  const int64_t heads = N * H * W * M;
  if (heads == 0) return SMOS_OK;
  SMOS_REQUIRE(value && qp && out && H * W * M * D < (1LL << 31), "msda_fwd_qp: null pointer / map too large");
  hipLaunchKernelGGL(msda_fwd_qp_d32, dim3(grid_for(heads * 32, kBlock, 256 * 16)), dim3(kBlock), 0, (hipStream_t)stream, value, qp, out,
                     heads, (int)M, (int)H, (int)W, (int)P);
  return check_launch("msda_fwd_qp");
}

extern "C" int smos_msda_bwd(const void* grad_out, const void* value, const int64_t* spatial_shapes,
                             const int64_t* level_start_index, const void* sampling_loc, const void* attn_weight,
                             void* grad_value, void* grad_sampling_loc, void* grad_attn_weight, int64_t N, int64_t S, int64_t M,
                             int64_t D, int64_t L, int64_t Lq, int64_t P, int32_t dtype, smos_stream_t stream) {
  SMOS_REQUIRE(N >= 0 && S >= 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_bwd: bad sizes");
  if (dtype != SMOS_F32 && dtype != SMOS_F64) {
    set_error("msda_bwd: dtype code %d not implemented (float/double only, like the reference)", (int)dtype);
    return SMOS_ERR_UNSUPPORTED;
  }
  const int64_t heads = N * Lq * M;
  if (heads == 0) return SMOS_OK;
  SMOS_REQUIRE(grad_out && value && spatial_shapes && level_start_index && sampling_loc && attn_weight && grad_value &&
                   grad_sampling_loc && grad_attn_weight, "msda_bwd: null device pointer");
  hipStream_t s = (hipStream_t)stream;
#define SMOS_BWD(T, G)                                                                                                  \
  hipLaunchKernelGGL((msda_bwd<T, G>), dim3(grid_for(heads * G, kBlock, 256 * 16)), dim3(kBlock), 0, s, (const T*)grad_out,   \
                     (const T*)value, spatial_shapes, level_start_index, (const T*)sampling_loc, (const T*)attn_weight,  \
                     (T*)grad_value, (T*)grad_sampling_loc, (T*)grad_attn_weight, heads, (int)S, (int)M, (int)D, (int)L,  \
                     (int)Lq, (int)P)
  if (dtype == SMOS_F32) {
    if (D <= 32) SMOS_BWD(float, 32); else SMOS_BWD(float, 64);
  } else {
    if (D <= 32) SMOS_BWD(double, 32); else SMOS_BWD(double, 64);
  }
#undef SMOS_BWD
  return check_launch("msda_bwd");
}
