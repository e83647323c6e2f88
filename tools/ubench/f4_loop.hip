// Inner-loop model of a Winograd F(4x4, 3x3) kernel on gfx950 (what would the matrix pipe see?): per k-step (4 input channels)
// a wave reads a 6x6 patch per lane from an LDS region image (pixel pitch 17 words, as csrc/conv_wino.hip), runs B^T d B
// (144 VALU operations), reads the 36 A operands (9 ds_read_b128) from a weight slot and issues 36 v_mfma_f32_16x16x4_f32;
// one barrier per k-step; two blocks of four waves per CU.  No staging, no epilogue: an upper bound for the real kernel.
//   hipcc -O3 --offload-arch=gfx950 -o f4_loop tools/ubench/f4_loop.hip && ./f4_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FENCE()                          \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

constexpr int kPP = 17, kRegW = 66, kRegH = 6;
constexpr int kInWords = kRegW * kRegH * kPP;

__device__ __forceinline__ void t6(float& d0, float& d1, float& d2, float& d3, float& d4, float& d5) {
  const float a = __builtin_fmaf(-4.f, d2, d4), b = __builtin_fmaf(-4.f, d1, d3), c = d4 - d2, e = d3 - d1;
  const float t0 = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4)), t5 = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
  d0 = t0; d1 = a + b; d2 = a - b; d3 = __builtin_fmaf(2.f, e, c); d4 = __builtin_fmaf(-2.f, e, c); d5 = t5;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void f4_loop(const float* __restrict__ src, float* __restrict__ out, int ksteps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4* w_lds = reinterpret_cast<float4*>(lds + kInWords);       // 2 slots x 9 x 64 float4
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = lane >> 4, tx = lane & 15;
  for (int i = tid; i < kInWords + 2 * 9 * 64 * 4; i += 256) lds[i] = src[i % 4096];
  __syncthreads();
  f32x4 acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float va[36], vb[36];
  const int in_base = (4 * tx) * kPP + q;
#define D_READ(v, i)                                                                         \
  do {                                                                                       \
    if (MODE & 1) break;                                                                     \
    const float* p_ = lds + in_base + 4 * (i);                                               \
    _Pragma("unroll") for (int r_ = 0; r_ < 6; ++r_)                                         \
        _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) v[6 * r_ + c_] = p_[(r_ * kRegW + c_) * kPP]; \
  } while (0)
#define T_COLS(v)                                                                            \
  do {                                                                                       \
    if (MODE & 2) break;                                                                     \
    _Pragma("unroll") for (int c_ = 0; c_ < 6; ++c_) t6(v[c_], v[6 + c_], v[12 + c_], v[18 + c_], v[24 + c_], v[30 + c_]); \
  } while (0)
#define T_ROWS(v, r0, r1)                                                                    \
  do {                                                                                       \
    if (MODE & 2) break;                                                                     \
    _Pragma("unroll") for (int r_ = r0; r_ < r1; ++r_) t6(v[6 * r_], v[6 * r_ + 1], v[6 * r_ + 2], v[6 * r_ + 3], v[6 * r_ + 4], v[6 * r_ + 5]); \
  } while (0)
#define MFMA6(v, g, a0, a1)                                                                                   \
  do {                                                                                                         \
    acc[6 * (g) + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, v[6 * (g) + 0], acc[6 * (g) + 0], 0, 0, 0);  \
    acc[6 * (g) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, v[6 * (g) + 1], acc[6 * (g) + 1], 0, 0, 0);  \
    acc[6 * (g) + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, v[6 * (g) + 2], acc[6 * (g) + 2], 0, 0, 0);  \
    acc[6 * (g) + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, v[6 * (g) + 3], acc[6 * (g) + 3], 0, 0, 0);  \
    acc[6 * (g) + 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, v[6 * (g) + 4], acc[6 * (g) + 4], 0, 0, 0);  \
    acc[6 * (g) + 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, v[6 * (g) + 5], acc[6 * (g) + 5], 0, 0, 0);  \
  } while (0)
  // A operands of row group g: 6 values = one b128 + one b64 (here: b128 + b128 of a 9-float4 slot, same LDS cost class)
#define A_READ(a0, a1, so, g)                      \
  do {                                             \
    a0 = w_lds[(so) + ((g) * 3 / 2) * 64 + lane];  \
    a1 = w_lds[(so) + (((g) * 3 + 1) / 2) * 64 + lane]; \
  } while (0)
#define KSTEP(v, vn, i)                                          \
  do {                                                           \
    float4 a0, a1, b0, b1;                                       \
    A_READ(a0, a1, so, 0);                                       \
    D_READ(vn, i);                                               \
    FENCE();                                                     \
    MFMA6(v, 0, a0, a1);                                         \
    FENCE();                                                     \
    A_READ(b0, b1, so, 1);                                       \
    T_COLS(vn);                                                  \
    FENCE();                                                     \
    MFMA6(v, 1, b0, b1);                                         \
    FENCE();                                                     \
    A_READ(a0, a1, so, 2);                                       \
    T_ROWS(vn, 0, 2);                                            \
    FENCE();                                                     \
    MFMA6(v, 2, a0, a1);                                         \
    FENCE();                                                     \
    A_READ(b0, b1, so, 3);                                       \
    T_ROWS(vn, 2, 4);                                            \
    FENCE();                                                     \
    MFMA6(v, 3, b0, b1);                                         \
    FENCE();                                                     \
    A_READ(a0, a1, so, 4);                                       \
    T_ROWS(vn, 4, 6);                                            \
    FENCE();                                                     \
    MFMA6(v, 4, a0, a1);                                         \
    FENCE();                                                     \
    A_READ(b0, b1, so, 5);                                       \
    FENCE();                                                     \
    MFMA6(v, 5, b0, b1);                                         \
    FENCE();                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
    if (!(MODE & 4)) __builtin_amdgcn_s_barrier();               \
    so = so == 0 ? 9 * 64 : 0;                                   \
  } while (0)
  int so = 0;
#pragma unroll
  for (int k = 0; k < 36; ++k) va[k] = vb[k] = (float)(lane + k);
  D_READ(va, 0);
  T_COLS(va);
  T_ROWS(va, 0, 6);
#pragma unroll 1
  for (int g = 0; g < ksteps; g += 4) {
    KSTEP(va, vb, 1);
    KSTEP(vb, va, 2);
    KSTEP(va, vb, 3);
    KSTEP(vb, va, 0);
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 36; ++k) s += acc[k];
  out[(size_t)blockIdx.x * 256 + tid] = s.x + s.y + s.z + s.w;
}

template <int MODE>
static void run(const char* name, const float* src, float* out, int ksteps) {
  const size_t ldsb = (size_t)(kInWords + 2 * 9 * 64 * 4) * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&f4_loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(f4_loop<MODE>, dim3(512), dim3(256), ldsb, 0, src, out, ksteps);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(f4_loop<MODE>, dim3(512), dim3(256), ldsb, 0, src, out, ksteps);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  // per SIMD: 2 waves x ksteps x 36 MFMAs x 32 cycles (8 passes) at 2.4 GHz
  const double floor_ms = 2.0 * ksteps * 36 * 32 / 2.4e6;
  printf("%-44s %.4f ms  MFMA floor %.4f ms  utilisation %.3f  (%d cycles per k-step and wave pair)\n", name, ms, floor_ms, floor_ms / ms,
         (int)(ms * 2.4e6 / ksteps));
}

int main() {
  float *src, *out;
  hipMalloc(&src, 4096 * 4);
  hipMalloc(&out, 512 * 256 * 4);
  hipMemset(src, 0, 4096 * 4);
  const int ksteps = 512;
  run<0>("full inner loop", src, out, ksteps);
  run<1>("no patch reads", src, out, ksteps);
  run<2>("no transform", src, out, ksteps);
  run<3>("no patch reads, no transform", src, out, ksteps);
  run<4>("no barrier", src, out, ksteps);
  run<7>("MFMA + A reads only", src, out, ksteps);
  return 0;
}
