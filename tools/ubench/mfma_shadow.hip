// What does an instruction cost when it sits between the MFMAs of a saturated matrix pipe (gfx950, v_mfma_f32_16x16x4_f32:
// 32 cycles each)?  A wave issues 32 independent MFMAs per round; behind each MFMA come N copies of one kind of instruction
// (independent of the MFMAs); s_memtime around 64 rounds.  One and two waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shadow.bin tools/ubench/mfma_shadow.hip && ./mfma_shadow.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP1(x) x
#define REP2(x) x x
#define REP4(x) x x x x

template <int KIND, int N>
__device__ __forceinline__ void filler(float& va, float& vb, int& sa, const float* lds, float4& l4, float& l1) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(va) : "v"(vb));
    if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sa));
    if (KIND == 3) asm volatile("s_nop 0");
    if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
    if (KIND == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(l1) : "v"((unsigned)(size_t)lds));
    if (KIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(l4) : "v"((unsigned)(size_t)lds));
    if (KIND == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&l4) : "v"(*(double*)&l4.z));
    if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va) : "v"(vb));
  }
}

template <int KIND, int N>
__global__ __launch_bounds__(256, 2) void shadow(float* out, unsigned long long* cyc, int rounds) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  f32x4 acc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = (float)threadIdx.x, b = 1.0f, va = 0.f, vb = 1.f, l1 = 0.f;
  float4 l4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int sa = 0;
  const float* lp = lds + (threadIdx.x & 63) * 4;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
      filler<KIND, N>(va, vb, sa, lp, l4, l1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 32; ++k) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w + va + (float)sa + l1 + l4.x + l4.z;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int N>
static double run(int blocks_per_cu, float* out, unsigned long long* cyc) {
  const int rounds = 64;
  hipLaunchKernelGGL((shadow<KIND, N>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, cyc, rounds);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((shadow<KIND, N>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, cyc, rounds);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  return (double)c / (rounds * 32);
}

template <int KIND>
static void row(const char* name, float* out, unsigned long long* cyc) {
  printf("%-26s", name);
  for (int bpc = 1; bpc <= 2; ++bpc)
    printf("  | %d wave/SIMD: +0 %5.1f  +1 %5.1f  +2 %5.1f  +4 %5.1f", bpc, run<0, 0>(bpc, out, cyc), run<KIND, 1>(bpc, out, cyc),
           run<KIND, 2>(bpc, out, cyc), run<KIND, 4>(bpc, out, cyc));
  printf("   (cycles per MFMA of the timed wave)\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 512 * 256 * 4);
  hipMalloc(&cyc, 8);
  row<1>("v_add_f32", out, cyc);
  row<8>("v_fma_f32", out, cyc);
  row<7>("v_pk_add_f32", out, cyc);
  row<2>("s_add_u32", out, cyc);
  row<3>("s_nop 0", out, cyc);
  row<4>("s_waitcnt lgkmcnt(0)", out, cyc);
  row<5>("ds_read_b32", out, cyc);
  row<6>("ds_read_b128", out, cyc);
  return 0;
}
