// Dev: where does the dispatcher put the blocks of a 2-blocks-per-CU grid?  Every block records the hardware id of the CU
// it runs on and stays resident for a while; the host prints the histogram of blocks per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void census(unsigned* out, long long spin) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  lds[threadIdx.x] = 1.0f;
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (lds[threadIdx.x] == 2.0f) out[0] = 0;
}

int main(int argc, char** argv) {
  int grid = argc > 1 ? atoi(argv[1]) : 512;
  int lds = argc > 2 ? atoi(argv[2]) : 12288;
  unsigned* d;
  hipMalloc(&d, grid * 8);
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(census, dim3(grid), dim3(256), lds, 0, d, 2000000LL);
  hipDeviceSynchronize();
  std::vector<unsigned> h(grid * 2);
  hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  for (int b = 0; b < grid; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
    unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
  }
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[kv.second]++;
  printf("grid %d lds %d: %zu distinct CUs used;", grid, lds, per_cu.size());
  for (auto& kv : hist) printf("  %d CUs with %d blocks", kv.second, kv.first);
  printf("\n");
  return 0;
}
