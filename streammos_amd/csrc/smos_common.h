// Shared helpers for the gfx950 kernels of libsmos_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/smos.h"

namespace smos {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves, one per SIMD
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 resident blocks: grid-stride beyond that

// Largest element count a 32-bit grid-stride loop (`for (int i = ...; i < total; i += gridDim.x * blockDim.x)`) may be
// given: the counter of the last iteration is up to one grid stride past `total` and must still be a positive int
// (grids are capped at 256 * 32 blocks of kBlock threads = 2^21 elements per stride).
constexpr int64_t kMaxTotal32 = (1LL << 31) - (1LL << 22);

void set_error(const char* fmt, ...);

inline int grid_for(int64_t work_items, int block = kBlock, int64_t cap = kMaxGrid) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SMOS_ERR_LAUNCH;
  }
  return SMOS_OK;
}

#define SMOS_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      smos::set_error(__VA_ARGS__);      \
      return SMOS_ERR_ARG;               \
    }                                    \
  } while (0)

// Launch facts of one kernel on ONE device: CU count, the dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize
// is a per-device attribute) and, if asked for, the resident blocks per CU.  Looked up per (kernel, current device) under
// a mutex, so a process that drives several GPUs -- or several threads -- gets each device set up exactly once.
struct KernelSetup {
  int cus = 0;
  int per_cu = 0;   // hipOccupancyMaxActiveBlocksPerMultiprocessor (0 when not requested)
};
int kernel_setup(const void* fn, size_t dyn_lds_bytes, int occupancy_block, KernelSetup* out, const char* what);

// smos_debug_set_conv_grid_cap: upper bound on the grid of the persistent convolution kernels (0 = none).  A test hook: it makes
// a block walk several work items on shapes small enough to check against float64, on any CU count.
int64_t conv_grid_cap(int64_t cap);

// One row of the 4x4 pose difference times (x, y, z, 1) in float64, in the operation order of the dgemm micro-kernel
// numpy's ``mat.dot`` runs for datasets/utils.py:116-126 (one accumulator per output element, k = 0..3, fused
// multiply-adds): written with the round-to-nearest intrinsics so that -ffp-contract has nothing to decide.  The result
// is rounded to float32 once by the caller, as the reference's ``pcds_out[..., :3] = pcds_tmp[..., :3]`` does.
__device__ __forceinline__ double pose_row_f64(const double* m, double x, double y, double z) {
  return __dadd_rn(__fma_rn(m[2], z, __fma_rn(m[1], y, __dmul_rn(m[0], x))), m[3]);
}

struct Dims4 {
  int64_t v[4];
};
struct Scale4 {
  float v[4];
};

}  // namespace smos
