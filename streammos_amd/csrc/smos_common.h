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

struct Dims4 {
  int64_t v[4];
};
struct Scale4 {
  float v[4];
};

}  // namespace smos
