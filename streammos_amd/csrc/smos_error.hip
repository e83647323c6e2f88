#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "smos_common.h"

namespace smos {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int kernel_setup(const void* fn, size_t dyn_lds_bytes, int occupancy_block, KernelSetup* out, const char* what) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, KernelSetup> table;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("%s: hipGetDevice failed", what);
    return SMOS_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(mu);
  auto it = table.find({fn, dev});
  if (it == table.end()) {
    KernelSetup ks;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
      set_error("%s: device query failed", what);
      return SMOS_ERR_LAUNCH;
    }
    ks.cus = cus;
    if (dyn_lds_bytes > 0 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds_bytes) != hipSuccess) {
      set_error("%s: cannot opt in to %zu bytes of dynamic LDS", what, dyn_lds_bytes);
      return SMOS_ERR_LAUNCH;
    }
    if (occupancy_block > 0 &&
        (hipOccupancyMaxActiveBlocksPerMultiprocessor(&ks.per_cu, fn, occupancy_block, dyn_lds_bytes) != hipSuccess ||
         ks.per_cu < 1)) {
      set_error("%s: occupancy query failed", what);
      return SMOS_ERR_LAUNCH;
    }
    it = table.emplace(std::make_pair(fn, dev), ks).first;
  }
  *out = it->second;
  return SMOS_OK;
}
static std::atomic<int> g_conv_grid_cap{0};
int64_t conv_grid_cap(int64_t cap) {
  const int lim = g_conv_grid_cap.load(std::memory_order_relaxed);
  return lim > 0 && lim < cap ? lim : cap;
}
}  // namespace smos

extern "C" int smos_debug_set_conv_grid_cap(int32_t blocks) {
  smos::g_conv_grid_cap.store(blocks > 0 ? blocks : 0, std::memory_order_relaxed);
  return SMOS_OK;
}
extern "C" int smos_abi_version(void) { return SMOS_ABI_VERSION; }
extern "C" const char* smos_last_error(void) { return smos::g_err; }
