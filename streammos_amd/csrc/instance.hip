// Instance-level voting for gfx950: the two heavy parts of cluster()
// (voxel_instance_voting.py:144-193) as device kernels.
//
//   dbscan     : DBSCAN(eps, min_samples).fit_predict of the scan's foreground points (:150-153).  scikit-learn's
//                result is a pure function of the eps-neighbourhood graph: core = at least min_samples points within
//                eps (the point itself included); clusters = connected components of the core points; a border point
//                takes the cluster that is expanded first among those it touches; clusters are numbered in the order
//                of their lowest-index core point.  Here every cluster is NAMED by the index of its lowest core point
//                (so "expanded first" = smallest name): a neighbour count, min-label propagation with pointer jumping
//                until nothing changes, and one pass for the border points -- each a banded pair test over the
//                x-sorted points.  Distances are taken the
//                way scikit-learn's KD-tree takes them: float32 coordinates widened to float64, squared differences
//                summed x, y, z without contraction, compared with eps*eps by <=.
//   box_vote   : for every kept cluster, the number of local-map points of class 1 / 2 inside its axis-aligned box
//                (:170-187) -- the local map is never materialised: history frames are pose-aligned (float64 matmul,
//                float32 result, datasets/utils.py:116-126) and cropped (utils/transforms.py:151-161) on the fly, as
//                in vote.hip.  in_hull() of the reference triangulates the 8 box corners and asks find_simplex >= 0;
//                for float32 points and float32 corners that is the closed interval test used here.
#include "smos_common.h"
#include <hipcub/hipcub.hpp>

namespace smos {

constexpr int kTile = kBlock;   // points staged per LDS tile

__device__ __forceinline__ bool within(double xi, double yi, double zi, float qx, float qy, float qz, double eps2) {
  const double dx = __dsub_rn(xi, (double)qx), dy = __dsub_rn(yi, (double)qy), dz = __dsub_rn(zi, (double)qz);
  const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
  return d2 <= eps2;
}

// The points are processed in x-sorted order (hipCUB radix sort of the x coordinate): a block of 256 consecutive sorted
// points only has to look at the sorted range whose x lies within eps of the block's own x interval, which turns the
// all-pairs test into a narrow band.  Cluster names stay ORIGINAL indices (orig[] maps sorted position -> original index,
// pos[] back), so the result does not depend on the sort.
__global__ __launch_bounds__(kBlock) void dbscan_keys(const float* __restrict__ pts, int n, int64_t stride, float* __restrict__ keys,
                                                      int* __restrict__ iota) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    keys[i] = pts[(int64_t)i * stride];
    iota[i] = i;
  }
}

__global__ __launch_bounds__(kBlock) void dbscan_gather(const float* __restrict__ pts, int n, int64_t stride, const int* __restrict__ orig,
                                                        const float* __restrict__ keys_sorted, double eps, float4* __restrict__ sp,
                                                        int* __restrict__ pos, int2* __restrict__ band) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    const int o = orig[i];
    const float* p = pts + (int64_t)o * stride;
    sp[i] = make_float4(p[0], p[1], p[2], 0.0f);
    pos[o] = i;
  }
  if (threadIdx.x == 0) {
    // band of this block: sorted positions whose x is in [x_first - eps, x_last + eps] (float64 bounds: a superset of
    // the exact neighbourhood, the exact test happens per pair)
    const int first = blockIdx.x * kBlock, last = min(n, first + kBlock) - 1;
    const double lo = (double)keys_sorted[first] - eps, hi = (double)keys_sorted[last] + eps;
    int a = 0, b = first;                       // lowest position with key >= lo
    while (a < b) {
      const int m = (a + b) >> 1;
      if ((double)keys_sorted[m] < lo) a = m + 1; else b = m;
    }
    int c = last + 1, d = n;                    // lowest position with key > hi
    while (c < d) {
      const int m = (c + d) >> 1;
      if ((double)keys_sorted[m] <= hi) c = m + 1; else d = m;
    }
    band[blockIdx.x] = make_int2(a, c);
  }
}

// mode 0: label[i] = orig[i] if sorted point i is a core point, else -1
// mode 1: one propagation sweep over the core points (label[] in place; *changed set when a label dropped)
// mode 2: out[orig[i]] = cluster name (core: its label; border: smallest label among its core neighbours; else -1)
template <int kMode>
__global__ __launch_bounds__(kBlock) void dbscan_pass(const float4* __restrict__ sp, int n, const int2* __restrict__ band,
                                                      const int* __restrict__ orig, const int* __restrict__ pos, double eps2,
                                                      int min_samples, int* label, int* __restrict__ out, int* changed) {
  __shared__ float tx[kTile], ty[kTile], tz[kTile];
  __shared__ int tl[kTile];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < n;
  double xi = 0, yi = 0, zi = 0;
  int mine = -1;
  if (live) {
    const float4 p = sp[i];
    xi = p.x; yi = p.y; zi = p.z;
    if (kMode != 0) {
      mine = label[i];
      if (kMode == 1 && mine >= 0) mine = label[pos[mine]];   // pointer jumping: adopt the label of my representative
    }
  }
  int count = 0;
  int best = (kMode == 2 && mine < 0) ? 0x7fffffff : mine;
  const bool active = live && (kMode == 0 || (kMode == 1 && mine >= 0) || (kMode == 2 && mine < 0));
  const int2 range = band[blockIdx.x];
  for (int j0 = range.x; j0 < range.y; j0 += kTile) {
    const int j = j0 + threadIdx.x;
    __syncthreads();
    if (j < range.y) {
      const float4 q = sp[j];
      tx[threadIdx.x] = q.x; ty[threadIdx.x] = q.y; tz[threadIdx.x] = q.z;
      if (kMode != 0) tl[threadIdx.x] = label[j];
    }
    __syncthreads();
    if (!active) continue;
    const int m = min(kTile, range.y - j0);
    for (int t = 0; t < m; ++t) {
      if (kMode != 0 && tl[t] < 0) continue;   // only core points carry labels
      if (!within(xi, yi, zi, tx[t], ty[t], tz[t], eps2)) continue;
      if (kMode == 0) ++count;
      else best = min(best, tl[t]);
    }
  }
  if (!live) return;
  if (kMode == 0) {
    label[i] = count >= min_samples ? orig[i] : -1;
  } else if (kMode == 1) {
    if (mine >= 0 && best < label[i]) {
      label[i] = best;
      *changed = 1;
    }
  } else {
    out[orig[i]] = mine >= 0 ? mine : (best == 0x7fffffff ? -1 : best);
  }
}

struct BoxPose {
  double m[12];
  int identity;
};

// counts[k*3 + c] += 1 for every kept point of class c inside box k.  boxes: [K, 6] float32 (lo xyz, hi xyz).
__global__ __launch_bounds__(kBlock) void box_vote(const float* __restrict__ pts, int64_t n, int64_t stride,
                                                   const uint8_t* __restrict__ labels, BoxPose pose, float clo0, float clo1,
                                                   float clo2, float chi0, float chi1, float chi2,
                                                   const float* __restrict__ boxes, int K, unsigned* __restrict__ counts) {
  extern __shared__ float lds_box[];
  for (int t = threadIdx.x; t < K * 6; t += blockDim.x) lds_box[t] = boxes[t];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned lab = labels[i];
    if (lab == 0 || lab > 2) continue;          // only classes 1 and 2 are ever read back (:178-179)
    const float* row = pts + i * stride;
    float x = row[0], y = row[1], z = row[2];
    if (!pose.identity) {
      const double dx = x, dy = y, dz = z;
      x = (float)pose_row_f64(pose.m + 0, dx, dy, dz);
      y = (float)pose_row_f64(pose.m + 4, dx, dy, dz);
      z = (float)pose_row_f64(pose.m + 8, dx, dy, dz);
    }
    if (!((x > clo0) && (x < chi0) && (y > clo1) && (y < chi1) && (z > clo2) && (z < chi2))) continue;
    for (int k = 0; k < K; ++k) {
      const float* b = lds_box + k * 6;
      if (x >= b[0] && x <= b[3] && y >= b[1] && y <= b[4] && z >= b[2] && z <= b[5]) atomicAdd(counts + k * 3 + lab, 1u);
    }
  }
}

}  // namespace smos

using namespace smos;

namespace {
struct DbscanWork {
  size_t keys_in, keys_out, iota, orig, pos, sp, label, band, flag, cub, cub_bytes, total;
};

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

bool dbscan_layout(int64_t n, DbscanWork& w) {
  const size_t nn = (size_t)n, blocks = (nn + kBlock - 1) / kBlock;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += up256(bytes); return at; };
  w.keys_in = take(nn * 4); w.keys_out = take(nn * 4); w.iota = take(nn * 4); w.orig = take(nn * 4); w.pos = take(nn * 4);
  w.sp = take(nn * 16); w.label = take(nn * 4); w.band = take(blocks * 8); w.flag = take(4);
  w.cub_bytes = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, w.cub_bytes, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                         (int*)nullptr, (int)n) != hipSuccess)
    return false;
  w.cub = take(w.cub_bytes);
  w.total = off;
  return true;
}
}  // namespace

extern "C" int64_t smos_dbscan_work_bytes(int64_t n) {
  if (n <= 0 || n >= (1LL << 31)) return 0;
  DbscanWork w;
  return dbscan_layout(n, w) ? (int64_t)w.total : -1;
}

extern "C" int smos_dbscan(const float* pts, int64_t n, int64_t pt_stride, double eps, int32_t min_samples, int32_t* labels,
                           void* work, int64_t work_bytes, int32_t max_sweeps, smos_stream_t stream) {
  SMOS_REQUIRE(n >= 0 && n < (1LL << 31) && pt_stride >= 3 && eps > 0 && min_samples >= 1 && max_sweeps >= 1,
               "dbscan: bad arguments");
  if (n == 0) return SMOS_OK;
  SMOS_REQUIRE(pts && labels && work, "dbscan: null device pointer");
  DbscanWork w;
  SMOS_REQUIRE(dbscan_layout(n, w), "dbscan: sort workspace query failed");
  SMOS_REQUIRE(work_bytes >= (int64_t)w.total && (reinterpret_cast<uintptr_t>(work) & 255) == 0,
               "dbscan: workspace too small or not 256-byte aligned (%lld bytes needed)", (long long)w.total);
  hipStream_t s = (hipStream_t)stream;
  char* base = static_cast<char*>(work);
  float* keys_in = (float*)(base + w.keys_in);
  float* keys_out = (float*)(base + w.keys_out);
  int* iota = (int*)(base + w.iota);
  int* orig = (int*)(base + w.orig);
  int* pos = (int*)(base + w.pos);
  float4* sp = (float4*)(base + w.sp);
  int* core = (int*)(base + w.label);
  int2* band = (int2*)(base + w.band);
  int* flag = (int*)(base + w.flag);
  const double eps2 = eps * eps;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL(dbscan_keys, grid, block, 0, s, pts, (int)n, pt_stride, keys_in, iota);
  size_t cub_bytes = w.cub_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(base + w.cub, cub_bytes, (const float*)keys_in, keys_out, (const int*)iota, orig, (int)n, 0,
                                         32, s) != hipSuccess) {
    set_error("dbscan: radix sort failed");
    return SMOS_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(dbscan_gather, grid, block, 0, s, pts, (int)n, pt_stride, (const int*)orig, (const float*)keys_out, eps, sp, pos, band);
  hipLaunchKernelGGL(dbscan_pass<0>, grid, block, 0, s, (const float4*)sp, (int)n, (const int2*)band, (const int*)orig, (const int*)pos,
                     eps2, (int)min_samples, core, (int*)nullptr, (int*)nullptr);
  // propagation sweeps; the "changed" flag is read back every kBatch sweeps (small inputs are latency-bound on that sync)
  constexpr int kBatch = 4;
  int sweeps = 0;
  for (;;) {
    if (hipMemsetAsync(flag, 0, sizeof(int), s) != hipSuccess) break;
    for (int k = 0; k < kBatch; ++k)
      hipLaunchKernelGGL(dbscan_pass<1>, grid, block, 0, s, (const float4*)sp, (int)n, (const int2*)band, (const int*)orig,
                         (const int*)pos, eps2, (int)min_samples, core, (int*)nullptr, flag);
    int host_flag = 0;
    if (hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) break;
    if (hipStreamSynchronize(s) != hipSuccess) break;
    if (!host_flag) {
      hipLaunchKernelGGL(dbscan_pass<2>, grid, block, 0, s, (const float4*)sp, (int)n, (const int2*)band, (const int*)orig,
                         (const int*)pos, eps2, (int)min_samples, core, labels, (int*)nullptr);
      return check_launch("dbscan");
    }
    sweeps += kBatch;
    if (sweeps >= max_sweeps) {
      set_error("dbscan: labels did not settle within %d sweeps", (int)max_sweeps);
      return SMOS_ERR_LAUNCH;
    }
  }
  return check_launch("dbscan");
}

extern "C" int smos_box_vote(const float* pts, int64_t n, int64_t pt_stride, const uint8_t* labels, const double* pose_diff,
                             const float* boxes, int32_t K, uint32_t* counts, smos_stream_t stream) {
  SMOS_REQUIRE(n >= 0 && pt_stride >= 3 && K >= 0 && K <= 2048, "box_vote: bad arguments (at most 2048 boxes)");
  if (n == 0 || K == 0) return SMOS_OK;
  SMOS_REQUIRE(pts && labels && boxes && counts, "box_vote: null device pointer");
  BoxPose p;
  p.identity = pose_diff ? 0 : 1;
  for (int i = 0; i < 12; ++i) p.m[i] = pose_diff ? pose_diff[i] : 0.0;
  // open crop interval with eps = 1e-4, bounds rounded to float32 as torch does (utils/transforms.py:155-157)
  const float clo[3] = {(float)(-50.0 + 1e-4), (float)(-50.0 + 1e-4), (float)(-4.0 + 1e-4)};
  const float chi[3] = {(float)(50.0 - 1e-4), (float)(50.0 - 1e-4), (float)(2.0 - 1e-4)};
  hipLaunchKernelGGL(box_vote, dim3(grid_for(n)), dim3(kBlock), (size_t)K * 6 * sizeof(float), (hipStream_t)stream, pts, n,
                     pt_stride, labels, p, clo[0], clo[1], clo[2], chi[0], chi[1], chi[2], boxes, (int)K, counts);
  return check_launch("box_vote");
}
