// Voxel voting and the TTA reduce for gfx950.
//
// Replaces the per-frame body of voxel_voting.py:214-242: Crop (utils/transforms.py:151-161),
// Quantize (voxel_voting.py:77-91), determine_voxel_labels (:55-75) and
// get_point_labels_from_voxel_labels (:38-53), plus the pose alignment of the history frames
// (datasets/utils.py:116-126) that the reference does on the CPU after re-reading 16 files per frame.
//
// The reference builds a dense int64 [512*512*30, C] histogram (>= 189 MB zero-fill + 189 MB argmax
// read + 63 MB label write per frame).  Here a voxel is ONE packed 64-bit word holding three 21-bit
// class counters (a local map has < 2^21 points: 9 frames x 160 000), one native 64-bit atomic add per
// point, and the argmax is evaluated only at the voxels the current frame's points fall in.
// Class count: the reference sizes the histogram by max(label)+1; classes that never occur have zero
// votes and cannot win an argmax whose ties go to the lowest index, so three fixed counters give the
// identical label for labels in {0,1,2}.
#include "smos_common.h"

namespace smos {

constexpr int kNX = SMOS_VOTE_NX, kNY = SMOS_VOTE_NY, kNZ = SMOS_VOTE_NZ;
constexpr float kLoX = -50.0f, kLoY = -50.0f, kLoZ = -4.0f;  // voxel_voting.py:138,230-232

struct Pose {
  double m[12];  // top three rows of the 4x4
  int identity;
};

struct Quant {
  // bounds of the open crop interval, rounded to float32 the way torch compares a float32 tensor with
  // a Python scalar (utils/transforms.py:155-157 with eps = 1e-4), and the cell sizes as float32
  float clo[3], chi[3], cell[3], rcell[3];
  int recip;
};

__device__ __forceinline__ int64_t voxel_of(float x, float y, float z, const Quant& q) {
  const bool keep = (x > q.clo[0]) && (x < q.chi[0]) && (y > q.clo[1]) && (y < q.chi[1]) && (z > q.clo[2]) &&
                    (z < q.chi[2]);
  if (!keep) return -1;
  float qx, qy, qz;
  if (q.recip) {
    qx = __fmul_rn(__fsub_rn(x, kLoX), q.rcell[0]);
    qy = __fmul_rn(__fsub_rn(y, kLoY), q.rcell[1]);
    qz = __fmul_rn(__fsub_rn(z, kLoZ), q.rcell[2]);
  } else {
    qx = __fdiv_rn(__fsub_rn(x, kLoX), q.cell[0]);
    qy = __fdiv_rn(__fsub_rn(y, kLoY), q.cell[1]);
    qz = __fdiv_rn(__fsub_rn(z, kLoZ), q.cell[2]);
  }
  // .to(int64) truncates; inside the crop every coordinate is in range, the clamp only guards the table
  const int ix = min(max((int)qx, 0), kNX - 1), iy = min(max((int)qy, 0), kNY - 1), iz = min(max((int)qz, 0), kNZ - 1);
  return ((int64_t)ix * kNY + iy) * kNZ + iz;
}

__device__ __forceinline__ void load_xyz(const float* __restrict__ row, const Pose& p, float& x, float& y, float& z) {
  x = row[0];
  y = row[1];
  z = row[2];
  if (!p.identity) {
    // float32 -> float64 matmul with w = 1 -> float32, as datasets/utils.py:116-126; the sum order is the
    // dot-product order of a row of the 4x4 with (x, y, z, 1)
    const double dx = x, dy = y, dz = z;
    x = (float)pose_row_f64(p.m + 0, dx, dy, dz);
    y = (float)pose_row_f64(p.m + 4, dx, dy, dz);
    z = (float)pose_row_f64(p.m + 8, dx, dy, dz);
  }
}

__global__ __launch_bounds__(kBlock) void vote_accumulate(const float* __restrict__ pts, int64_t n, int64_t stride,
                                                          const uint8_t* __restrict__ labels, Pose pose, Quant q,
                                                          unsigned long long* __restrict__ table) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float x, y, z;
    load_xyz(pts + i * stride, pose, x, y, z);
    const int64_t v = voxel_of(x, y, z, q);
    if (v < 0) continue;
    const unsigned lab = labels[i];
    if (lab > 2) continue;
    atomicAdd(table + v, 1ULL << (21 * lab));
  }
}

// The whole voting window in one launch: blockIdx.y = frame.  Nine launches of ~7 us each (launch-bound at 120 k points)
// become one; the table update is a commutative integer add, so the result does not depend on how frames are grouped.
constexpr int kMaxFrames = 12;
struct FrameSet {
  const float* pts[kMaxFrames];
  const uint8_t* labels[kMaxFrames];
  int64_t n[kMaxFrames];
  int64_t stride[kMaxFrames];
  Pose pose[kMaxFrames];
};

__global__ __launch_bounds__(kBlock) void vote_accumulate_frames(FrameSet fs, Quant q, unsigned long long* __restrict__ table) {
  const int f = blockIdx.y;
  const float* __restrict__ pts = fs.pts[f];
  const uint8_t* __restrict__ labels = fs.labels[f];
  const int64_t n = fs.n[f], stride = fs.stride[f];
  const Pose& pose = fs.pose[f];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float x, y, z;
    load_xyz(pts + i * stride, pose, x, y, z);
    const int64_t v = voxel_of(x, y, z, q);
    if (v < 0) continue;
    const unsigned lab = labels[i];
    if (lab > 2) continue;
    atomicAdd(table + v, 1ULL << (21 * lab));
  }
}

__global__ __launch_bounds__(kBlock) void vote_resolve(const float* __restrict__ pts, int64_t n, int64_t stride,
                                                       const uint8_t* __restrict__ labels, Quant q,
                                                       const unsigned long long* __restrict__ table,
                                                       const int32_t* __restrict__ lut, int32_t* __restrict__ out) {
  Pose ident;
  ident.identity = 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float x, y, z;
    load_xyz(pts + i * stride, ident, x, y, z);
    const int64_t v = voxel_of(x, y, z, q);
    int lab = labels[i];
    if (v >= 0) {
      const unsigned long long w = table[v];
      const unsigned c0 = (unsigned)(w & 0x1FFFFF), c1 = (unsigned)((w >> 21) & 0x1FFFFF), c2 = (unsigned)((w >> 42) & 0x1FFFFF);
      lab = 0;                       // argmax, ties -> lowest class (torch.argmax returns the first maximum)
      unsigned best = c0;
      if (c1 > best) { best = c1; lab = 1; }
      if (c2 > best) { lab = 2; }
    }
    out[i] = lut ? lut[lab & 0xFF] : lab;
  }
}

// softmax over K classes, mean over B variants, argmax -- val_StreamMOS.py:97-98,113
__global__ __launch_bounds__(kBlock) void tta_argmax(const float* __restrict__ pred, int B, int K, int64_t N,
                                                     uint8_t* __restrict__ labels, float* __restrict__ prob) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int b = 0; b < B; ++b) {
      float v[8];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < K) {
          v[k] = pred[((int64_t)b * K + k) * N + n];
          mx = fmaxf(mx, v[k]);
        }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < K) {
          v[k] = expf(v[k] - mx);
          sum += v[k];
        }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < K) acc[k] += v[k] / sum;
    }
    int best = 0;
    float bv = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < K) {
        const float m = acc[k] / (float)B;
        if (prob) prob[n * K + k] = m;
        if (m > bv) {
          bv = m;
          best = k;
        }
      }
    labels[n] = (uint8_t)best;
  }
}

static Quant make_quant(int recip) {
  Quant q;
  const double lo[3] = {-50.0, -50.0, -4.0}, hi[3] = {50.0, 50.0, 2.0};
  const int size[3] = {kNX, kNY, kNZ};
  for (int d = 0; d < 3; ++d) {
    q.clo[d] = (float)(lo[d] + 1e-4);
    q.chi[d] = (float)(hi[d] - 1e-4);
    q.cell[d] = (float)((hi[d] - lo[d]) / size[d]);
    q.rcell[d] = 1.0f / q.cell[d];
  }
  q.recip = recip;
  return q;
}

}  // namespace smos

using namespace smos;

extern "C" int smos_vote_clear(uint64_t* table, smos_stream_t stream) {
  SMOS_REQUIRE(table, "vote_clear: null table");
  if (hipMemsetAsync(table, 0, (size_t)SMOS_VOTE_CELLS * sizeof(uint64_t), (hipStream_t)stream) != hipSuccess)
    return check_launch("vote_clear");
  return SMOS_OK;
}

extern "C" int smos_vote_accumulate(const float* pts, int64_t n, int64_t pt_stride, const uint8_t* labels,
                                    const double* pose_diff, int32_t recip_quantize, uint64_t* table,
                                    smos_stream_t stream) {
  SMOS_REQUIRE(n >= 0 && pt_stride >= 3, "vote_accumulate: bad sizes (n=%lld stride=%lld)", (long long)n, (long long)pt_stride);
  SMOS_REQUIRE(n < (1LL << 21), "vote_accumulate: %lld points overflow a 21-bit vote counter", (long long)n);
  if (n == 0) return SMOS_OK;
  SMOS_REQUIRE(pts && labels && table, "vote_accumulate: null device pointer");
  Pose p;
  p.identity = pose_diff ? 0 : 1;
  for (int i = 0; i < 12; ++i) p.m[i] = pose_diff ? pose_diff[i] : 0.0;
  hipLaunchKernelGGL(vote_accumulate, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, pts, n, pt_stride, labels,
                     p, make_quant(recip_quantize), (unsigned long long*)table);
  return check_launch("vote_accumulate");
}

extern "C" int smos_vote_accumulate_frames(int32_t count, const float* const* pts, const int64_t* n, const int64_t* pt_stride,
                                           const uint8_t* const* labels, const double* const* pose_diff,
                                           int32_t recip_quantize, uint64_t* table, smos_stream_t stream) {
  SMOS_REQUIRE(count >= 0 && (count == 0 || (pts && n && pt_stride && labels && pose_diff)), "vote_accumulate_frames: null array");
  int64_t total = 0;
  for (int f = 0; f < count; ++f) {
    SMOS_REQUIRE(n[f] >= 0 && pt_stride[f] >= 3, "vote_accumulate_frames: bad sizes in frame %d (n=%lld stride=%lld)", f,
                 (long long)n[f], (long long)pt_stride[f]);
    SMOS_REQUIRE(n[f] == 0 || (pts[f] && labels[f]), "vote_accumulate_frames: null device pointer in frame %d", f);
    total += n[f];
  }
  SMOS_REQUIRE(total < (1LL << 21), "vote_accumulate_frames: %lld points overflow a 21-bit vote counter", (long long)total);
  SMOS_REQUIRE(total == 0 || table, "vote_accumulate_frames: null table");
  for (int f = 0; f < count;) {
    FrameSet fs;
    int m = 0;
    int64_t longest = 0;
    for (; f < count && m < kMaxFrames; ++f) {
      if (n[f] == 0) continue;
      fs.pts[m] = pts[f];
      fs.labels[m] = labels[f];
      fs.n[m] = n[f];
      fs.stride[m] = pt_stride[f];
      fs.pose[m].identity = pose_diff[f] ? 0 : 1;
      for (int i = 0; i < 12; ++i) fs.pose[m].m[i] = pose_diff[f] ? pose_diff[f][i] : 0.0;
      longest = n[f] > longest ? n[f] : longest;
      ++m;
    }
    if (m == 0) continue;
    hipLaunchKernelGGL(vote_accumulate_frames, dim3(grid_for(longest), m), dim3(kBlock), 0, (hipStream_t)stream, fs,
                       make_quant(recip_quantize), (unsigned long long*)table);
    if (int rc = check_launch("vote_accumulate_frames")) return rc;
  }
  return SMOS_OK;
}

extern "C" int smos_vote_resolve(const float* pts, int64_t n, int64_t pt_stride, const uint8_t* labels,
                                 int32_t recip_quantize, const uint64_t* table, const int32_t* lut,
                                 int32_t* out_labels, smos_stream_t stream) {
  SMOS_REQUIRE(n >= 0 && pt_stride >= 3, "vote_resolve: bad sizes");
  if (n == 0) return SMOS_OK;
  SMOS_REQUIRE(pts && labels && table && out_labels, "vote_resolve: null device pointer");
  hipLaunchKernelGGL(vote_resolve, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, pts, n, pt_stride, labels,
                     make_quant(recip_quantize), (const unsigned long long*)table, lut, out_labels);
  return check_launch("vote_resolve");
}

extern "C" int smos_tta_argmax(const float* pred, int64_t B, int64_t K, int64_t N, uint8_t* labels, float* prob,
                               smos_stream_t stream) {
  SMOS_REQUIRE(B > 0 && K > 0 && K <= 8 && N >= 0, "tta_argmax: bad sizes (B=%lld K=%lld)", (long long)B, (long long)K);
  if (N == 0) return SMOS_OK;
  SMOS_REQUIRE(pred && labels, "tta_argmax: null device pointer");
  hipLaunchKernelGGL(tta_argmax, dim3(grid_for(N)), dim3(kBlock), 0, (hipStream_t)stream, pred, (int)B, (int)K, N, labels, prob);
  return check_launch("tta_argmax");
}
