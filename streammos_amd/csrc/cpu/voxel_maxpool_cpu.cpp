// Host twin of the point -> grid max-pool (see include/smos_cpu.h).
// Semantics follow deep_point/src/point_deep.cpp:19-88,98-132: an occupied cell ends with the maximum of
// its members (a member is stored first so that negative maxima survive the zero-filled output).
// Unlike the reference loop the cell of a point is computed once, not once per channel.
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../../include/smos_cpu.h"

namespace {

struct Geom {
  int D;
  const int64_t* size;
  const int64_t* ostride;  // 2 + D entries
  const float* scale;
};

template <typename T>
inline int64_t cell_of(const T* row, const Geom& g) {
  int64_t off = 0;
  for (int d = 0; d < g.D; ++d) {
    const float p = static_cast<float>(row[d]) * g.scale[d];
    if (!(p > -1.0f && p < static_cast<float>(g.size[d]))) return -1;
    off += static_cast<int64_t>(p) * g.ostride[2 + d];
  }
  return off;
}

template <typename T>
void forward(const T* feat, const int64_t* fs, const T* ind, T* out, int64_t* vidx, int64_t BS, int64_t C, int64_t N,
             const Geom& g) {
  std::vector<int64_t> cell(static_cast<size_t>(N));
  for (int64_t b = 0; b < BS; ++b) {
    for (int64_t n = 0; n < N; ++n) {
      const int64_t off = cell_of(ind + (b * N + n) * g.D, g);
      cell[n] = off < 0 ? -1 : off + b * g.ostride[0];
      if (vidx && off >= 0) vidx[b * N + n] = cell[n];
    }
    for (int64_t c = 0; c < C; ++c) {
      const T* f = feat + b * fs[0] + c * fs[1];
      T* o = out + c * g.ostride[1];
      for (int64_t n = 0; n < N; ++n)
        if (cell[n] >= 0) o[cell[n]] = f[n * fs[2]];
      for (int64_t n = 0; n < N; ++n)
        if (cell[n] >= 0 && o[cell[n]] < f[n * fs[2]]) o[cell[n]] = f[n * fs[2]];
    }
  }
}

template <typename T>
void backward(const T* feat, const int64_t* fs, const T* ind, const T* out, const T* gout, T* gfeat, int64_t BS,
              int64_t C, int64_t N, const Geom& g) {
  for (int64_t b = 0; b < BS; ++b)
    for (int64_t n = 0; n < N; ++n) {
      int64_t off = cell_of(ind + (b * N + n) * g.D, g);
      if (off < 0) continue;
      off += b * g.ostride[0];
      for (int64_t c = 0; c < C; ++c) {
        const int64_t fo = b * fs[0] + c * fs[1] + n * fs[2];
        const int64_t o = off + c * g.ostride[1];
        if (out[o] == feat[fo]) gfeat[fo] = gout[o];
      }
    }
}

inline bool bad(int64_t BS, int64_t C, int64_t N, int32_t D) { return BS < 0 || C < 0 || N < 0 || D < 1 || D > 4; }

}  // namespace

extern "C" int smos_cpu_voxel_maxpool_fwd(const void* feat, const int64_t* feat_stride, const void* ind, void* out,
                                          const int64_t* out_stride, int64_t* voxel_max_idx, int64_t BS, int64_t C,
                                          int64_t N, int32_t D, const int64_t* out_size, const float* scale,
                                          int32_t dtype) {
  if (bad(BS, C, N, D) || !feat_stride || !out_stride || !out_size || !scale) return 1;
  if (BS == 0 || C == 0 || N == 0) return 0;
  if (!feat || !ind || !out) return 1;
  const Geom g{D, out_size, out_stride, scale};
  if (dtype == 0)
    forward<float>((const float*)feat, feat_stride, (const float*)ind, (float*)out, voxel_max_idx, BS, C, N, g);
  else if (dtype == 2)
    forward<double>((const double*)feat, feat_stride, (const double*)ind, (double*)out, voxel_max_idx, BS, C, N, g);
  else
    return 2;
  return 0;
}

extern "C" int smos_cpu_voxel_maxpool_bwd(const void* feat, const int64_t* feat_stride, const void* ind,
                                          const void* out, const void* grad_out, const int64_t* out_stride,
                                          void* grad_feat, int64_t BS, int64_t C, int64_t N, int32_t D,
                                          const int64_t* out_size, const float* scale, int32_t dtype) {
  if (bad(BS, C, N, D) || !feat_stride || !out_stride || !out_size || !scale) return 1;
  if (BS == 0 || C == 0 || N == 0) return 0;
  if (!feat || !ind || !out || !grad_out || !grad_feat) return 1;
  const Geom g{D, out_size, out_stride, scale};
  if (dtype == 0)
    backward<float>((const float*)feat, feat_stride, (const float*)ind, (const float*)out, (const float*)grad_out,
                    (float*)grad_feat, BS, C, N, g);
  else if (dtype == 2)
    backward<double>((const double*)feat, feat_stride, (const double*)ind, (const double*)out,
                     (const double*)grad_out, (double*)grad_feat, BS, C, N, g);
  else
    return 2;
  return 0;
}
