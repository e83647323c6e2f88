// The reference-side binding of INTEGRATION.md section 3 as code that is compiled and tested: the two pybind11 extension
// modules of the reference, same module-level function names and argument lists, with libsmos_hip.so's C ABI
// (include/smos.h) behind them instead of the reference's CUDA translation units.
//
//   -DSMOS_SHIM_POINT_DEEP  ->  point_deep.cuda_kernel            (deep_point/src/point_deep_cuda.cpp:21-62)
//   -DSMOS_SHIM_MSDA        ->  MultiScaleDeformableAttention     (deformattn/src/vision.cpp:13-16,
//                                                                  deformattn/src/ms_deform_attn.h:20-60)
//
// Host compiler only (g++): no device code here, torch supplies tensors and the current HIP stream.  Built by
// streammos_amd/build.py::build_pybind_shims into streammos_amd/lib/; the shipped Python path (ctypes, refapi/) does
// not depend on it.
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <vector>

#include "smos.h"

namespace {

// torch-ROCm presents its HIP devices under the device type "cuda": the generic c10::DeviceGuard accepts that, and the HIP
// stream pool is indexed by the device ordinal either way.
hipStream_t current_stream(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check_input(const at::Tensor& t, const char* name) {
  // CHECK_INPUT of point_deep_cuda.cpp:11-13
  TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

int dtype_code(const at::Tensor& t, const char* what) {
  if (t.scalar_type() == at::kFloat) return SMOS_F32;
  if (t.scalar_type() == at::kDouble) return SMOS_F64;
  if (t.scalar_type() == at::kHalf) return SMOS_F16;
  TORCH_CHECK(false, what, ": unsupported dtype ", t.scalar_type());
  return -1;
}

}  // namespace

#ifdef SMOS_SHIM_POINT_DEEP
namespace {

struct Geometry {
  int64_t bs, c, n;
  int32_t d;
  std::vector<int64_t> feat_stride, out_stride, out_size;
  std::vector<float> scale;
};

// The reference passes output size / strides / scale as small DEVICE tensors (deep_point/__init__.py:29-32); sizes and
// strides are properties of voxel_out itself, only the D scale factors have to come back to the host.
Geometry geometry(const at::Tensor& pcds_feat, const at::Tensor& pcds_ind, const at::Tensor& voxel_out,
                  const at::Tensor& scale_rate) {
  Geometry g;
  TORCH_CHECK(pcds_feat.dim() >= 3 && pcds_ind.dim() >= 3, "pcds_feat [BS,C,N,1] and pcds_ind [BS,N,D,1] expected");
  g.bs = pcds_feat.size(0);
  g.c = pcds_feat.size(1);
  g.n = pcds_feat.size(2);
  g.d = (int32_t)pcds_ind.size(2);
  TORCH_CHECK(voxel_out.dim() == 2 + g.d, "voxel_out must be [BS,C,*output_size]");
  TORCH_CHECK(pcds_ind.size(0) == g.bs && pcds_ind.size(1) == g.n, "pcds_ind does not match pcds_feat");
  TORCH_CHECK(voxel_out.size(0) == g.bs && voxel_out.size(1) == g.c, "voxel_out does not match pcds_feat");
  g.feat_stride = {pcds_feat.stride(0), pcds_feat.stride(1), pcds_feat.stride(2)};
  g.out_stride = voxel_out.strides().vec();
  g.out_size = voxel_out.sizes().slice(2).vec();
  TORCH_CHECK(scale_rate.numel() == g.d, "scale_rate must hold one factor per grid dimension");
  // The D scale factors have to come back to the host: one small synchronising copy per call.  It cannot be cached by
  // tensor identity -- the reference builds a FRESH device tensor per call (deep_point/__init__.py:32), so an address is
  // reused with other values; a CPU scale_rate tensor (what a maintainer would pass) costs no sync.
  at::Tensor sc = scale_rate.to(at::kCPU, at::kFloat).contiguous();
  g.scale.assign(sc.data_ptr<float>(), sc.data_ptr<float>() + g.d);
  return g;
}

void voxel_maxpooling_forward(at::Tensor pcds_feat, at::Tensor pcds_ind, at::Tensor voxel_out, at::Tensor voxel_max_idx,
                              at::Tensor voxel_out_size, at::Tensor voxel_out_stride, at::Tensor output_size,
                              at::Tensor scale_rate) {
  check_input(pcds_feat, "pcds_feat");
  check_input(pcds_ind, "pcds_ind");
  check_input(voxel_out, "voxel_out");
  check_input(voxel_max_idx, "voxel_max_idx");
  check_input(voxel_out_size, "voxel_out_size");
  check_input(voxel_out_stride, "voxel_out_stride");
  check_input(output_size, "output_size");
  TORCH_CHECK(scale_rate.is_contiguous(), "scale_rate must be contiguous");     // may live on the host: no sync then
  TORCH_CHECK(voxel_max_idx.scalar_type() == at::kLong, "voxel_max_idx must be int64");
  TORCH_CHECK(pcds_ind.scalar_type() == pcds_feat.scalar_type() && voxel_out.scalar_type() == pcds_feat.scalar_type(),
              "pcds_feat / pcds_ind / voxel_out dtypes differ");
  // one slot per POINT (deep_point/__init__.py:27: torch.full([BS, N], -1)); the kernel writes BS * N int64 words
  TORCH_CHECK(voxel_max_idx.numel() == pcds_ind.size(0) * pcds_ind.size(1), "voxel_max_idx must be [BS, N]");
  TORCH_CHECK(pcds_ind.device() == pcds_feat.device() && voxel_out.device() == pcds_feat.device() &&
              voxel_max_idx.device() == pcds_feat.device(), "all tensors must live on pcds_feat's device");
  Geometry g = geometry(pcds_feat, pcds_ind, voxel_out, scale_rate);
  c10::DeviceGuard guard(pcds_feat.device());
  at::Tensor flag = at::zeros({4}, pcds_feat.options().dtype(at::kInt));   // "saw a negative feature" scratch
  int rc = smos_voxel_maxpool_fwd(pcds_feat.data_ptr(), g.feat_stride.data(), pcds_ind.data_ptr(), voxel_out.data_ptr(),
                                  g.out_stride.data(), voxel_max_idx.data_ptr<int64_t>(), g.bs, g.c, g.n, g.d,
                                  g.out_size.data(), g.scale.data(), dtype_code(pcds_feat, "voxel_maxpooling_forward"),
                                  flag.data_ptr<int32_t>(), current_stream(pcds_feat));
  TORCH_CHECK(rc == SMOS_OK, "smos_voxel_maxpool_fwd: ", smos_last_error());
}

void voxel_maxpooling_backward(at::Tensor pcds_feat, at::Tensor pcds_ind, at::Tensor voxel_out, at::Tensor voxel_max_idx,
                               at::Tensor grad_pcds_feat, at::Tensor grad_voxel_out, at::Tensor voxel_out_size,
                               at::Tensor voxel_out_stride, at::Tensor output_size, at::Tensor scale_rate) {
  check_input(pcds_feat, "pcds_feat");
  check_input(pcds_ind, "pcds_ind");
  check_input(voxel_out, "voxel_out");
  check_input(voxel_max_idx, "voxel_max_idx");
  check_input(grad_pcds_feat, "grad_pcds_feat");
  check_input(grad_voxel_out, "grad_voxel_out");
  check_input(voxel_out_size, "voxel_out_size");
  check_input(voxel_out_stride, "voxel_out_stride");
  check_input(output_size, "output_size");
  TORCH_CHECK(scale_rate.is_contiguous(), "scale_rate must be contiguous");
  TORCH_CHECK(grad_voxel_out.sizes() == voxel_out.sizes() && grad_pcds_feat.sizes() == pcds_feat.sizes(),
              "gradient shapes must match their tensors");
  TORCH_CHECK(pcds_ind.scalar_type() == pcds_feat.scalar_type() && voxel_out.scalar_type() == pcds_feat.scalar_type() &&
              grad_voxel_out.scalar_type() == pcds_feat.scalar_type() && grad_pcds_feat.scalar_type() == pcds_feat.scalar_type(),
              "pcds_feat / pcds_ind / voxel_out / gradient dtypes differ");
  TORCH_CHECK(pcds_ind.device() == pcds_feat.device() && voxel_out.device() == pcds_feat.device() &&
              grad_voxel_out.device() == pcds_feat.device() && grad_pcds_feat.device() == pcds_feat.device(),
              "all tensors must live on pcds_feat's device");
  Geometry g = geometry(pcds_feat, pcds_ind, voxel_out, scale_rate);
  c10::DeviceGuard guard(pcds_feat.device());
  int rc = smos_voxel_maxpool_bwd(pcds_feat.data_ptr(), g.feat_stride.data(), pcds_ind.data_ptr(), voxel_out.data_ptr(),
                                  grad_voxel_out.data_ptr(), g.out_stride.data(), grad_pcds_feat.data_ptr(), g.bs, g.c, g.n,
                                  g.d, g.out_size.data(), g.scale.data(), dtype_code(pcds_feat, "voxel_maxpooling_backward"),
                                  current_stream(pcds_feat));
  TORCH_CHECK(rc == SMOS_OK, "smos_voxel_maxpool_bwd: ", smos_last_error());
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("voxel_maxpooling_forward", &voxel_maxpooling_forward, "maxpooling forward (gfx950, libsmos_hip)");
  m.def("voxel_maxpooling_backward", &voxel_maxpooling_backward, "maxpooling backward (gfx950, libsmos_hip)");
}
#endif  // SMOS_SHIM_POINT_DEEP

#ifdef SMOS_SHIM_MSDA
namespace {

struct Dims {
  int64_t n, s, m, d, l, lq, p;
};

Dims msda_dims(const at::Tensor& value, const at::Tensor& spatial_shapes, const at::Tensor& level_start_index,
               const at::Tensor& sampling_loc, const at::Tensor& attn_weight, int im2col_step) {
  // the checks of ms_deform_attn_cuda.cu:28-52
  TORCH_CHECK(value.is_cuda(), "Not implemented on the CPU");
  TORCH_CHECK(value.is_contiguous(), "value tensor has to be contiguous");
  TORCH_CHECK(spatial_shapes.is_contiguous(), "spatial_shapes tensor has to be contiguous");
  TORCH_CHECK(level_start_index.is_contiguous(), "level_start_index tensor has to be contiguous");
  TORCH_CHECK(sampling_loc.is_contiguous(), "sampling_loc tensor has to be contiguous");
  TORCH_CHECK(attn_weight.is_contiguous(), "attn_weight tensor has to be contiguous");
  TORCH_CHECK(spatial_shapes.is_cuda() && level_start_index.is_cuda() && sampling_loc.is_cuda() && attn_weight.is_cuda(),
              "every tensor must be a CUDA tensor");
  TORCH_CHECK(spatial_shapes.scalar_type() == at::kLong && level_start_index.scalar_type() == at::kLong,
              "spatial_shapes / level_start_index must be int64");
  TORCH_CHECK(value.dim() == 4 && sampling_loc.dim() == 6 && attn_weight.dim() == 5, "unexpected tensor ranks");
  Dims q{value.size(0), value.size(1), value.size(2), value.size(3), spatial_shapes.size(0), sampling_loc.size(1),
         sampling_loc.size(4)};
  // what the raw pointers handed to the C ABI are assumed to be: one dtype, one device, the exact shapes
  TORCH_CHECK(sampling_loc.scalar_type() == value.scalar_type() && attn_weight.scalar_type() == value.scalar_type(),
              "value / sampling_loc / attn_weight dtypes differ");
  TORCH_CHECK(sampling_loc.device() == value.device() && attn_weight.device() == value.device() &&
              spatial_shapes.device() == value.device() && level_start_index.device() == value.device(),
              "all tensors must live on value's device");
  TORCH_CHECK(spatial_shapes.dim() == 2 && spatial_shapes.size(1) == 2 && level_start_index.numel() == q.l,
              "spatial_shapes must be [L,2] and level_start_index [L]");
  TORCH_CHECK(sampling_loc.sizes() == at::IntArrayRef({q.n, q.lq, q.m, q.l, q.p, 2}), "sampling_loc must be [N,Lq,M,L,P,2]");
  TORCH_CHECK(attn_weight.sizes() == at::IntArrayRef({q.n, q.lq, q.m, q.l, q.p}), "attn_weight must be [N,Lq,M,L,P]");
  const int64_t step = std::min<int64_t>(q.n, im2col_step);
  TORCH_CHECK(step > 0 && q.n % step == 0, "batch(", q.n, ") must divide im2col_step(", step, ")");
  return q;
}

at::Tensor ms_deform_attn_forward(const at::Tensor& value, const at::Tensor& spatial_shapes,
                                  const at::Tensor& level_start_index, const at::Tensor& sampling_loc,
                                  const at::Tensor& attn_weight, const int im2col_step) {
  Dims q = msda_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step);
  c10::DeviceGuard guard(value.device());
  at::Tensor out = at::empty({q.n, q.lq, q.m * q.d}, value.options());
  int rc = smos_msda_fwd(value.data_ptr(), spatial_shapes.data_ptr<int64_t>(), level_start_index.data_ptr<int64_t>(),
                         sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(), q.n, q.s, q.m, q.d, q.l, q.lq, q.p,
                         dtype_code(value, "ms_deform_attn_forward"), current_stream(value));
  TORCH_CHECK(rc == SMOS_OK, "smos_msda_fwd: ", smos_last_error());
  return out;
}

std::vector<at::Tensor> ms_deform_attn_backward(const at::Tensor& value, const at::Tensor& spatial_shapes,
                                                const at::Tensor& level_start_index, const at::Tensor& sampling_loc,
                                                const at::Tensor& attn_weight, const at::Tensor& grad_output,
                                                const int im2col_step) {
  Dims q = msda_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step);
  TORCH_CHECK(grad_output.is_cuda() && grad_output.device() == value.device() && grad_output.scalar_type() == value.scalar_type(),
              "grad_output must be a CUDA tensor of value's dtype on value's device");
  TORCH_CHECK(grad_output.sizes() == at::IntArrayRef({q.n, q.lq, q.m * q.d}), "grad_output must be [N,Lq,M*D]");
  c10::DeviceGuard guard(value.device());
  at::Tensor go = grad_output.contiguous();
  at::Tensor grad_value = at::zeros_like(value);
  at::Tensor grad_loc = at::zeros_like(sampling_loc);
  at::Tensor grad_attn = at::zeros_like(attn_weight);
  int rc = smos_msda_bwd(go.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr<int64_t>(),
                         level_start_index.data_ptr<int64_t>(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
                         grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(), q.n, q.s, q.m, q.d, q.l, q.lq, q.p,
                         dtype_code(value, "ms_deform_attn_backward"), current_stream(value));
  TORCH_CHECK(rc == SMOS_OK, "smos_msda_bwd: ", smos_last_error());
  return {grad_value, grad_loc, grad_attn};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("ms_deform_attn_forward", &ms_deform_attn_forward, "ms_deform_attn_forward");
  m.def("ms_deform_attn_backward", &ms_deform_attn_backward, "ms_deform_attn_backward");
}
#endif  // SMOS_SHIM_MSDA
