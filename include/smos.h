/*
 * smos.h -- C ABI of libsmos_hip.so: the MI355X (gfx950) kernels of the StreamMOS
 * streaming-inference path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no torch types.  Every entry point
 *   - takes DEVICE pointers for tensors (HBM resident), HOST pointers for small shape/stride/scale
 *     vectors unless the comment says "device",
 *   - launches asynchronously on the hipStream_t passed as `stream` (never on the legacy default
 *     stream, unlike the reference's deep_point launches, point_deep_cuda_kernel.cu:153-160),
 *   - allocates nothing, synchronises nothing, keeps no global state (graph-capturable; the one exception is the test
 *     hook smos_debug_set_conv_grid_cap below),
 *   - returns SMOS_OK or an error code; smos_last_error() gives the thread-local message that the
 *     Python shims turn into RuntimeError (the reference raises c10::Error -> RuntimeError,
 *     point_deep_cuda.cpp:11-13).
 *
 * Each function names the reference interface it replaces (paths relative to the reference root).
 */
#ifndef SMOS_H_
#define SMOS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMOS_ABI_VERSION 1

enum smos_status {
  SMOS_OK = 0,
  SMOS_ERR_ARG = 1,          /* null pointer, bad size, unsupported rank */
  SMOS_ERR_UNSUPPORTED = 2,  /* dtype / shape outside what this build implements */
  SMOS_ERR_LAUNCH = 3        /* the HIP runtime reported a launch error */
};

enum smos_dtype { SMOS_F32 = 0, SMOS_F16 = 1, SMOS_F64 = 2 };

typedef void* smos_stream_t; /* hipStream_t */

int smos_abi_version(void);
const char* smos_last_error(void);
/* Test hook (process-wide, not part of the data path; the one piece of state the library keeps): caps the grid of the
 * persistent convolution kernels (smos_conv_cl / _rows_cl / _wino_cl / _wino1d_cl) at `blocks`, 0 = no cap.  It makes one
 * block walk several work items -- the item-boundary code paths -- on shapes small enough to check against float64. */
int smos_debug_set_conv_grid_cap(int32_t blocks);

/* --------------------------------------------------------------------------------------------
 * Point -> grid max-pool scatter.
 * Replaces point_deep.cuda_kernel.voxel_maxpooling_forward (deep_point/src/point_deep_cuda.cpp:20-37,
 * kernels deep_point/src/point_deep_cuda_kernel.cu:24-99) -- three launches, an int64 index scratch
 * and CAS-loop float atomics there; one fused launch with native integer atomic max here.
 *
 *   feat   [BS, C, N]      element strides feat_stride[3] (reference layout: {C*N, N, 1})
 *   ind    [BS, N, D]      contiguous, same dtype as feat; D <= 4
 *   out    [BS, C, D1..Dn] element strides out_stride[2+D]; MUST be zero-filled by the caller
 *                          (the reference's caller does so, deep_point/__init__.py:26)
 *   voxel_max_idx [BS, N]  int64, pre-filled with -1 by the caller, or NULL to skip it; receives
 *                          bs*out_stride[0] + sum_d cell_d*out_stride[2+d] for kept points
 *   out_size[D], scale[D]  host
 *   flag_ws                device int32 scratch (4 bytes); used to detect negative features, which take
 *                          a slower exact path (see DESIGN.md "VoxelMaxPool")
 * cell_d = (int64)((float)ind_d * scale_d) truncated toward zero; a point is kept iff every
 * 0 <= cell_d < out_size[d].  Occupied cell = max over its members; empty cell stays 0.
 */
int smos_voxel_maxpool_fwd(const void* feat, const int64_t* feat_stride, const void* ind, void* out,
                           const int64_t* out_stride, int64_t* voxel_max_idx, int64_t BS, int64_t C,
                           int64_t N, int32_t D, const int64_t* out_size, const float* scale,
                           int32_t dtype, int32_t* flag_ws, smos_stream_t stream);

/* Replaces point_deep.cuda_kernel.voxel_maxpooling_backward (point_deep_cuda.cpp:39-57, kernel
 * point_deep_cuda_kernel.cu:109-132): grad_feat[b,c,n] = grad_out[cell] iff out[cell] == feat[b,c,n].
 * grad_feat must be zero-filled by the caller; grad_out uses out's strides. */
int smos_voxel_maxpool_bwd(const void* feat, const int64_t* feat_stride, const void* ind, const void* out,
                           const void* grad_out, const int64_t* out_stride, void* grad_feat, int64_t BS,
                           int64_t C, int64_t N, int32_t D, const int64_t* out_size, const float* scale,
                           int32_t dtype, smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Grid -> point bilinear gather.
 * Replaces networks/backbone.py:453-475 (BilinearSample = F.grid_sample, bilinear, zeros padding,
 * align_corners=True) including its float32 normalise / un-normalise round trip.
 *   grid  [B, C, H, W]  element strides grid_stride[4]  (NCHW or NHWC both fine)
 *   coord [B, N, K]     contiguous, K >= 2; row = coord[...,0]*scale[0], col = coord[...,1]*scale[1]
 *   out   [B, C, N]     element strides out_stride[3]
 */
int smos_bilinear_gather_fwd(const float* grid, const int64_t* grid_stride, const float* coord, int32_t K,
                             float* out, const int64_t* out_stride, int64_t B, int64_t C, int64_t H,
                             int64_t W, int64_t N, const float* scale, smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Multi-scale deformable attention sampler, forward.
 * Replaces MultiScaleDeformableAttention.ms_deform_attn_forward (deformattn/src/vision.cpp:13-16 ->
 * deformattn/src/cuda/ms_deform_attn_cuda.cu:20-80, kernel ms_deform_im2col_cuda.cuh:237-299).
 *   value [N, S, M, D]; spatial_shapes [L,2] int64 DEVICE; level_start_index [L] int64 DEVICE
 *   sampling_loc [N, Lq, M, L, P, 2]; attn_weight [N, Lq, M, L, P]; out [N, Lq, M*D]  (all contiguous)
 * dtype: SMOS_F32 or SMOS_F64 (the reference dispatches float/double only, ms_deform_attn_cuda.cu:64).
 */
int smos_msda_fwd(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                  const void* sampling_loc, const void* attn_weight, void* out, int64_t N, int64_t S,
                  int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P, int32_t dtype,
                  smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * TTA reduce: softmax over classes, mean over the B test-time-augmentation variants, argmax.
 * Replaces val_StreamMOS.py:97-98,113.  pred [B, K, N] float32 contiguous (K <= 8) -> labels [N] uint8,
 * prob [N, K] float32 (optional, may be NULL).
 */
int smos_tta_argmax(const float* pred, int64_t B, int64_t K, int64_t N, uint8_t* labels, float* prob,
                    smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Voxel voting (voxel_voting.py:38-91,214-242).  The vote table is a dense array of
 * SMOS_VOTE_CELLS packed 64-bit words (three 21-bit class counters per voxel) that stays in HBM;
 * the caller zero-fills it per frame (smos_vote_clear), accumulates the 8 history frames and the
 * current frame, then resolves the current frame's labels.
 *
 *   pts [n, pt_stride] float32 rows (x, y, z, ...) in the frame's OWN sensor coordinates
 *   labels [n] uint8 in {0,1,2}
 *   pose_diff: host, 16 doubles row-major = inv(P_cur) * P_frame, or NULL for the current frame;
 *              applied in float64 then rounded to float32 (datasets/utils.py:116-126)
 *   recip_quantize: 0 = float32 true division (numpy / torch-CPU behaviour, the pinned one),
 *                   1 = multiply by the float32 reciprocal of the cell size (what torch's CUDA div by a
 *                       Python scalar does in voxel_voting.py:86-88 when run on a GPU)
 * Crop (utils/transforms.py:151-161) and Quantize (voxel_voting.py:77-91) are fused in.
 */
#define SMOS_VOTE_NX 512
#define SMOS_VOTE_NY 512
#define SMOS_VOTE_NZ 30
#define SMOS_VOTE_CELLS (512LL * 512LL * 30LL)

int smos_vote_clear(uint64_t* table, smos_stream_t stream);
int smos_vote_accumulate(const float* pts, int64_t n, int64_t pt_stride, const uint8_t* labels,
                         const double* pose_diff, int32_t recip_quantize, uint64_t* table,
                         smos_stream_t stream);
/* The same accumulation for a whole voting window in one launch (voxel_voting.py:214-238 loops over the 8 history frames
 * and the current one): host arrays of `count` device pointers / sizes; pose_diff[f] = 16 host doubles or NULL (identity,
 * the current frame).  Integer adds commute, so the table equals the one `count` single-frame calls produce. */
int smos_vote_accumulate_frames(int32_t count, const float* const* pts, const int64_t* n, const int64_t* pt_stride,
                                const uint8_t* const* labels, const double* const* pose_diff, int32_t recip_quantize,
                                uint64_t* table, smos_stream_t stream);
/* out_labels[i] = argmax of the voxel of point i (ties -> lowest class) if the point survives the crop,
 * else labels[i]; lut (device, 256 int32 entries, e.g. {0:0,1:9,2:251}) is applied when non-NULL. */
int smos_vote_resolve(const float* pts, int64_t n, int64_t pt_stride, const uint8_t* labels,
                      int32_t recip_quantize, const uint64_t* table, const int32_t* lut,
                      int32_t* out_labels, smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Instance-level voting (voxel_instance_voting.py:144-193): the two heavy steps of cluster().
 *
 * smos_dbscan: sklearn.cluster.DBSCAN(eps, min_samples).fit_predict(pts[:, :3]) (:150-153) on the device.
 *   labels[i] (device, int32[n]) = NAME of point i's cluster = index of the cluster's lowest-index core point, or -1 for
 *   noise.  scikit-learn numbers clusters 0,1,2.. in the order of exactly that index, so ranking the distinct names
 *   gives its labels; a border point gets the smallest name among the clusters it touches (the one scikit-learn expands
 *   first).  Distances: float32 coordinates widened to float64, d2 = dx*dx + dy*dy + dz*dz, d2 <= eps*eps.
 *   work: device scratch of smos_dbscan_work_bytes(n) bytes, 256-byte aligned (x-sorted copy of the points, the
 *   permutation, labels, hipCUB sort space).  The call synchronises `stream` once per 4 propagation sweeps (it reads
 *   back a "changed" flag); max_sweeps bounds the loop.
 * smos_box_vote: counts[k*3 + c] (device, uint32, caller zero-fills) += number of points of class c in {1,2} of one
 *   frame that survive the voting crop (utils/transforms.py:151-161) and lie inside the closed axis-aligned box
 *   boxes[k] = (lo_x, lo_y, lo_z, hi_x, hi_y, hi_z) float32 -- in_hull() of :62-76 for a box.  pose_diff as in
 *   smos_vote_accumulate (NULL for the current frame).  At most 2048 boxes per call.
 */
int64_t smos_dbscan_work_bytes(int64_t n);   /* 0 for n == 0, -1 on failure */
int smos_dbscan(const float* pts, int64_t n, int64_t pt_stride, double eps, int32_t min_samples, int32_t* labels,
                void* work, int64_t work_bytes, int32_t max_sweeps, smos_stream_t stream);
int smos_box_vote(const float* pts, int64_t n, int64_t pt_stride, const uint8_t* labels, const double* pose_diff,
                  const float* boxes, int32_t K, uint32_t* counts, smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Fused elementwise epilogues of the encoder (inference: BatchNorm scale folded into the conv weights on
 * the host, so every conv -> BN -> ReLU (-> add -> ReLU) chain of the reference collapses into one
 * pass).  All tensors float32; a "plane" is one contiguous (batch, channel) H*W slab, addressed as
 * base + b*stride_b + c*stride_c (element strides), which lets an output land inside a channel slice of a
 * concatenation buffer.
 */
/* out = act(x + bias[c] (+ res)); act: 0 none, 1 ReLU, 2 LeakyReLU(0.01).  bias / res may be NULL.
 * Replaces BN+ReLU after a conv (networks/backbone.py:136-159, multi_view_encoder.py:460-497). */
int smos_bias_act(const float* x, int64_t xs_b, int64_t xs_c, const float* bias, const float* res, int64_t rs_b,
                  int64_t rs_c, float* out, int64_t os_b, int64_t os_c, int64_t B, int64_t C, int64_t HW,
                  int32_t act, smos_stream_t stream);
/* DownSample2D tail (networks/backbone.py:29-34): out = relu(a + bias[c] + maxpool3x3(p; stride, pad 1)).
 * a [B,C,Ho,Wo] and p [B,C,H,W] with full element strides (NCHW or channels-last); out planes contiguous. */
int smos_downsample_epilogue(const float* a, const int64_t* a_stride, const float* p, const int64_t* p_stride,
                             const float* bias, float* out, int64_t os_b, int64_t os_c, int64_t B, int64_t C,
                             int64_t H, int64_t W, int32_t stride, smos_stream_t stream);
/* BasicBlock tail with ChannelAtt (networks/backbone.py:87-102,151-159):
 * g = sigmoid(w2 * relu(w1 * mean_hw(y + bias) + b1) + b2); out = relu((y + bias) * g + xres).
 * w1 [Cr,C], w2 [C,Cr] (the 1x1 conv weights), sums_ws: device scratch of B*C floats. */
int smos_channel_gate_residual(const float* y, int64_t ys_b, int64_t ys_c, const float* bias, const float* w1,
                               const float* b1, const float* w2, const float* b2, const float* xres, int64_t rs_b,
                               int64_t rs_c, float* out, int64_t os_b, int64_t os_c, float* sums_ws, int64_t B,
                               int64_t C, int64_t Cr, int64_t HW, smos_stream_t stream);
/* Decoder input (networks/multi_view_encoder.py:441-447): bilinear resize (align_corners=True) of up to three
 * NCHW maps to (Ho, Wo), concatenated along channels into out [B, sum C_i, Ho, Wo] (contiguous).
 * src[i]: device pointers (host array), per-source C/H/W and batch/channel strides (host arrays). */
int smos_upsample_concat(const float* const* src, const int64_t* src_c, const int64_t* src_h, const int64_t* src_w,
                         const int64_t* src_sb, const int64_t* src_sc, int32_t n_src, float* out, int64_t B,
                         int64_t Ho, int64_t Wo, smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Fused point-side kernels of the inference engine (csrc/point_fused.hip).  Scatter targets are
 * channels-last and zero-filled by the caller; features are assumed >= 0 (post-ReLU), as on every call
 * site of the model.
 */
/* point_pre + input scatter (models/StreamMOS.py:101-103): per-point MLP cin->cmid->cout (BatchNorm folded
 * into w/b, ReLU after both layers; only 7->64->64 is built) on xyzi [B*T, cin, N], max-scattered into
 * bev [B, H, W, T*cout] at cell (int(coord0), int(coord1)), coord [B*T, N, K].  pts_out (optional): the
 * features of the t == 0 scan as rows pts_out[b*po_b + n*po_n + c]. */
int smos_pointnet_scatter(const float* xyzi, const float* coord, int32_t K, const float* w1, const float* b1,
                          const float* w2, const float* b2, float* bev, float* pts_out, int64_t po_b, int64_t po_n,
                          int64_t B, int64_t T, int64_t N, int64_t H, int64_t W, int32_t cin, int32_t cmid,
                          int32_t cout, smos_stream_t stream);
/* Sparse DownSample2D on the input grid (networks/backbone.py:136-159 as used by header_bev[0],
 * multi_view_encoder.py:340-347): see csrc/stem.hip.  All buffers are device memory owned by the caller; no call reads
 * anything back to the host (row counts stay in `meta`), so the sequence is graph-capturable.
 *   smos_stem_mark    flags (int32 [4, B, H/2, W/2]; index = parity class (y&1)*2+(x&1), sample, y>>1, x>>1) <- 1 for
 *                     every cell a point of coord [B, T, N, K] falls into.  flags must be all zero on entry (zero it once;
 *                     smos_stem_scan leaves it all zero again).  Also re-arms scan_state (uint64
 *                     [smos_stem_scan_state_words(B*H*W)]) for the smos_stem_scan call that follows on the stream.
 *   smos_stem_scan    one pass over the flags: row_cell[row] <- natural cell id (b*H + y)*W + x; row_of [B,H,W] <- row id
 *                     or -1; meta (12 x int32): [4..7] first row of each class, [8..11] one past its last row (meta[11] =
 *                     number of rows); flags cleared; if rows != NULL, rows[0 .. meta[11]) (row_floats floats each: the
 *                     compact table of smos_pointnet_scatter_rows) zero-filled.
 *   smos_stem_gemm    y4[cls] (device, capacity B*(H/2)*(W/2) rows of (taps+1)*Cout floats; taps = 1,2,2,4) <-
 *                     occupied rows of bev [B*H*W, Cin] times the class weights; with row_cell == NULL, bev is the
 *                     compact row table itself (row r of the table = global row r).  wprep4[cls]: weights of the class in
 *                     MFMA operand order [(taps+1)][Cin/2][64]: entry (mt, s, lane) = W[mt*32 + (lane & 31)][(lane >> 5) *
 *                     (Cin/2) + s], where W stacks the class's 3x3 taps (ky-major over the kernel rows / columns that
 *                     reach an output pixel) and the 1x1 pool-branch weights last.  Built for Cin = 192, Cout = 32.
 *   smos_stem_epilogue out[b,ho,wo,c] <- relu(sum_taps Y[cell][slot][c] + max_window(occupied ? Y[cell][q][c] : 0) +
 *                     bias[c]); out is channels-last [B, H/2, W/2, *] with row pitch out_pitch; C must be 32.
 * y4 / wprep4 are HOST arrays of 4 device pointers. */
int64_t smos_stem_scan_state_words(int64_t cells);
int smos_stem_mark(const float* coord, int32_t K, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W, int32_t* flags,
                   uint64_t* scan_state, smos_stream_t stream);
int smos_stem_scan(int32_t* flags, int64_t B, int64_t H, int64_t W, uint64_t* scan_state, int32_t* row_cell, int32_t* row_of,
                   int32_t* meta, float* rows, int64_t row_floats, smos_stream_t stream);
int smos_stem_gemm(const float* bev, const int32_t* row_cell, const int32_t* meta, const float* const* wprep4, float* const* y4,
                   int64_t Cin, int64_t Cout, smos_stream_t stream);
int smos_stem_epilogue(const float* const* y4, const int32_t* meta, const int32_t* row_of, const float* bias, float* out,
                       int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t C, smos_stream_t stream);
/* The sampler of smos_msda_fwd fed by the raw query projection of a DeformAttnLayer (multi_view_encoder.py:300-316):
 * qp [N, H*W, M*P*2 + M*P] = per token the sampling offsets (x, y per head and point) followed by the attention logits;
 * single level H x W, queries on the same H x W lattice with the cell centres as reference points.  Folds softmax over the
 * P logits, off / (W, H) and + reference point into the kernel.  value [N, H*W, M, D], out [N, H*W, M*D]; D == 32, P <= 8. */
int smos_msda_fwd_qp(const float* value, const float* qp, float* out, int64_t N, int64_t H, int64_t W, int64_t M, int64_t D,
                     int64_t P, smos_stream_t stream);

/* Temporal fusion on the matrix cores (csrc/tfusion.hip; networks/multi_view_encoder.py:285-321 DeformAttnLayer,
 * deformattn/modules/ms_deform_attn.py:94-115 MSDeformAttn's projections).  d_model = 128.
 * smos_tfusion_project: up to eight token-wise Linear jobs y = W x (+ b) in one launch -- the projections of a frame that depend
 *   on no previous layer (value_proj of every layer, the first layer's [sampling_offsets | attention_weights]) and the
 *   decoder's tap products (the [B Hs Ws, 128] x [128, 9 * 128] GEMMs in front of smos_upconv_xy).  Host arrays of n_jobs
 *   entries; x[j] [tokens[j], *] rows of pitch x_pitch[j] floats (>= 128); wstream[j] = the weights as 16 x 16 blocks in MFMA
 *   operand order (streammos_amd.ops.tfusion_pack_linear: cout rounded up to a multiple of 64, zero padded); bias[j] [cout]
 *   or NULL; out[j] [tokens[j], *] rows of pitch out_pitch[j] >= cout (a job may fill a column range of a wider matrix); cout a
 *   multiple of 4, <= 2048.
 * smos_tfusion_layer: everything of one layer behind its sampler in one launch:
 *   q1 = norm1(query + output_proj(sampled)); out = norm2(q1 + linear2(relu(linear1(q1)))); and, if qp_next != NULL,
 *   qp_next = W_q out + b_q -- the NEXT layer's offset / logit projection (nq <= 64 channels), so that the following
 *   smos_msda_fwd_qp can start at once.  wstream / params = streammos_amd.ops.TfusionLayer (sizes:
 *   smos_tfusion_layer_stream_floats / _param_floats); ffn a multiple of 32. */
int smos_tfusion_project(int32_t n_jobs, const float* const* x, const int64_t* x_pitch, const float* const* wstream,
                         const float* const* bias, float* const* out, const int64_t* out_pitch, const int64_t* cout,
                         const int64_t* tokens, smos_stream_t stream);
int64_t smos_tfusion_layer_param_floats(int64_t ffn);
int64_t smos_tfusion_layer_stream_floats(int64_t ffn, int32_t has_next);
int smos_tfusion_layer(const float* sampled, const float* query, int64_t q_pitch, const float* wstream, const float* params,
                       float* out, int64_t o_pitch, float* qp_next, int64_t nq, int64_t tokens, int64_t ffn, float eps1,
                       float eps2, smos_stream_t stream);

/* out = LayerNorm(x + res) over rows of C floats (res may be NULL): the residual + norm steps of a DeformAttnLayer
 * (multi_view_encoder.py:314-320).  C in {64, 128, 256, 512}; biased variance, eps inside the square root (torch). */
int smos_add_layer_norm(const float* x, const float* res, const float* gamma, const float* beta, float* out, int64_t rows,
                        int64_t C, float eps, smos_stream_t stream);

/* General channels-last convolution on the matrix cores with the epilogue fused (csrc/conv_igemm.hip):
 *   out = act(conv_{KH x KW, stride, pad}(x) + bias [+ res]);  act 0 none, 1 ReLU, 2 LeakyReLU(0.01).
 * Replaces every conv2d -> BatchNorm (folded by the caller) -> ReLU / LeakyReLU (-> + residual -> ReLU) chain of
 * networks/backbone.py:9-34,136-159 and networks/multi_view_encoder.py:460-497 (cuDNN in the reference).
 * x [B, H, W, *], res / out [B, Ho, Wo, *]: row pitches in floats (multiples of 4; channel slices of wider buffers are
 * fine), 16-byte aligned, each tensor < 2 GiB.  Cin % 32 == 0, Cout % (32 * mt) == 0, mt in {1, 2, 4} = 32-channel output
 * blocks per wave (a tuning knob), KH, KW <= 7, stride 1 or 2.  bias / res may be NULL.
 * wprep: Cout * Cin * KH * KW floats in MFMA operand order for the chosen mt; with stage = (ky * KW + kx) * (Cin / 32) + cc:
 *   wprep[((((ct * n_stage + stage) * 4 + i4) * mt + m) * 64 + lane) * 4 + c]
 *       = w[ct * 32 * mt + m * 32 + (lane & 31)][cc * 32 + 8 * i4 + 4 * (lane >> 5) + c][ky][kx].
 * chan_sums (may be NULL; needs res == NULL): [B][chunks][Cout] with chunks = smos_conv_cl_sum_chunks(Ho, Wo); entry
 *   (b, chunk, c) = sum of out[b, y, x, c] over one output row segment of 32 pixels (chunk = ((y / 4) * ceil(Wo / 32) +
 *   x / 32) * 4 + y % 4; segments outside the image hold 0): the global-average-pool input of a ChannelAtt block
 *   (networks/backbone.py:57-73) without another pass over the map.  Summation order is fixed (run-to-run identical). */
int64_t smos_conv_cl_sum_chunks(int64_t Ho, int64_t Wo);
/* smos_conv_cl for stride 1, "same" padding (pad = K / 2), KW in {3, 5, 7}, odd KH <= 7, mt in {1, 2} (32 / 64 output
 * channels per block), with the input rows staged through LDS once per kernel row and 32-channel chunk
 * (csrc/conv_rows.hip): one coalesced request per row instead of one per tap.  Same operands and epilogue; the weight block
 * is ordered (ky, cin chunk, kx) inside a tile:
 *   wprep[((((ct * KH + ky) * (Cin / 32) + cc) * KW + kx) * 4 + i4) * mt + m][lane][c]
 *       = w[ct * 32 * mt + m * 32 + (lane & 31)][cc * 32 + 8 * i4 + 4 * (lane >> 5) + c][ky][kx]. */
int smos_conv_rows_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, const float* res, int64_t res_pitch,
                      float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int32_t KH,
                      int32_t KW, int32_t mt, int32_t act, float* chan_sums, smos_stream_t stream);
int smos_conv_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, const float* res, int64_t res_pitch,
                 float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int32_t KH,
                 int32_t KW, int32_t stride, int32_t pad_h, int32_t pad_w, int32_t mt, int32_t act, float* chan_sums,
                 smos_stream_t stream);
/* smos_conv_cl for the stride-1 3x3 "same" layers in the Winograd F(2x2, 3x3) form (csrc/conv_wino.hip): 4 instead of 9
 * multiply-adds per output and channel pair, arithmetic still plain fp32 -- the weights are transformed on the host in
 * float64 (U = G g G^T, rounded once), the input and output transforms are additions.  Replaces the same reference layers as
 * smos_conv_cl (networks/backbone.py:136-159, networks/multi_view_encoder.py:446-447,478-497) where the kernel is 3x3 and
 * the stride 1.  Cin % 16 == 0, mb in {1, 2}: 16 * mb output channels per block, Cout % (16 * mb) == 0.  Operand order:
 *   wprep[((((ct * (Cin / 16) + cc) * 4 + i) * mb + m) * 4 + xi) * 64 + lane][nu]
 *       = U[xi][nu] of w[ct * 16 * mb + m * 16 + (lane & 15)][cc * 16 + 4 * (lane >> 4) + i],  U = G w G^T (4 x 4),
 *   G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]].
 * chan_sums (may be NULL; needs res == NULL): [B][chunks][Cout], chunks = smos_conv_wino_sum_chunks(H, W); entry
 *   (b, chunk, c) = sum of out[b, y, x, c] over two output rows x 32 columns (chunk = ((y / 8) * ceil(W / 32) + x / 32) * 4
 *   + (y % 8) / 2; parts outside the image contribute 0).  Summation order is fixed (run-to-run identical). */
int64_t smos_conv_wino_sum_chunks(int64_t H, int64_t W);
int smos_conv_wino_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, const float* res, int64_t res_pitch,
                      float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int32_t mb,
                      int32_t act, float* chan_sums, smos_stream_t stream);

/* act(conv(x, w) + bias) for a stride-1 KH x KW kernel with one extent 3 and the other 5 or 7 ("same" padding) in the 1-D
 * Winograd F(2, 3) form along the 3-tap axis (csrc/conv_wino1d.hip): 4 instead of 6 multiply-adds per pair of outputs, long-
 * axis tap and channel pair, fp32 throughout.  wprep = U = G g per long-axis tap, computed in float64 and rounded once, in
 * operand order [cout tile][cin chunk of 16][k-step i][long tap][mb][lane = q * 16 + m][position] (ops.conv_wino1d_prepare);
 * Cin % 16 == 0, Cout % (16 * mb) == 0, mb in {1, 2}; channels-last maps with row pitches in floats, 16-byte aligned.
 * Replaces nn.Conv2d + BatchNorm2d + ReLU of the two parallel branches of an Unbalance_BasicBlock
 * (networks/multi_view_encoder.py:478-497). */
int smos_conv_wino1d_cl(const float* x, int64_t x_pitch, const float* wprep, const float* bias, float* out, int64_t out_pitch,
                        int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, int32_t mb, int32_t act,
                        smos_stream_t stream);

/* conv3x3(bilinear_up(x)) without upsampling x (decoder conv_1, multi_view_encoder.py:441-453; csrc/upconv.hip).
 * z [B, Hs, Ws, 9*C] = the nine tap products W_{ky,kx} x at the source resolution (tap t = 3 ky + kx occupies channels
 * [t*C, (t+1)*C)), computed by the caller with one GEMM.
 *   smos_upconv_xpass  t [B, 3, Hs, Wo, C] <- sum_kx up_x(z tap (ky,kx))[X + kx - 1]   (taps leaving [0, Wo) dropped)
 *   smos_upconv_ypass  out [B, Ho, Wo, *] (row pitch out_pitch) <- act(conv_a + bias + sum_src sum_ky up_y(t_src)[Y + ky - 1]);
 *                      conv_a [B, Ho, Wo, *] (row pitch a_pitch) is the direct convolution of the channels that are not
 *                      upsampled; t2 may be NULL; act 0 none, 1 ReLU, 2 LeakyReLU(0.01); interpolation with ATen's
 *                      align_corners=True weights.
 *   smos_upconv_xy     the two passes in one launch, t kept in registers (same operations in the same order: same
 *                      results): out <- act(conv_a + bias + sum over z1 [B, H1, W1, 9*C], z2 [B, H2, W2, 9*C] (one may
 *                      be NULL)).  Only for sources of at most half the output height: smos_upconv_xy_ok(Hs, Ho) != 0. */
int smos_upconv_xy_ok(int64_t Hs, int64_t Ho);
int smos_upconv_xy(const float* conv_a, int64_t a_pitch, const float* bias, const float* z1, int64_t H1, int64_t W1, const float* z2,
                   int64_t H2, int64_t W2, float* out, int64_t out_pitch, int64_t B, int64_t Ho, int64_t Wo, int64_t C, int32_t act,
                   smos_stream_t stream);
int smos_upconv_xpass(const float* z, float* t, int64_t B, int64_t Hs, int64_t Ws, int64_t C, int64_t Wo, smos_stream_t stream);
int smos_upconv_ypass(const float* conv_a, int64_t a_pitch, const float* bias, const float* t1, int64_t H1, const float* t2,
                      int64_t H2, float* out, int64_t out_pitch, int64_t B, int64_t Ho, int64_t Wo, int64_t C, int32_t act,
                      smos_stream_t stream);

/* CatFusion + PredBranch (networks/backbone.py:387-413, 188-196): rows [B*N, K1] (row pitch in floats) ->
 * 1x1 K1->M1 + ReLU -> 1x1 M1->M2 + ReLU -> 1x1 M2->M3 + bias, out [B, M3, N]; one kernel, intermediates in registers
 * (csrc/point_head.hip).  Built for 192 -> 96 -> 64 -> M3 <= 32.  wprep: smos_point_head_weight_floats() floats =
 * the three weight matrices in MFMA operand order followed by the biases (b3 padded to 32):
 *   A1[(mt*96 + s)*64 + lane] = W1[mt*32 + (lane&31)][(lane>>5)*96 + s]
 *   A2[(mt*48 + s)*64 + lane] = W2[mt*32 + (lane&31)][ch(s, lane>>5)],  A3[s*64 + lane] = W3[lane&31][ch(s, lane>>5)] (0 beyond M3)
 *   with ch(s, h) = 32 (s >> 4) + 8 ((s & 15) >> 2) + 4 h + (s & 3)   (the accumulator order of the previous layer). */
int64_t smos_point_head_weight_floats(void);
int smos_point_head(const float* rows, int64_t row_pitch, const float* wprep, float* out, int64_t B, int64_t N, int64_t K1,
                    int64_t M1, int64_t M2, int64_t M3, smos_stream_t stream);
/* The same with the scan's padding tail left out: n_live (DEVICE int32, may be NULL = no tail) says how many points at the
 * front of every sample are real -- the reference pads every scan to frame_point_num with points at -1000
 * (datasets/data_StreamMOS.py:568-571) and cuts their predictions off again (val_StreamMOS.py:113); the logits of points
 * [*n_live, N) are written as zeros without being computed.  The streaming runner's form; AttNet.infer computes all N. */
int smos_point_head_live(const float* rows, int64_t row_pitch, const float* wprep, float* out, int64_t B, int64_t N, int64_t K1,
                         int64_t M1, int64_t M2, int64_t M3, const int32_t* n_live, smos_stream_t stream);

/* smos_pointnet_scatter with a COMPACT target: rows [n_rows, T*cout] (zero-filled by smos_stem_scan) instead of
 * the dense [B,H,W,T*cout] grid; the features of cell (b, y, x) go to row row_of[b][y][x] (smos_stem_scan). */
int smos_pointnet_scatter_rows(const float* xyzi, const float* coord, int32_t K, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* rows, const int32_t* row_of, float* pts_out,
                               int64_t po_b, int64_t po_n, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W, int32_t cin,
                               int32_t cmid, int32_t cout, smos_stream_t stream);
/* The same, told how many points of the current scan are real (n_live: DEVICE int32 or NULL; see smos_point_head_live): the point
 * rows of the padding tail [*n_live, N) are not computed and not written.  The scatter itself is unaffected (padding points lie
 * outside the grid). */
int smos_pointnet_scatter_rows_live(const float* xyzi, const float* coord, int32_t K, const float* w1, const float* b1,
                                    const float* w2, const float* b2, float* rows, const int32_t* row_of, float* pts_out,
                                    int64_t po_b, int64_t po_n, int64_t B, int64_t T, int64_t N, int64_t H, int64_t W, int32_t cin,
                                    int32_t cmid, int32_t cout, const int32_t* n_live, smos_stream_t stream);
/* BilinearSample (networks/backbone.py:453-475) of grid [B,C,Hg,Wg] (element strides grid_stride[4]) at
 * gcoord*gscale, fused with VoxelMaxPool of the result into out [B,Ho,Wo,C] (channels-last) at
 * int(scoord*sscale) (networks/multi_view_encoder.py:395-404,410-419).  out may be NULL (gather only);
 * pts_out (optional) receives the gathered features as rows pts_out[b*po_b + n*po_n + c]. C % 32 == 0. */
int smos_gather_scatter(const float* grid, const int64_t* grid_stride, const float* gcoord, int32_t Kg,
                        const float* gscale, const float* scoord, int32_t Ks, const float* sscale, float* out,
                        float* pts_out, int64_t po_b, int64_t po_n, int64_t B, int64_t C, int64_t Hg, int64_t Wg,
                        int64_t N, int64_t Ho, int64_t Wo, smos_stream_t stream);
/* channels-last [B, HW, C] -> NCHW planes dst[b*ds_b + c*ds_c + p] (a channel slice of a larger buffer). */
int smos_nhwc_to_nchw(const float* src, float* dst, int64_t ds_b, int64_t ds_c, int64_t B, int64_t C, int64_t HW,
                      smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Device-side validation preprocessing (csrc/preprocess.hip): what DataloadVal does with
 * numpy per sample (datasets/data_StreamMOS.py:397-599, datasets/utils.py:98-192).
 *   range6     host doubles {x_lo, x_hi, y_lo, y_hi, z_lo, z_hi}     (config Voxel.range_*)
 *   bev_size3  host int64   {512, 512, 30}                            (config Voxel.bev_shape)
 *   rv4        host doubles {phi_hi, dphi, theta_hi, dtheta} in radians (datasets/utils.py:176-181)
 */
/* moved[i] = (float32)(pose_diff * (x, y, z, 1)) computed in float64, intensity carried; mask[i] = lo <= p < hi.
 * pose_diff: host, 16 doubles row-major, or NULL (identity, the current scan). */
int smos_prep_transform_mask(const float* scan, int64_t n, const double* pose_diff, const double* range6, float* moved,
                             int32_t* mask, smos_stream_t stream);
/* Writes scan t of every TTA variant v (x *= tta_sx[v], y *= tta_sy[v]; V <= 4):
 *   xyzi [V,T,7,N] = (x, y, z, intensity, dist, frac(x_quan), frac(y_quan)), coord [V,T,N,3] = (x,y,z)_quan,
 *   sphere [V,T,N,2] = (theta_quan, phi_quan); kept points are compacted in order (slot prefix[i]-1, prefix = inclusive
 * prefix sum of mask), slots >= count hold the reference's padding point (-1000, -1000, -4000, -1000). */
int smos_prep_emit(const float* moved, const int32_t* mask, const int32_t* prefix, int64_t n, int32_t t, int32_t T, int64_t N,
                   int32_t V, const float* tta_sx, const float* tta_sy, const double* range6, const int64_t* bev_size3,
                   const double* rv4, float* xyzi, float* coord, float* sphere, smos_stream_t stream);
/* raw[i] = mask[i] ? labels[prefix[i]-1] : 0   (val_StreamMOS.py:112-118) */
int smos_prep_unpad_labels(const uint8_t* labels, int64_t N, const int32_t* mask, const int32_t* prefix, int64_t n,
                           uint8_t* raw, smos_stream_t stream);

/* Multi-scale deformable attention, backward (training row f2).  Replaces
 * MultiScaleDeformableAttention.ms_deform_attn_backward (deformattn/src/cuda/ms_deform_attn_cuda.cu:83-153, kernels
 * ms_deform_im2col_cuda.cuh:87-159,301-920).  grad_out [N,Lq,M*D]; grad_value [N,S,M,D] MUST be zero-filled by the
 * caller (accumulated with atomic adds); grad_sampling_loc [N,Lq,M,L,P,2] and grad_attn_weight [N,Lq,M,L,P] are
 * fully written. */
int smos_msda_bwd(const void* grad_out, const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                  const void* sampling_loc, const void* attn_weight, void* grad_value, void* grad_sampling_loc,
                  void* grad_attn_weight, int64_t N, int64_t S, int64_t M, int64_t D, int64_t L, int64_t Lq, int64_t P,
                  int32_t dtype, smos_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Channels-last ("cl") engine kernels (csrc/cl_kernels.hip).  A map is [B, H, W, C] with C innermost; every tensor
 * argument has its own row pitch (elements between consecutive pixels / points), so it may be a channel slice of a
 * wider buffer.  P = number of rows (B*H*W).  All pointers 16-byte aligned, C and pitches multiples of 4.
 */
int smos_bias_act_cl(const float* x, int64_t x_pitch, const float* bias, const float* res, int64_t res_pitch, float* out,
                     int64_t out_pitch, int64_t P, int64_t C, int32_t act, smos_stream_t stream);
int smos_downsample_epilogue_cl(const float* a, int64_t a_pitch, const float* p, int64_t p_pitch, const float* bias, float* out,
                                int64_t out_pitch, int64_t B, int64_t C, int64_t H, int64_t W, int32_t stride,
                                smos_stream_t stream);
/* ws: device scratch of at least B*C*(ceil(HW/512)+1) floats (deterministic two-stage column sums + the gates). */
int smos_channel_gate_residual_cl(const float* y, int64_t y_pitch, const float* bias, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* xres, int64_t res_pitch, float* out,
                                  int64_t out_pitch, float* ws, int64_t ws_floats, int64_t B, int64_t C, int64_t Cr, int64_t HW,
                                  smos_stream_t stream);
/* The same block fed by the channel sums a smos_conv_cl launch left behind (chan_sums [B][chunks][C], see smos_conv_cl):
 * gate MLP + out = relu((y + bias) * gate + xres), no pass over y for the average pool.  gate_ws: B*C floats of scratch. */
int smos_channel_gate_apply_cl(const float* y, int64_t y_pitch, const float* bias, const float* w1, const float* b1,
                               const float* w2, const float* b2, const float* xres, int64_t res_pitch, float* out,
                               int64_t out_pitch, const float* chan_sums, int64_t chunks, float* gate_ws, int64_t B, int64_t C,
                               int64_t Cr, int64_t HW, smos_stream_t stream);
/* EXPERIMENTAL (off by default in the engine): n_layers (2 .. 12) stride-1 3x3 convolutions C -> C of one map size -- the 2 k
 * convolutions of k consecutive BasicBlocks (networks/backbone.py:136-159) -- as ONE launch of the Winograd kernel, every block
 * computing its work item of layer 0, 1, .. in turn behind a per-region dataflow wait instead of a launch boundary
 * (csrc/conv_wino_chain.hip).  Layer L reads x0 (L = 0) or outs[L - 1]; res_from[L]: -1 no residual, 0 = x0, j > 0 = outs[j - 1];
 * acts[L] as smos_conv_wino_cl; wprep[L] = ops.conv_wino_prepare(w, mb); chan_sums: channel sums of the last layer (or NULL).
 * Every layer needs its own output map.  ws: smos_conv_wino_chain_ws_ints int32 words, zeroed once; launch_no = 1, 2, .. counts
 * the launches that used this ws with these sizes; ws[last] != 0 afterwards = a wait gave up, results invalid.  C % (16 mb) == 0 and
 * B * ceil(H/8) * ceil(W/32) * C/(16 mb) <= resident blocks (else SMOS_ERR_ARG: use the launch-by-launch form).  Results equal the
 * separate smos_conv_wino_cl launches bit for bit. */
int64_t smos_conv_wino_chain_ws_ints(int64_t n_layers, int64_t B, int64_t H, int64_t W);
int smos_conv_wino_chain_cl(int32_t n_layers, const float* x0, int64_t x0_pitch, const float* const* wprep, const float* const* bias,
                            const int32_t* res_from, float* const* outs, const int64_t* out_pitches, const int32_t* acts,
                            float* chan_sums, int32_t* ws, int32_t launch_no, int64_t B, int64_t H, int64_t W, int64_t C,
                            int32_t mb, smos_stream_t stream);
/* BasicBlock.forward (networks/backbone.py:136-159: conv3x3-BN-ReLU, conv3x3-BN, optional ChannelAtt gate, + x, ReLU) on
 * channels-last maps as ONE foreign call: enqueues smos_conv_wino_cl twice (+ smos_channel_gate_apply_cl when gw1 != NULL) on
 * `stream` -- the same launches with the same arguments as the separate calls, so results are bit-identical; what is saved is
 * the host's per-call marshalling (csrc/blocks.hip).  u1 / u2: the Winograd weight blocks of the two convs (BN folded,
 * ops.conv_wino_prepare, same mb); b1 / b2 their biases; gw1 [Cr, C], gb1 [Cr], gw2 [C, Cr], gb2 [C]: the gate MLP or all NULL;
 * y: scratch map [B,H,W,*] (pitch y_pitch) for the first conv's output; out may not alias x or y; ws: smos_basic_block_ws_floats
 * floats of scratch, needed with a gate only.  Shape limits are those of smos_conv_wino_cl with Cin == Cout == C. */
int64_t smos_basic_block_ws_floats(int64_t B, int64_t H, int64_t W, int64_t C);
int smos_basic_block_cl(const float* x, int64_t x_pitch, const float* u1, const float* b1, const float* u2, const float* b2,
                        const float* gw1, const float* gb1, const float* gw2, const float* gb2, int64_t Cr, float* y,
                        int64_t y_pitch, float* out, int64_t out_pitch, float* ws, int64_t B, int64_t H, int64_t W, int64_t C,
                        int32_t mb, smos_stream_t stream);
/* Unbalance_BasicBlock.forward (networks/multi_view_encoder.py:478-497) as one foreign call: smos_conv_wino1d_cl for the
 * (kha x kwa) and (khb x kwb) branches into channels [0, C) and [C, 2C) of `both` (ReLU), then smos_conv_wino_cl 2C -> C over
 * `both` with residual x and ReLU into out.  ua / ub = ops.conv_wino1d_prepare, uc = ops.conv_wino_prepare, all with the same mb.
 * Same launches and arguments as the separate calls (bit-identical). */
int smos_unbalance_block_cl(const float* x, int64_t x_pitch, const float* ua, const float* ba, int64_t kha, int64_t kwa,
                            const float* ub, const float* bb, int64_t khb, int64_t kwb, const float* uc, const float* bc,
                            float* both, int64_t both_pitch, float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W,
                            int64_t C, int32_t mb, smos_stream_t stream);
/* Zero fill of n <= 4 channels-last views (rows[i] rows of row_floats[i] floats at pitch pitches[i], 16-byte aligned) in one
 * launch: the zero-initialised scatter-max targets of the cross-view transfers (networks/multi_view_encoder.py:395-420 create
 * theirs with torch.zeros inside VoxelMaxPool, deep_point/point_deep.py:27). */
int smos_zero_views_cl(int32_t n, float* const* ptrs, const int64_t* rows, const int64_t* row_floats, const int64_t* pitches,
                       smos_stream_t stream);
int smos_upsample_concat_cl(const float* const* src, const int64_t* src_c, const int64_t* src_h, const int64_t* src_w,
                            const int64_t* src_pitch, int32_t n_src, float* out, int64_t B, int64_t Ho, int64_t Wo,
                            smos_stream_t stream);
/* DownSample2D's tail with the pool branch computed on the fly (networks/backbone.py:105-134; csrc/downsample.hip):
 * out = relu(a + bias + maxpool3x3(conv1x1(x, w); stride, pad 1)) -- replaces smos_conv_cl (1x1, full resolution) +
 * smos_downsample_epilogue_cl; q = W x stays in LDS.  x [B,H,W,*] pitch x_pitch; wpairs = the [Cout, Cin] weights (BN folded) as
 * 16 x 16 blocks in MFMA operand order (streammos_amd.ops.pool_branch_prepare); a / out [B,Ho,Wo,*]; Cin == Cout in {32, 64, 128}
 * (128 at stride 2 only); stride 1 or 2. */
int smos_downsample_pool_branch(const float* x, int64_t x_pitch, const float* wpairs, const float* a, int64_t a_pitch,
                                const float* bias, float* out, int64_t out_pitch, int64_t B, int64_t H, int64_t W, int64_t Cin,
                                int64_t Cout, int32_t stride, smos_stream_t stream);
/* smos_gather_scatter with a channels-last source grid [B,Hg,Wg,*] and target [B,Ho,Wo,*]; C is 32 or 64. */
int smos_gather_scatter_cl(const float* grid, int64_t grid_pitch, const float* gcoord, int32_t Kg, const float* gscale,
                           const float* scoord, int32_t Ks, const float* sscale, float* out, int64_t out_pitch, float* pts_out,
                           int64_t po_b, int64_t po_n, int64_t B, int64_t C, int64_t Hg, int64_t Wg, int64_t N, int64_t Ho,
                           int64_t Wo, smos_stream_t stream);
/* The same with n_live (DEVICE int32 or NULL; see smos_point_head_live): point rows of the padding tail are not written. */
int smos_gather_scatter_cl_live(const float* grid, int64_t grid_pitch, const float* gcoord, int32_t Kg, const float* gscale,
                                const float* scoord, int32_t Ks, const float* sscale, float* out, int64_t out_pitch, float* pts_out,
                                int64_t po_b, int64_t po_n, int64_t B, int64_t C, int64_t Hg, int64_t Wg, int64_t N, int64_t Ho,
                                int64_t Wo, const int32_t* n_live, smos_stream_t stream);
/* The same with the coordinates as strided views: sample b, point n at coord + b * batch_stride + n * K floats (the first two
 * of a point's K values are read).  Lets a caller pass pcds_coord[:, 0, :, :, 0] of the reference's [B, T, N, 3, 1] tensor
 * (networks/multi_view_encoder.py:395-420 slice it the same way) without a compacting copy. */
int smos_gather_scatter_cl_view(const float* grid, int64_t grid_pitch, const float* gcoord, int32_t Kg, int64_t g_batch_stride,
                                const float* gscale, const float* scoord, int32_t Ks, int64_t s_batch_stride, const float* sscale,
                                float* out, int64_t out_pitch, float* pts_out, int64_t po_b, int64_t po_n, int64_t B, int64_t C,
                                int64_t Hg, int64_t Wg, int64_t N, int64_t Ho, int64_t Wo, const int32_t* n_live,
                                smos_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SMOS_H_ */
