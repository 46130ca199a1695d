/*
 * v3d.h -- C ABI of lib3dvnet_hip.so: MI355X (gfx950) kernels for 3DVNet's plane-sweep
 * cost-volume and volumetric-refinement hot path.
 *
 * The reference (alexrich021/3dvnet) has no FFI layer: the path sits behind PyTorch nn.Module
 * methods whose arithmetic lives in torch / torch_scatter / MinkowskiEngine CUDA kernels.  Each
 * entry point below replaces one such fused region; the reference call sites it replaces are
 * cited per function (paths relative to the reference root).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()) unless the name ends in `_host`;
 *     tensors are contiguous; float = IEEE f32; indices int32 unless noted;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     calls only enqueue work on it and never synchronise;
 *   - the library never allocates or frees tensor memory: outputs and scratch are allocated by
 *     the caller, `*_workspace_bytes()` reports scratch sizes.  The only library-owned device
 *     memory is inside weight handles (`v3d_*_pack` / `v3d_*_free`);
 *   - return value: 0 = V3D_OK, negative = error; `v3d_last_error()` returns a thread-local
 *     message.  No C++ exception crosses the boundary.
 */
#ifndef V3D_H_
#define V3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V3D_OK 0
#define V3D_ERR_BAD_SHAPE (-1)
#define V3D_ERR_BAD_ARG (-2)
#define V3D_ERR_WORKSPACE_TOO_SMALL (-3)
#define V3D_ERR_HIP (-4)
#define V3D_ERR_UNSUPPORTED (-5)

/* Arithmetic of the matrix-core kernels (every entry point that takes `precision`).  Storage and accumulation are
 * fp32 in both modes; the modes differ in the MFMA operands:
 *   V3D_PRECISION_SPLIT_BF16  each fp32 operand x = hi + lo (hi = RNE_bf16(x), lo = RNE_bf16(x - hi), 16 mantissa bits);
 *                             product = hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 (lo*lo dropped).  Narrower than
 *                             the reference's fp32; passes the 1e-4 relative depth gate of BASELINE.json (the default of
 *                             the Python modules and of bench.py's headline value).
 *   V3D_PRECISION_FP32        exact fp32 products on v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain) -- the reference's
 *                             arithmetic type; about 2x slower on the regulariser. */
#define V3D_PRECISION_SPLIT_BF16 0
#define V3D_PRECISION_FP32 1

/* Developer options: process-wide integers the launch paths read (they replace the environment variables of earlier rounds; a
 * production caller never needs them -- every default is the shipped path).  Names / values:
 *   "psv_kernel"      0 auto (window kernel; reuse kernel for feature stacks >= 2 GB) | 1 reuse kernel | 2 gather kernel
 *   "psv_threads"     64 | 256   workgroup size of the gather kernel
 *   "c12_march"       1 conv1 + conv2 of CostRegNet as one depth march | 0 the two tile kernels (another summation order of conv2)
 *   "c12_nseg"        0 auto | z segments per tile of that march
 *   "c9_kernel"       0 tile kernel | 1 depth-march experiment (libraries built with -DV3D_EXPERIMENTS only) | 2 exact-fp32 unfused
 *   "conv_vec"        1 | 0      float4 staging in the exact-fp32 per-layer kernel
 *   "stop_after"      layer after which v3d_costreg_depth_* returns (-DV3D_PHASE_TIMING builds)
 *   "gemm_rounds"     1 | 0 | 2  gather-GEMM in rounds for small M | the one-step kernel | rounds for every M (bit-identical)
 *   "gemm_round_rows" 0 auto | 32 | 64 | 128
 *   "gemm_pipe"       1 | 2 | 0  sparse convolutions on the loader / matrix pipeline kernel (2: its first version) | the rounds kernel;
 *                                bit-identical to each other and to the one-step kernel
 *   "prop_fused"      1 | 0      PropagationNet as one row-marching kernel | the per-layer kernels (encode + 4 conv + finish)
 *   "tail_streams"    1 .. 8     sub-batches of views in which the regulariser's layers behind conv0 run on concurrent side streams
 *                                (forked from / joined into the caller's stream by events; same kernels, bit-identical results)
 *   "tail_from" / "tail_to"      first / last step of that concurrent section: 1 conv1 + conv2, 3 .. 8 conv3 .. conv8, 9 conv9 + prob,
 *                                10 soft-argmin
 * Unknown names -> V3D_ERR_BAD_ARG; an option this build cannot honour -> V3D_ERR_UNSUPPORTED. */
int v3d_set_option(const char* name, int value);
int v3d_get_option(const char* name, int* value);

/* ABI version (bumped on any signature change) and last error text of the calling thread. */
int v3d_version(void);
const char* v3d_last_error(void);

/* Diagnostics (no reference counterpart): when enabled, every kernel launched by the library is
 * bracketed by hipEvents on its stream; v3d_timing_collect synchronises them, sums the elapsed time
 * per kernel name and clears the log.  names: max_entries x name_stride chars (HOST memory). */
int v3d_timing_enable(int on);
int v3d_timing_collect(int max_entries, char* names_host, int name_stride, float* total_ms_host,
                       int* launches_host);

/* ------------------------------------------------------------------------------------------
 * Row A7 bookkeeping: edge list -> per-reference CSR on the device, no host round trip.
 * Replaces mvsnet.py:179 (ref_idx, gather_idx = torch.unique(ref_src_edges[0], return_inverse=True), whose output length
 * the host must read back) and the grouping scatter(.., gather_idx) implies at mvsnet.py:214-215.
 *
 *   edges     [2, n_edges] int64 (row 0 = reference image, row 1 = source image), any order
 *   n_ref     the number of distinct reference images, known to the caller (the batch holds that many depth maps)
 *   ref_img   [n_ref]      out: ascending distinct values of edges[0] (= torch.unique's order)
 *   edge_ofs  [n_ref+1]    out: first edge of every reference in edge_src
 *   edge_src  [n_edges]    out: edges[1] grouped per reference, original edge order inside a group
 *   workspace >= v3d_edges_csr_workspace_bytes(n_img, n_ref)
 * If n_ref is not the number of distinct references, or an index is outside [0, n_img), the kernel writes an empty CSR
 * (all n_ref + 1 offsets 0 AND all n_ref entries of ref_img 0: the consumers read ref_img[r] and that image's camera block
 * even for a reference without edges, so both tables keep them inside their buffers; the result is a zero variance
 * volume) and sets the workspace's error word; v3d_edges_csr_status copies it to the host (synchronises) ->
 * V3D_ERR_BAD_SHAPE.  Nothing downstream reads that word: callers check it once per edge list (Python: EdgeCsr.check(),
 * MVSNet.check_edges(); CostVolumeGraph checks at capture and on update(ref_src_edges=)).
 * ------------------------------------------------------------------------------------------ */
size_t v3d_edges_csr_workspace_bytes(int n_img, int n_ref);
int v3d_edges_csr(const int64_t* edges, int n_edges, int n_img, int n_ref, int32_t* ref_img, int32_t* edge_ofs,
                  int32_t* edge_src, void* workspace, size_t workspace_bytes, void* stream);
int v3d_edges_csr_status(const void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Rows A1-A4: plane-sweep warp + cross-view variance, one fused kernel.
 * Replaces mv3d/utils.py:86-108 (batched_build_plane_sweep_volume_tensor),
 * mv3d/subnetworks/mvsnet.py:192-206 (projection, |z|+1e-8, normalisation),
 * mvsnet.py:209-211 (F.grid_sample bilinear/zeros/align_corners=True) and
 * mvsnet.py:214-216 (two torch_scatter means -> variance).
 *
 *   feat      [n_img, C, Hf, Wf]  quarter-resolution features (C in {16, 32})
 *   K, R, t   [n_img,3,3], [n_img,3,3], [n_img,3]   intrinsics at image size, world->camera
 *   ref_img   [n_ref]      image index of each reference view (ascending = torch.unique order)
 *   edge_ofs  [n_ref+1]    CSR offsets into edge_src (edges grouped per reference, original order)
 *   edge_src  [n_edges]    source image index per edge
 *   H, W      image size the intrinsics refer to; h, w = plane grid; D planes at
 *             depth_start + i*depth_interval (float32 values of numpy.linspace, utils.py:94)
 *   var       [n_ref, C, D, h, w] out
 *   workspace >= v3d_psv_workspace_bytes(n_img, C, Hf, Wf) bytes, 256-byte aligned
 * ------------------------------------------------------------------------------------------ */
size_t v3d_psv_workspace_bytes(int n_img, int C, int Hf, int Wf);
int v3d_psv_variance_f32(const float* feat, const float* K, const float* R, const float* t,
                         const int32_t* ref_img, const int32_t* edge_ofs, const int32_t* edge_src,
                         int n_img, int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                         double depth_start, double depth_interval, int D, int h, int w,
                         float* var, void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostic twin of the projection inside v3d_psv_variance_* (rows A1-A2 + grid_sample's un-normalisation): the sample
 * position (ix, iy), in feature-map pixels, of every (edge, plane, pixel) -- computed by the same device functions the warp
 * kernels use, so tests can compare the coordinates with the reference's bit for bit.
 *   pos   [n_edges, D*h*w, 2] out;  world [n_ref, 3, D*h*w] out, optional (NULL to skip): the plane-sweep points (row A1)
 *   workspace >= n_img * 36 floats */
int v3d_psv_sample_positions_f32(const float* K, const float* R, const float* t, const int32_t* ref_img,
                                 const int32_t* edge_ofs, const int32_t* edge_src, int n_img, int n_ref, int n_edges,
                                 int Hf, int Wf, int H, int W, double depth_start, double depth_interval, int D, int h,
                                 int w, float* pos, float* world, void* workspace, size_t workspace_bytes, void* stream);

/* Same computation, but `var_split` receives the volume in the regulariser's private input format (no
 * reference counterpart; it only exists to keep the 1.2 GB volume from being re-formatted by the next kernel):
 * every fp32 value x stored as bf16 hi = RNE(x) and bf16 lo = RNE(x - hi), channel-last in 16-byte slots of
 * 8 channels, [n_ref][4 channel groups][hi, lo][D][h][w][8].  n_ref*C*D*h*w*4 bytes like `var`; C must be 32.
 * hi + lo reproduces what conv0's own on-the-fly split of `var` produces, bit for bit. */
int v3d_psv_variance_split(const float* feat, const float* K, const float* R, const float* t,
                         const int32_t* ref_img, const int32_t* edge_ofs, const int32_t* edge_src,
                         int n_img, int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                         double depth_start, double depth_interval, int D, int h, int w,
                         void* var_split, void* workspace, size_t workspace_bytes, void* stream);

/* The same volume as fp32 in the channel-last layout of the exact-fp32 chain's conv0 (V3D_PRECISION_FP32 through
 * v3d_costreg_depth_cl8): [n_ref][4 channel groups][2 halves][D][h][w] 16-byte slots of 4 floats -- half 0 = channels 0..3
 * of the group, half 1 = channels 4..7.  The numbers are those of v3d_psv_variance_f32, bit for bit; C must be 32. */
int v3d_psv_variance_cl8(const float* feat, const float* K, const float* R, const float* t,
                         const int32_t* ref_img, const int32_t* edge_ofs, const int32_t* edge_src,
                         int n_img, int n_ref, int n_edges, int C, int Hf, int Wf, int H, int W,
                         double depth_start, double depth_interval, int D, int h, int w,
                         float* var_cl8, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Rows A5-A6: CostRegNet (dense 3D-conv U-Net, BatchNorm folded) + soft-argmin depth.
 * Replaces mvsnet.py:133-163 (CostRegNet.forward) and mvsnet.py:219-227.
 *
 * v3d_costreg_pack takes HOST pointers to the reference state_dict tensors
 * (mvsnet.py:136-152; conv{0..6}.conv.weight [Co,Ci,3,3,3], conv{7,8,9}.deconv.weight
 * [Ci,Co,3,3,3], *.bn.{weight,bias,running_mean,running_var}, prob.weight [1,base,3,3,3],
 * prob.bias [1]), folds eval-mode BatchNorm (eps) into the convolutions, re-orders the weights
 * into MFMA fragment order and uploads them.  Order of the 10 conv/deconv layers in the arrays:
 * conv0..conv9.  Built: (in_channels, base_channels) = (32, 8) -- mv3d/config.py:42 -- and (16, 8), the
 * reference's signature default feat_dim (lightningmodel.py:18); a 16-channel net takes its variance
 * volume through v3d_costreg_depth_f32 only (the split / channel-last hand-off formats are defined
 * for 32 channels).
 * ------------------------------------------------------------------------------------------ */
typedef struct v3d_costreg_weights v3d_costreg_weights;

int v3d_costreg_pack(const float* const* conv_weight_host, const float* const* bn_weight_host,
                     const float* const* bn_bias_host, const float* const* bn_mean_host,
                     const float* const* bn_var_host, const float* prob_weight_host,
                     const float* prob_bias_host, int in_channels, int base_channels, float bn_eps,
                     v3d_costreg_weights** out_handle);
void v3d_costreg_free(v3d_costreg_weights* handle);

/*   var        [n_ref, Cin, D, h, w] in   (D, h, w divisible by 8)
 *   depth_vals [D]  depth hypothesis values (torch.linspace(depth_start, depth_end, D), :223)
 *   depth      [n_ref, h, w] out
 *   reg        [n_ref, D, h, w] out, optional (NULL to skip): the regularised volume x_reg
 *   workspace  >= v3d_costreg_workspace_bytes(...) */
size_t v3d_costreg_workspace_bytes(const v3d_costreg_weights* handle, int n_ref, int D, int h, int w);
int v3d_costreg_depth_f32(const v3d_costreg_weights* handle, const float* var,
                          const float* depth_vals, int n_ref, int D, int h, int w, float* depth,
                          float* reg, int precision, void* workspace, size_t workspace_bytes, void* stream);
/* As above with the variance volume in the split format written by v3d_psv_variance_split (the split format IS the
 * V3D_PRECISION_SPLIT_BF16 operand encoding, so this entry point has no precision argument). */
int v3d_costreg_depth_split(const v3d_costreg_weights* handle, const void* var_split,
                          const float* depth_vals, int n_ref, int D, int h, int w, float* depth,
                          float* reg, void* workspace, size_t workspace_bytes, void* stream);
/* As v3d_costreg_depth_f32 with V3D_PRECISION_FP32 (exact fp32 products) and the volume in the fp32 channel-last layout
 * of v3d_psv_variance_cl8: conv0 then runs as a depth march that streams the volume with LDS-DMA (csrc/conv0z.hip). */
int v3d_costreg_depth_cl8(const v3d_costreg_weights* handle, const void* var_cl8,
                          const float* depth_vals, int n_ref, int D, int h, int w, float* depth,
                          float* reg, void* workspace, size_t workspace_bytes, void* stream);

/* Single dense 3D layer of the regulariser (exposed for per-layer parity tests):
 * layer 0..9 = conv0..conv9 of CostRegNet incl. folded BN + ReLU (+ `skip` added after the ReLU
 * when non-NULL, mvsnet.py:159-161).  in [n, Cin, Di, Hi, Wi] -> out [n, Cout, Do, Ho, Wo].
 * precision: V3D_PRECISION_FP32 = the exact-fp32 kernel of every layer; V3D_PRECISION_SPLIT_BF16 = conv0 on its
 * split-bf16 product kernel (the other layers' split-bf16 kernels use the split activation layout: next entry point). */
int v3d_costreg_layer_f32(const v3d_costreg_weights* handle, int layer, const float* in,
                          const float* skip, int n, int Di, int Hi, int Wi, float* out, int precision,
                          void* stream);

/* The same layers 1..8 (conv1..conv8) on the kernels the fused path uses: split-bf16 matrix cores reading the split
 * channel-last activation layout.  `in`, `skip` (conv7 / conv8, required) and `out` are fp32 [n, C, D, H, W]; the
 * input is re-encoded into `workspace` (>= v3d_costreg_layer_split_workspace_bytes).  Layer 0 already runs its
 * product kernel through v3d_costreg_layer_f32; conv9 only exists fused with the prob conv (v3d_costreg_depth_*). */
size_t v3d_costreg_layer_split_workspace_bytes(int n, int cin, int Di, int Hi, int Wi);
int v3d_costreg_layer_split_f32(const v3d_costreg_weights* handle, int layer, const float* in, const float* skip,
                                int n, int Di, int Hi, int Wi, float* out, void* workspace, size_t workspace_bytes,
                                void* stream);

/* ------------------------------------------------------------------------------------------
 * Rows B1-B2 and C1: back-project depth pixels / depth hypotheses to world points and compute their
 * multi-view variance feature.  Replaces mv3d/utils.py:67-83 (build_img_pts),
 * mv3d/lightningmodel.py:138-169 (construct_feature_rich_pointcloud) and :191-229 (run_pointflow).
 *   depth [n_ref, h, w]; feat [n_img, C, Hf, Wf]; cameras / edge CSR as in v3d_psv_variance_f32;
 *   hypotheses depth + i*offset, i in [-n_half, n_half] (n_half = 0: the depth itself);
 *   pts [n_ref*h*w, 2*n_half+1, 3] out, var [n_ref*h*w, 2*n_half+1, C] out (= pts_hyp / pts_feat of
 *   lightningmodel.py:231-235; for n_half = 0: pts / pts_feat of :171-172).
 *   The call keeps a channel-last copy of `feat` in `workspace`; feat == NULL means "the workspace still holds the copy the
 *   previous call made of the same feature tensor" (a driver that sweeps one scene's features repeatedly skips the copy).
 * ------------------------------------------------------------------------------------------ */
size_t v3d_backproject_workspace_bytes(int n_img, int C, int Hf, int Wf);
int v3d_backproject_variance_f32(const float* depth, const float* feat, const float* K, const float* R,
                                 const float* t, const int32_t* ref_img, const int32_t* edge_ofs,
                                 const int32_t* edge_src, int n_img, int n_ref, int n_edges, int C,
                                 int Hf, int Wf, int H, int W, int h, int w, double offset, int n_half,
                                 float* pts, float* var, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Gather-GEMM on the matrix cores (fp32 in/out; operands per `precision`): Y[m,:] = epilogue(sum_s act(X_s[row_s(m), 0:K]) @ W_s + bias).
 * Replaces the dense arithmetic of: PointNet's Linear layers incl. the concat with the pooled voxel
 * feature and torch_scatter max (mv3d/subnetworks/scenemodeling.py:127-144); MinkowskiConvolution /
 * ConvolutionTranspose / 1x1 + MinkowskiGroupNorm + ReLU + residual (scenemodeling.py:16-44,160,181,
 * 186); Conv1d+BN+ReLU of the hypothesis decoder (mv3d/subnetworks/refinement.py:8-13,20-23).
 *
 * v3d_gemm_pack: HOST weight tensor addressed as w[seg*stride_seg + co*stride_co + k*stride_k]
 *   (Linear [N, n_seg*K]: K, n_seg*K, 1;  ME kernel [27, Ci, Co]: Ci*Co, 1, Co;  Conv1d [Co, Ci, 3]:
 *   1, 3*Ci, 3), optional per-output scale (folded BatchNorm), bias, GroupNorm affine.  N <= 128.
 * v3d_gemm_gather_f32: seg_src/seg_idx/seg_ld are HOST arrays of n_seg device pointers / strides;
 *   seg_idx[s] NULL = identity row map, entry -1 = zero row; group_len > 0 selects the conv1d row map
 *   (segment s reads row m + s - n_seg/2 inside each group of group_len rows; seg_idx ignored);
 *   relu_in: ReLU applied to gathered inputs; use_gn: per-row GroupNorm (1 or 16: over 16-channel groups, 8: over 8-channel
 *   groups; eps gn_eps) before the optional residual add and ReLU; pool/pool_idx: scatter-max of the result
 *   into pool[pool_idx[m], :] (pre-filled with -inf); out may be NULL.
 * ------------------------------------------------------------------------------------------ */
typedef struct v3d_gemm_weights v3d_gemm_weights;
int v3d_gemm_pack(const float* w_host, long long stride_seg, long long stride_co, long long stride_k,
                  int n_seg, int N, int K, const float* scale_host, const float* bias_host,
                  const float* gn_w_host, const float* gn_b_host, v3d_gemm_weights** out_handle);
void v3d_gemm_free(v3d_gemm_weights* handle);
int v3d_gemm_gather_f32(const v3d_gemm_weights* handle, int M, const float* const* seg_src_host,
                        const int32_t* const* seg_idx_host, const int* seg_ld_host, int group_len,
                        int relu_in, int use_gn, float gn_eps, const float* residual, int ld_res,
                        int relu_out, float* pool, const int32_t* pool_idx, int ld_pool, float* out,
                        int ld_out, int precision, void* stream);
/* A sparse convolution as one call (MinkowskiConvolution / ConvolutionTranspose + MinkowskiGroupNorm + residual + ReLU,
 * scenemodeling.py:16-44,160,181): v3d_gemm_gather_f32 with every segment reading `src` [*, ld_src] through column k of the
 * neighbour table `nbr` ([n_seg, nbr_stride] int32 from v3d_sparse_neighbors, -1 = absent voxel).  Same kernels, same bits;
 * the caller passes 4 pointers instead of three host arrays of n_seg entries. */
int v3d_sparse_conv_f32(const v3d_gemm_weights* handle, int M, const float* src, int ld_src, const int32_t* nbr,
                        long long nbr_stride, int use_gn, float gn_eps, const float* residual, int ld_res, int relu_out,
                        float* out, int ld_out, int precision, void* stream);
int v3d_fill_f32(float* ptr, size_t n, float value, void* stream);
/* PointNet input of PL3DVNet.model_scene (mv3d/lightningmodel.py:180-183): out[i] = [pts[edge_pt[i]] - anchor_pts[edge_anchor[i]]
 * | pts_feat[edge_pt[i]]], out [n_edges, 3 + C]; pts [*, 3], anchor_pts [*, 3], pts_feat [*, C], edges int64. */
int v3d_pointnet_input_f32(const float* pts, const float* anchor_pts, const float* pts_feat, const int64_t* edge_anchor,
                           const int64_t* edge_pt, int n_edges, int C, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sparse-tensor structure (replaces MinkowskiEngine's coordinate manager; semantics: SURVEY.md
 * Appendix A).  coords are int32 [N,4] = (batch, x, y, z), unique rows.
 *   v3d_hash_build        open-addressing table over the coordinate map (buffer >= v3d_hash_bytes(n))
 *   v3d_sparse_neighbors  nbr[k][p] = row of out_coords[p] + step*o_k (or -1), k = (ox+1)+3(oy+1)+9(oz+1);
 *                         conv: step = +tensor_stride_in, transposed conv: step = -tensor_stride_out
 *   v3d_sparse_interp_f32 MinkowskiInterpolation (refinement.py:26,39): trilinear interpolation of
 *                         feats [N,C] at pts [n_pts, n_hyp, 3] (world), query coordinate
 *                         ((p - min_pts[batch]) / res) * tensor_stride, missing corners add 0; result
 *                         written to out[q*ld_out + col0 .. +C); workspace >= v3d_sparse_interp_workspace_bytes.
 * ------------------------------------------------------------------------------------------ */
size_t v3d_hash_bytes(int n);
int v3d_hash_build(const int32_t* coords, int n, void* table, size_t table_bytes, void* stream);
/* Keys pack 16 bits per field: batch in [0, 65535], x/y/z in [-8, 65519].  A row outside that range is left out and
 * sets the table's error word; v3d_hash_status copies it to the host (synchronises) -> V3D_ERR_BAD_SHAPE. */
int v3d_hash_status(const void* table, int n, void* stream);
int v3d_sparse_neighbors(const void* table, int n_in, const int32_t* out_coords, int n_out, int step,
                         int32_t* nbr, void* stream);
size_t v3d_sparse_interp_workspace_bytes(int n_pts, int n_hyp);
int v3d_sparse_interp_f32(const void* table, int n_in, const float* feats, int C, int tensor_stride,
                          const float* pts, const int64_t* pts_batch, int n_pts, int n_hyp,
                          const float* min_pts, float res, float* out, int ld_out, int col0,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row B3 (mv3d/utils.py:38-64 incl. the un-vendored torch_geometric voxel_grid / torch_cluster grid,
 * restated) and the stride-2 coordinate maps of row B6.  Output sizes are data dependent, so
 * v3d_sort_unique_u64 returns the count to the HOST and synchronises the stream (as torch.unique does
 * in the reference, utils.py:48); everything else is asynchronous.
 *   v3d_voxel_keys       bounding box of pts [n,3] -> ceil-based grid size + trunc+1 cell counts -> 1-D
 *                        voxel id per point (utils.py:39-45); metadata stays in `workspace`
 *   v3d_sort_unique_u64  ascending unique keys (torch.unique, :48)
 *   v3d_lower_bound_u64  position of each point's key in the unique list = inverse index (:48-49)
 *   v3d_voxel_decode     anchor batch, 3-D voxel index, voxel centre, per-batch shift to min 0 (:50-62);
 *                        `workspace` must be the buffer v3d_voxel_keys filled; half_edge = edge_len / 2
 *   v3d_strided_keys / v3d_unpack_coords   packed keys of floor(c / 2ts) * 2ts and back to int32 [n,4]
 * ------------------------------------------------------------------------------------------ */
/* Row B4's max-pooling over the points of a voxel (scenemodeling.py:129-141, torch_scatter.scatter(reduce='max')) without
 * atomics: v3d_segment_csr groups the rows by segment id once (perm [n]: row indices sorted by id, stable; offsets [n_seg+1]),
 * v3d_segment_max_f32 reduces every segment's rows of src [n, ld] to out [n_seg, ld_out] (first N columns; an empty segment
 * yields -inf, the initial value of the atomic version's pool).  Ids must lie in [0, n_seg). */
size_t v3d_segment_csr_workspace_bytes(int n);
int v3d_segment_csr(const int32_t* seg_id, int n, int n_seg, int32_t* perm, int32_t* offsets, void* workspace,
                    size_t workspace_bytes, void* stream);
int v3d_segment_max_f32(const float* src, int ld, const int32_t* perm, const int32_t* offsets, int n_seg, int N, float* out,
                        int ld_out, void* stream);
size_t v3d_sort_unique_workspace_bytes(int n);
int v3d_sort_unique_u64(const uint64_t* keys_in, int n, uint64_t* keys_out, int* n_unique_host,
                        void* workspace, size_t workspace_bytes, void* stream);
int v3d_strided_keys(const int32_t* coords, int n, int tensor_stride, uint64_t* keys_out, void* stream);
int v3d_unpack_coords(const uint64_t* keys, int n, int32_t* coords_out, void* stream);
size_t v3d_voxelize_workspace_bytes(void);
int v3d_voxel_keys(const float* pts, const int64_t* pts_batch, int n, float edge_len, uint64_t* keys_out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* Range checks of v3d_voxel_keys (batch ids in [0, 1024), at most 65000 cells per axis, no NaN): the error word lives in
 * `workspace`; this call copies it to the host (synchronises) and returns V3D_ERR_BAD_SHAPE if it is set, in which case
 * v3d_voxel_decode writes nothing. */
int v3d_voxelize_status(const void* workspace, size_t workspace_bytes, void* stream);
int v3d_lower_bound_u64(const uint64_t* sorted_unique, int n_unique, const uint64_t* queries, int n,
                        int64_t* index_out, void* stream);
int v3d_voxel_decode(const uint64_t* unique_keys, int n_unique, float edge_len, float half_edge,
                     float* anchor_pts, int32_t* anchor_idx3d, int64_t* anchor_batch, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Row C2b tail + C3: last Conv1d(C -> 1, k3, pad 1, bias) over the hypothesis axis, softmax
 * (refinement.py:24,43) and optional expectation sum_i p_i*offset_vals_i (lightningmodel.py:238-241).
 * act [n_pts, n_hyp, C]; weight [1, C, 3]; preds [n_pts, n_hyp]; expect [n_pts] or NULL. */
int v3d_decoder_head_f32(const float* act, int n_pts, int n_hyp, int C, const float* weight,
                         const float* bias, const float* offset_vals, float* preds, float* expect,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY.md 8f rank 2 -- PropagationNet, the learned 3x3 depth propagation of stage 3
 * (mv3d/subnetworks/upsampling.py:14-36; called at 1/4, 1/2 and full resolution by mv3d/eval-3dvnet.py:101-125 and
 * mv3d/lightningmodel.py:85,98,111).  Replaces PropagationNet.forward(features, depth):
 *   x = cat(features, depth); four Conv2d(3x3, pad 1, no bias) + BatchNorm2d (eval, folded) + ReLU (in -> 32 -> 32 -> 32 -> 9);
 *   p = softmax over the 9 logits; out = sum_k p_k * unfold(replicate_pad(depth))_k.
 * v3d_propagation_pack takes HOST pointers to conv{1..4}.0.weight [Co, Ci, 3, 3] and conv{1..4}.1.{weight, bias,
 * running_mean, running_var}; in_dim = guide channels + 1 (33 for the feature-guided nets, 4 for the image-guided one).
 *   features [B, in_dim - 1, H, W], depth [B, 1, H, W] (= [B, H, W]), out [B, H, W]; `precision` (ABI version 5) =
 *   V3D_PRECISION_SPLIT_BF16 | V3D_PRECISION_FP32 (exact fp32 products on v_mfma_f32_16x16x4_f32, the reference's arithmetic).
 * ------------------------------------------------------------------------------------------ */
typedef struct v3d_propagation_weights v3d_propagation_weights;
int v3d_propagation_pack(const float* const* conv_weight_host, const float* const* bn_weight_host,
                         const float* const* bn_bias_host, const float* const* bn_mean_host,
                         const float* const* bn_var_host, int in_dim, int h_dim, float bn_eps,
                         v3d_propagation_weights** out_handle);
void v3d_propagation_free(v3d_propagation_weights* handle);
size_t v3d_propagation_workspace_bytes(const v3d_propagation_weights* handle, int B, int H, int W);
int v3d_propagation_f32(const v3d_propagation_weights* handle, const float* features, const float* depth, int B, int Cf,
                        int H, int W, float* out, int precision, void* workspace, size_t workspace_bytes, void* stream);
/* The same network with the nearest-neighbour resize that precedes every call of stage 3 (F.interpolate(depth, size, 'nearest'),
 * mv3d/eval-3dvnet.py:103,111,119) folded into the kernel's addressing (ABI version 5): depth_lo [B, h0, w0] is the depth BEFORE
 * the resize, iy [H] / ix [W] (DEVICE int32) the source row / column of every output row / column -- the caller obtains them
 * from the host framework's own nearest rule; both NULL with h0 == H, w0 == W = no resize.  One row-marching kernel (csrc/propz.hip):
 * the four layers' activations stay in LDS, no workspace.  v3d_propagation_f32 runs the same kernel (developer option
 * "prop_fused" = 0: the per-layer kernels of round 4, other summation orders). */
int v3d_propagation_up_f32(const v3d_propagation_weights* handle, const float* features, const float* depth_lo, int B, int Cf,
                           int H, int W, int h0, int w0, const int32_t* iy, const int32_t* ix, float* out, int precision, void* stream);

/* Rows C2a + C2b + C3 fused: MinkowskiInterpolation of the three U-Net levels at the hypothesis
 * points (mv3d/subnetworks/refinement.py:28-41) -> the three Conv1d+BN+ReLU layers along the hypothesis axis (:16-23) ->
 * Conv1d(128 -> 1) + softmax (:24,43) -> expected offset (mv3d/lightningmodel.py:237-241): one index kernel (the 8 lattice
 * corners of every hypothesis point on the three levels -> (row, weight) pairs in `workspace`) and ONE matrix kernel; the
 * [n_pts, C, n_hyp] feature tensor and the intermediate activations never reach HBM.  Split-bf16 MFMA operands
 * (V3D_PRECISION_SPLIT_BF16; for exact fp32 use v3d_sparse_interp_f32 + v3d_gemm_gather_f32 + v3d_decoder_head_f32).
 *   layers_host   3 handles from v3d_gemm_pack: Conv1d weights [128, Cin, 3] (strides 1, 3*Cin, 3; n_seg 3) with the folded
 *                 BatchNorm scale / bias; Cin = sum of the level widths + c_feat for the first, 128 for the others
 *   level_*_host  HOST arrays of 3 entries in feature-row order (finest level first, refinement.py:41): hash table
 *                 (v3d_hash_build) and its row count, feats [N, C] (C a multiple of 16, 16-byte aligned, N*C*4 < 2 GB), tensor
 *                 stride, per-batch minimum point [n_batch, 3] (refinement.py:33), x.res
 *   pts [n_pts, n_hyp, 3], pts_batch [n_pts] int64, pts_feat [n_pts, n_hyp, c_feat] or NULL (c_feat 0 or a multiple of 16),
 *   n_hyp <= 8; head_weight [1, 128, 3], head_bias [1]; offset_vals [n_hyp] or NULL; preds [n_pts, n_hyp]; expect [n_pts] or NULL
 *   depth_inout   [n_pts] or NULL: the expected offset is also ADDED to it in place (the driver's `depth += offset`,
 *                 mv3d/eval-3dvnet.py:99: one elementwise launch less per sweep)
 *   workspace     v3d_decoder_fused_workspace_bytes(n_pts, n_hyp) bytes, 16-byte aligned (ABI version 3) */
size_t v3d_decoder_fused_workspace_bytes(int n_pts, int n_hyp);
int v3d_decoder_fused_f32(const v3d_gemm_weights* const* layers_host, const float* head_weight, const float* head_bias,
                          const void* const* level_table_host, const int* level_n_host,
                          const float* const* level_feats_host, const int* level_C_host, const int* level_stride_host,
                          const float* const* level_min_pts_host, const float* level_res_host, const float* pts,
                          const int64_t* pts_batch, const float* pts_feat, int c_feat, int n_pts, int n_hyp,
                          const float* offset_vals, float* preds, float* expect, float* depth_inout, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY.md 8f rank 3 -- the 2D feature extractor of MVSNet (mv3d/subnetworks/mvsnet.py:55-105: torchvision MnasNet-1.0 trunk +
 * FeaturePyramidNetwork): the layer kernels on CHANNELS-LAST fp32 activations [n, H, W, C] (C a multiple of 8), eval-mode
 * BatchNorm folded into weights / bias by the caller (3dvnet_amd/backbone.py drives them with torchvision's key names).
 *   v3d_conv_pack        HOST weight [Cout, K], K = taps * Cin in (tap, channel) order (Conv2d weight permuted to
 *                        [Cout, kh, kw, Cin]), bias [Cout] or NULL -> handle (exact-fp32 MFMA fragments)
 *   v3d_conv_nhwc_f32    1x1 (taps 1) or 3x3 / pad 1 (taps 9) convolution + bias (+ ReLU) (+ residual: res_mode 1 = res
 *                        [n, H, W, Cout]; 2 = nearest-upsampled res [n, ceil(H/2), ceil(W/2), Cout], the FPN's top-down addition; odd H, W allowed)
 *   v3d_depthwise_nhwc_f32  k x k (3 | 5) depthwise, pad k/2, stride 1 | 2, DEVICE w [k*k][C], bias [C] -> [n, ceil(H/s), ceil(W/s), C]
 *   v3d_stem_f32         Conv2d(3 -> 32, k3, s2, p1) + bias + ReLU from the NCHW image; DEVICE w [27][32] ((c, ky, kx) major)
 *   v3d_nhwc_to_nchw_f32 [n, HW, C] -> [n, C, HW] (C a multiple of 32): the layout the cost-volume entry points take
 * ------------------------------------------------------------------------------------------ */
typedef struct v3d_conv_weights v3d_conv_weights;
int v3d_conv_pack(const float* w_host, const float* bias_host, int cout, int k, v3d_conv_weights** out_handle);
void v3d_conv_free(v3d_conv_weights* handle);
int v3d_conv_nhwc_f32(const v3d_conv_weights* handle, const float* x, int n, int H, int W, int cin, int taps, int relu,
                      int res_mode, const float* res, float* out, void* stream);
int v3d_depthwise_nhwc_f32(const float* x, const float* w, const float* bias, int n, int H, int W, int C, int ksize, int stride,
                           int relu, float* out, void* stream);
int v3d_stem_f32(const float* image, const float* w, const float* bias, int n, int H, int W, float* out, void* stream);
int v3d_nhwc_to_nchw_f32(const float* in, float* out, int n, int C, int HW, void* stream);

/* An inverted-residual block of the trunk (torchvision mnasnet._InvertedResidual: 1x1 expand + BN + ReLU -> k x k depthwise,
 * stride s, + BN + ReLU -> 1x1 project + BN, + x when `residual`) as ONE kernel (csrc/irb.hip; ABI version 6, round 6): the
 * expanded tensor never reaches HBM.  Split-bf16 matrix operands, fp32 depthwise taps; the exact-fp32 block is the three calls above.
 *   v3d_irb_pack       HOST weights with BatchNorm folded: w_expand [mid, cin], w_dw [mid, k, k], w_project [cout, mid], biases
 *                      [mid], [mid], [cout]; cin, mid, cout multiples of 8; k 3 | 5; stride 1 | 2; residual needs cin == cout, stride 1
 *   v3d_irb_supported  1 when a kernel instance exists for this block at input size H x W (the MnasNet-1.0 blocks at image sides
 *                      that are multiples of 32 and at 240 x 320), else 0: the caller then runs the three-call path
 *   v3d_irb_nhwc_f32   x [n, H, W, cin] -> out [n, ceil(H / s), ceil(W / s), cout], channels-last fp32; `workspace` of
 *                      v3d_irb_workspace_bytes(handle, n, H, W) bytes (16-byte aligned; 0 bytes / NULL for most maps): a map with fewer
 *                      tiles than the chip has CUs (71 images at 1/32 resolution) shares a tile's expanded channels out over several
 *                      workgroups whose partial sums meet there, summed in a fixed order by a second launch */
typedef struct v3d_irb_weights v3d_irb_weights;
int v3d_irb_pack(const float* w_expand, const float* b_expand, const float* w_dw, const float* b_dw, const float* w_project,
                 const float* b_project, int cin, int mid, int cout, int ksize, int stride, int residual,
                 v3d_irb_weights** out_handle);
void v3d_irb_free(v3d_irb_weights* handle);
int v3d_irb_supported(const v3d_irb_weights* handle, int H, int W);
/* The trunk's first three layers (mnasnet layers 0-7: 3x3 / stride 2 convolution 3 -> 32 + BN + ReLU, 3x3 depthwise + BN + ReLU, 1x1 -> 16
 * + BN) as the same kernel: the 27 (channel, ky, kx) taps of the first convolution are the K dimension of its first matrix product.
 *   v3d_stem_block_pack  HOST w_stem [32, 27] (Conv2d weight [32, 3, 3, 3] flattened), w_dw [32, 3, 3], w_pw [16, 32], BatchNorm folded
 *   v3d_stem_block_f32   image [n, 3, IH, IW] (NCHW, even sides) -> out [n, IH/2, IW/2, 16] channels-last; free with v3d_irb_free */
int v3d_stem_block_pack(const float* w_stem, const float* b_stem, const float* w_dw, const float* b_dw, const float* w_pw,
                        const float* b_pw, v3d_irb_weights** out_handle);
int v3d_stem_block_f32(const v3d_irb_weights* handle, const float* image, int n, int IH, int IW, float* out, void* stream);
size_t v3d_irb_workspace_bytes(const v3d_irb_weights* handle, int n, int H, int W);
int v3d_irb_nhwc_f32(const v3d_irb_weights* handle, const float* x, int n, int H, int W, float* out, void* workspace,
                     size_t workspace_bytes, void* stream);


/* One level of the feature pyramid (torchvision FeaturePyramidNetwork, mvsnet.py:83-105) as ONE kernel (csrc/fpn.hip; ABI version 6):
 * inner = lateral 1x1 (x) + bias + nearest-upsampled inner of the coarser level; out = 3x3 / pad 1 (inner) + bias, written in the
 * reference layout.  feat_dim 32, cin a multiple of 8 and <= 48 (the levels at 1/2, 1/4, 1/8); split-bf16 matrix operands.
 *   v3d_fpn_pack       HOST w_lateral [32, cin], b_lateral [32], w_out [32, 32, 3, 3], b_out [32]
 *   v3d_fpn_level_f32  x [n, H, W, cin] channels-last; coarse_inner [n, ceil(H/2), ceil(W/2), 32] channels-last or NULL;
 *                      inner_out [n, H, W, 32] channels-last or NULL (not needed for the finest level); out [n, 32, H, W] */
typedef struct v3d_fpn_weights v3d_fpn_weights;
int v3d_fpn_pack(const float* w_lateral, const float* b_lateral, const float* w_out, const float* b_out, int cin,
                 v3d_fpn_weights** out_handle);
void v3d_fpn_free(v3d_fpn_weights* handle);
int v3d_fpn_level_f32(const v3d_fpn_weights* handle, const float* x, const float* coarse_inner, int n, int H, int W,
                      float* inner_out, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* V3D_H_ */
