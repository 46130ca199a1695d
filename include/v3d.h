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

/* ------------------------------------------------------------------------------------------
 * Rows A5-A6: CostRegNet (dense 3D-conv U-Net, BatchNorm folded) + soft-argmin depth.
 * Replaces mvsnet.py:133-163 (CostRegNet.forward) and mvsnet.py:219-227.
 *
 * v3d_costreg_pack takes HOST pointers to the reference state_dict tensors
 * (mvsnet.py:136-152; conv{0..6}.conv.weight [Co,Ci,3,3,3], conv{7,8,9}.deconv.weight
 * [Ci,Co,3,3,3], *.bn.{weight,bias,running_mean,running_var}, prob.weight [1,base,3,3,3],
 * prob.bias [1]), folds eval-mode BatchNorm (eps) into the convolutions, re-orders the weights
 * into MFMA fragment order and uploads them.  Order of the 10 conv/deconv layers in the arrays:
 * conv0..conv9.
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
                          float* reg, void* workspace, size_t workspace_bytes, void* stream);

/* Single dense 3D layer of the regulariser (exposed for per-layer parity tests):
 * layer 0..9 = conv0..conv9 of CostRegNet incl. folded BN + ReLU (+ `skip` added after the ReLU
 * when non-NULL, mvsnet.py:159-161).  in [n, Cin, Di, Hi, Wi] -> out [n, Cout, Do, Ho, Wo]. */
int v3d_costreg_layer_f32(const v3d_costreg_weights* handle, int layer, const float* in,
                          const float* skip, int n, int Di, int Hi, int Wi, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* V3D_H_ */
