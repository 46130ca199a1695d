// Shared helpers for lib3dvnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "v3d.h"

namespace v3d {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define V3D_CHECK_HIP(expr)                                                                    \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return v3d::fail(V3D_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                       __FILE__, __LINE__);                                                    \
  } while (0)

#define V3D_CHECK_LAUNCH(name)                                                                 \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess)                                                                      \
      return v3d::fail(V3D_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_));   \
  } while (0)

#define V3D_REQUIRE(cond, code, ...)                                                           \
  do {                                                                                         \
    if (!(cond)) return v3d::fail(code, __VA_ARGS__);                                          \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Developer options (v3d_set_option in include/v3d.h): process-wide integers read at launch time.  They replace the environment
// variables earlier rounds read on the launch paths; none of them is needed in production, every default is the shipped path.
enum Option {
  kOptPsvKernel,       // "psv_kernel": 0 auto (window kernel; reuse kernel for feature stacks >= 2 GB), 1 reuse kernel, 2 gather kernel
  kOptPsvThreads,      // "psv_threads": 64 | 256, workgroup size of the gather kernel
  kOptC12March,        // "c12_march": 1 conv1 + conv2 as one depth march (conv12z.hip), 0 the two tile kernels
  kOptC12Nseg,         // "c12_nseg": 0 auto, else z segments per tile of the conv1 + conv2 march
  kOptC9Kernel,        // "c9_kernel": 0 tile kernel, 1 depth-march experiment (builds with -DV3D_EXPERIMENTS only), 2 exact-fp32 unfused
  kOptConvVec,         // "conv_vec": 1 float4 staging of halo rows in the exact-fp32 layer kernel, 0 scalar
  kOptStopAfter,       // "stop_after": regulariser returns after this layer (-DV3D_PHASE_TIMING builds: isolates a kernel's counters)
  kOptGemmRounds,      // "gemm_rounds": 1 gather-GEMM in rounds for small M, 0 the one-step kernel, 2 rounds for every M (measured: PointNet's 200 k-row layers 1.25 -> 1.21 ms per scene, not the default)
  kOptGemmRoundRows,   // "gemm_round_rows": 0 auto, 32 | 64 | 128 rows per tile of the rounds / pipeline kernel
  kOptGemmPipe,        // "gemm_pipe": 1 sparse convolutions on the loader / matrix pipeline kernel (2: its first version), 0 the rounds kernel
  kOptTailStreams,     // "tail_streams": sub-batches of views the regulariser's layers behind conv0 run in, on concurrent side streams (1 = the caller's stream only)
  kOptTailFrom,        // "tail_from": first step of the concurrent section (1 conv1 + conv2, 3 .. 8 conv3 .. conv8, 9 conv9 + prob, 10 soft-argmin)
  kOptTailTo,          // "tail_to": last step of the concurrent section
  kOptPropFused,       // "prop_fused": 1 PropagationNet as one row-marching kernel (propz.hip), 0 the per-layer kernels (encode + 4 conv + finish)
  kOptCount
};
int option(Option o);

// Optional per-kernel timing (v3d_timing_* in include/v3d.h): when enabled every launch made through
// V3D_LAUNCH is bracketed by hipEvents on its own stream.  Off by default (zero overhead).
bool timing_enabled();
void timing_begin(const char* name, hipStream_t s);
void timing_end(hipStream_t s);

// Grid of a persistent kernel: wgs_per_cu workgroups on every CU of the current device, a multiple of 8 (one equal share per
// XCD), never more than the tiles there are.
inline unsigned persistent_grid(long long n_tiles, int wgs_per_cu) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n_cu = v;
  }
  long long g = (long long)n_cu * wgs_per_cu;
  if (g > n_tiles) g = n_tiles;
  g = (g + 7) / 8 * 8;
  return (unsigned)g;
}

struct TimedScope {
  hipStream_t s;
  bool on;
  TimedScope(const char* name, hipStream_t s_) : s(s_), on(timing_enabled()) {
    if (on) timing_begin(name, s);
  }
  ~TimedScope() {
    if (on) timing_end(s);
  }
};

// XCD-aware workgroup order.  The dispatcher deals workgroups round-robin to the 8 XCDs (id & 7), each with a
// private 4 MB L2.  This bijection of [0, gridDim.x) hands every XCD one contiguous run of the logical tile order
// instead, so neighbouring tiles (shared halos, the same reference view's source maps) meet in the same L2.
__device__ __forceinline__ int xcd_contiguous_block() {
  const int total = (int)gridDim.x, per = total >> 3, rem = total & 7;
  const int x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
  return x * per + (x < rem ? x : rem) + i;
}

// Division of a (usually wave-uniform) index by a launch constant.  hipcc expands x / d for a run-time d into a float
// reciprocal sequence on the VECTOR unit (v_cvt, v_rcp_iflag_f32, v_mul_hi_u32, fix-ups: ~12 instructions, several at quarter
// rate) even when x is in SGPRs; with m = magic_u32(x_max, d) from the host it is one s_mul_hi_u32.
//   q = mulhi(x, ceil(2^32 / d)) is exact while x * (m * d - 2^32) < 2^32, i.e. for all x <= x_max when x_max * d < 2^32
//   (m * d - 2^32 < d); magic_u32 returns 0 = "divide" when that does not hold (or d == 1: 2^32 does not fit).
inline unsigned magic_u32(unsigned long long x_max, unsigned d) {
  if (d <= 1 || x_max * d >= (1ull << 32)) return 0u;
  return (unsigned)(((1ull << 32) + d - 1) / d);
}
__device__ __forceinline__ unsigned udiv_magic(unsigned x, unsigned d, unsigned m) { return m ? __umulhi(x, m) : x / d; }

// x / D for a compile-time D >= 2 and a small x (x * D < 2^20): one full-rate 24-bit multiply and a shift -- for such a
// division hipcc emits v_mul_hi_u32, a quarter-rate instruction
template <int D>
__device__ __forceinline__ unsigned small_div(unsigned x) {
  static_assert(D >= 2 && D < 1024, "small_div: divisor range");
  return __umul24(x, (unsigned)((1 << 20) / D + 1)) >> 20;
}

// Persistent kernels: the grid is the number of workgroups resident at once (a multiple of 8, see persistent_grid()).
// Workgroup (XCD x = id & 7, slot i = id >> 3) walks tiles first_x + i, first_x + i + G, ... of its XCD's contiguous run of the
// logical tile order (G = workgroups per XCD), so at any moment an XCD's workgroups sit on G consecutive tiles.
struct TileWalk { int t, end, step; };
__device__ __forceinline__ TileWalk xcd_tile_walk(int n_tiles) {
  const int G = (int)gridDim.x >> 3;
  const int x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
  const int per = n_tiles >> 3, rem = n_tiles & 7;
  const int first = x * per + (x < rem ? x : rem);
  return {first + i, first + per + (x < rem ? 1 : 0), G};
}

// Single fp32 operations the compiler must not fuse with their neighbours.  HIP's __fmul_rn / __fadd_rn are plain operators
// (hipcc's default -ffp-contract=fast-honor-pragmas contracts them into FMAs like any other a * b + c), so the rounding
// points of the reference's arithmetic are pinned with a contraction-free scope instead.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// Pinned evaluation orders of the camera arithmetic (scripts/coord_order_probe.py): torch.bmm evaluates the large batched
// products K^-1 p, R^T c and P [X;1] as FMA chains in k order whose first term is a plain product; hipcc's default
// -ffp-contract=fast would pick its own mul/add pairs to fuse (it did: one row of P [X;1] came out as
// fma(a0,b0, rnd(a1 b1)) + rnd(a2 b2)), so the chains are spelled out.  With these orders the sample coordinates and the
// bilinear taps reproduce the reference's CPU arithmetic bit for bit.
__device__ __forceinline__ float dot3_chain(float a0, float b0, float a1, float b1, float a2, float b2) {
  return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, mul_rn(a0, b0)));
}
// [a0 a1 a2 a3] . [b0 b1 b2 1]: the homogeneous term is fma(a3, 1, acc) = a rounded addition
__device__ __forceinline__ float dot4h_chain(float a0, float b0, float a1, float b1, float a2, float b2, float a3) {
  return add_rn(dot3_chain(a0, b0, a1, b1, a2, b2), a3);
}

// World point of a plane-sweep / back-projected sample: X = R^T (K^-1 [x z, y z, z] - t) (utils.py:98-106,
// lightningmodel.py:142-144).  cam = per-image block [0..8] K^-1, [9..17] R, [18..20] t.
__device__ __forceinline__ void world_point(const float* cam, float xf, float yf, float z, float& X, float& Y, float& Z) {
  const float p0 = mul_rn(xf, z), p1 = mul_rn(yf, z), p2 = z;
  const float c0 = sub_rn(dot3_chain(cam[0], p0, cam[1], p1, cam[2], p2), cam[18]);
  const float c1 = sub_rn(dot3_chain(cam[3], p0, cam[4], p1, cam[5], p2), cam[19]);
  const float c2 = sub_rn(dot3_chain(cam[6], p0, cam[7], p1, cam[8], p2), cam[20]);
  X = dot3_chain(cam[9], c0, cam[12], c1, cam[15], c2);
  Y = dot3_chain(cam[10], c0, cam[13], c1, cam[16], c2);
  Z = dot3_chain(cam[11], c0, cam[14], c1, cam[17], c2);
}

// x / c for a wave-uniform c with the correctly rounded reciprocal rc: q0 = x rc, one residual correction.  The result is
// the correctly rounded quotient (Markstein) for finite normal operands -- the same number as the IEEE division sequence.
__device__ __forceinline__ float div_uniform(float x, float c, float rc) {
  const float q0 = mul_rn(x, rc);
  return __builtin_fmaf(__builtin_fmaf(-q0, c, x), rc, q0);
}

// x / den and y / den with ONE shared reciprocal: the instruction sequence hipcc emits for an IEEE-correct fp32 division
// (v_rcp + one Newton step, quotient, two residual corrections) minus the two v_div_scale per quotient: den = |q_z| + 1e-8
// lies in [1e-8, ~1e7] and the quotients are pixel coordinates, far from the exponent ranges in which the scaling acts, so
// the bits are those of x / den (v_div_fixup still handles zero / inf / nan operands).  13 instructions instead of 22 for
// the two quotients of every sample.
__device__ __forceinline__ void div2_shared(float x, float y, float den, float& qx, float& qy) {
  float r = __builtin_amdgcn_rcpf(den);
  r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.f), r, r);
  float q = mul_rn(x, r);
  q = __builtin_fmaf(__builtin_fmaf(-den, q, x), r, q);
  q = __builtin_fmaf(__builtin_fmaf(-den, q, x), r, q);
  qx = __builtin_amdgcn_div_fixupf(q, den, x);
  q = mul_rn(y, r);
  q = __builtin_fmaf(__builtin_fmaf(-den, q, y), r, q);
  q = __builtin_fmaf(__builtin_fmaf(-den, q, y), r, q);
  qy = __builtin_amdgcn_div_fixupf(q, den, y);
}

// Source-view sample position of a world point: q = P [X;1]; uv = q_xy / (|q_z| + 1e-8) (mvsnet.py:199-202); normalised
// with the IMAGE size (:205-206); un-normalised by grid_sample with the FEATURE size (align_corners=True).
__device__ __forceinline__ void sample_position(const float* Pm, float X, float Y, float Z, float Wm1, float rWm1, float Hm1,
                                                float rHm1, float Wfm1, float Hfm1, float& ix, float& iy) {
  const float qx = dot4h_chain(Pm[0], X, Pm[1], Y, Pm[2], Z, Pm[3]);
  const float qy = dot4h_chain(Pm[4], X, Pm[5], Y, Pm[6], Z, Pm[7]);
  const float qz = dot4h_chain(Pm[8], X, Pm[9], Y, Pm[10], Z, Pm[11]);
  const float zb = add_rn(fabsf(qz), 1e-8f);
  float u, v;
  div2_shared(qx, qy, zb, u, v);
  // g = 2 q - 1: doubling is exact, so fma(q, 2, -1) rounds once, exactly where the reference's (q * 2) - 1 does;
  // ((g + 1) * 0.5) * (Wf - 1): halving is exact as well, so it is folded into the (exactly halved) feature extent --
  // the same single rounding of the same real product (3 instructions per coordinate instead of 5; same bits)
  const float gx = __builtin_fmaf(div_uniform(u, Wm1, rWm1), 2.f, -1.f);
  const float gy = __builtin_fmaf(div_uniform(v, Hm1, rHm1), 2.f, -1.f);
  ix = mul_rn(add_rn(gx, 1.f), mul_rn(0.5f, Wfm1));
  iy = mul_rn(add_rn(gy, 1.f), mul_rn(0.5f, Hfm1));
}

// conv0 of CostRegNet as a depth march (conv0z.hip).  f32 = false: split variance volume [n][4][hi, lo][D][H][W] -> split
// activation [n][hi, lo][D][H][W] (16-byte slots of 8 bf16), `wimg` = the split-bf16 weight image of conv0 (costreg.hip, c0bf).
// f32 = true: fp32 channel-last volume [n][4][2 halves][D][H][W] (16-byte slots of 4 floats) -> fp32 [n, 8, D, H, W],
// `wimg` = the fp32 fragment image (c0f32).
int launch_conv0z(bool f32, const void* in, const float* wimg, const float* bias, void* out, int n, int D, int H, int W,
                  hipStream_t s);

// conv9 + conv0 skip + prob of CostRegNet as a depth march (conv9z.hip): u8 [n][2][hi, lo][D/2][H/2][W/2] and the conv0 skip
// [n][hi, lo][D][H][W] (split layouts) -> x_reg [n, D, H, W]; `wbf` = the split-bf16 image of conv9 (costreg.hip, c9bf),
// `wprob` = the pair-interleaved prob weights
int launch_conv9z(const void* u8_split, const void* c0_split, const float* wbf, const float* bias9, const float* wprob,
                  const float* bprob, float* out, int n, int D, int H, int W, hipStream_t s);

// conv1 + conv2 of CostRegNet as one depth march (conv12z.hip, experiment): conv0 output [n][hi, lo][D][H][W] (split layout)
// -> conv2 output [n][2 groups][hi, lo][D2][H2][W2]; `w1` / `w2` = the split-bf16 images of conv1 / conv2 (costreg.hip, cgbf)
int launch_conv12z(const void* c0_split, const float* w1, const float* w2, const float* b1, const float* b2, void* out_split,
                   int n, int D, int H, int W, hipStream_t s);

// PropagationNet as one row-marching kernel (propz.hip, round 6): the fragment images of a layer (layer 0..3; `cinp` = padded input
// channels of layer 0: 8 | 24 | 40) packed from BN-folded weights [cout][cin][3][3], and the launch.  `depth` is [B, h0, w0]; with
// index tables iy [H] / ix [W] (device, may be null = identity with h0 == H, w0 == W) the nearest-neighbour resize of the depth
// (eval-3dvnet.py:103,111,119) happens in the kernel's addressing.
size_t propz_image_words(int layer, int cinp);
void propz_pack_layer(int layer, int cinp, int cin, int cout, const float* w_folded, unsigned* out, bool f32);
int launch_propz(int cinp, bool f32, const float* feat, const float* depth, const int* iy, const int* ix, float* out,
                 const float* const w[4], const float* const bias[4], int B, int Cf, int H, int W, int h0, int w0, hipStream_t s);

// [n_img, C, HW] -> [n_img, HW, C] (C in {16, 32}); defined in psv_variance.hip
int transpose_channel_last(const float* feat, float* featT, int n_img, int C, int HW, hipStream_t s);

}  // namespace v3d
