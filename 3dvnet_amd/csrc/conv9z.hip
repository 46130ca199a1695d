// conv9 + conv0 skip + prob of CostRegNet (mvsnet.py:152-153,161-162: x_reg = prob(conv0 + ReLU(BN(deconv9(u8))))) as a DEPTH
// MARCH (round 4), the structure of conv0z.hip applied to the third-largest kernel of the step.  Arithmetic, weight images
// and accumulation orders are those of conv9_prob_kernel (costreg.hip): the transposed convolution as a GEMM over 2x2x2 output
// cells on split-bf16 matrix cores (16 rows = 2 x parities x 8 output channels, K = 32 = 2 x inputs x 16 input channels, the
// (z, y) parities / inputs as 9 weight blocks), BN bias + ReLU + skip in fp32, the 8 -> 1 prob conv as packed fp32 FMAs over a
// u9 tile in LDS (the prob conv's 216 products per output are summed in six chains instead of two: same products, another
// order).  What changes is the schedule:
//
//   * the old kernel built a 6 x 10 x 30 u9 tile for 4 x 8 x 28 outputs (the transposed conv, its skip loads and staging
//     done 2.0x) in five barrier-separated phases -- loads -> staging -> MFMA -> u9 tile -> prob -- of which the arithmetic
//     was ~15 % of the time at two workgroups per CU;
//   * here a workgroup owns an 8 x 28 (y, x) tile and walks z one 2-plane cell layer per step: the u8 input plane and the two
//     conv0-skip planes of a step arrive through LDS-DMA rings three steps deep (no staging registers), four PRODUCER waves
//     turn them into the two new u9 planes of a ring of six, four CONSUMER waves run the prob conv one step behind on the
//     planes that are complete -- one barrier per step, the two roles overlap, the (y, x) halo is the only recompute (1.34x).
//
// STATUS (round 4): an experiment, NOT the default and since round 5 not even part of the default build (-DV3D_EXPERIMENTS, then
// v3d_set_option("c9_kernel", 1)).  It is correct (tests/test_costvolume_gpu.py passes
// with it: goldens, fuzz, batch invariance) but measures 0.51 ms per 64 cfg2 views against 0.46 for the tile kernel, after:
// prob weights out of the step loop's vector loads (0.60 -> 0.49: after the consumers' stores the compiler could not prove them
// invariant; __restrict__ + an LDS copy), all LDS reads of a step issued up front, the transposed conv's five accumulator
// streams issued round-robin, six independent FMA chains in the prob conv.  Cycle counters of wave 0 / wave 4
// (scripts/phase_conv0z.py --kernel conv9z): per step of ~5 200 cycles the consumers spend ~5 000 in the prob conv of one plane
// half (72 + 56 ds_read_b128, 216 packed FMAs -- ~1 400 cycles of issue), the producers ~3 500 (DMA issue 540, 26 reads + 39 MFMAs
// 1 640, u9 assembly 1 190); with the prob conv ablated the kernel still takes 0.34 ms.  Unlike conv0 -- 189 MFMAs per wave and
// step, a dense matrix stream -- a step here is a handful of short, dependent phases (reads -> 39 MFMAs -> 5 x (skip, VALU, 2
// stores); 4 x (32 reads -> 54 FMAs)), and with ONE wave per SIMD and role every LDS round trip and every dependent
// instruction is exposed; the tile kernel's 16 waves per CU hide the same latencies by brute occupancy although it computes
// 2.0x.  What would be needed: 3-4 waves per SIMD inside the march (registers: 217 now, <= 128 needed) or two workgroups per CU
// (LDS: 152 KB now, <= 80 KB needed) -- i.e. a smaller tile with more recompute.  Kept as a documented negative result.
#include <type_traits>
#include <utility>

#include "v3d_common.h"

#ifndef V3D_EXPERIMENTS
// The default build does not carry this experiment (v3d_set_option("c9_kernel", 1) is refused): build with
// V3D_EXTRA_FLAGS=-DV3D_EXPERIMENTS to time it (scripts/phase_conv0z.py --kernel conv9z).
int v3d::launch_conv9z(const void*, const void*, const float*, const float*, const float*, const float*, float*, int, int, int, int,
                       hipStream_t) {
  return v3d::fail(V3D_ERR_UNSUPPORTED, "conv9z: experimental kernel, not in this build (-DV3D_EXPERIMENTS)");
}
#else


namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct C9Z {
  static constexpr int TH = 8, TW = 28;                       // prob output tile (y, x)
  static constexpr int HH = TH + 2, HW = TW + 2;              // u9 rows / columns with halo
  static constexpr int CY = HH / 2, CX = HW / 2;              // 5 x 15 cells per cell layer
  static constexpr int VY = CY + 1;                           // 6 input rows
  static constexpr int RS = 32;                               // row stride (slots / floats pairs) of every ring
  // conv0-skip ring: per plane [hi, lo][HH rows][32 slots] 16-byte slots (the depth-march conv0's tile rows)
  static constexpr int SK_PIECES = HH / 2, SK_HL = SK_PIECES * 1024, SK_PLANE = 2 * SK_HL;       // 10 KB
  // u8 ring: per input plane [2 channel groups][hi, lo][2 pieces of 3 rows x 21 slots (17 used)]
  static constexpr int IN_RS = 21, IN_PIECES = VY / 3, IN_GH = IN_PIECES * 1024, IN_PLANE = 4 * IN_GH;       // 8 KB
  static_assert(VY % 3 == 0 && CX + 2 <= IN_RS && 3 * IN_RS <= 64, "u8 piece geometry");
  static constexpr int R = 3;                                 // DMA ring depth (steps)
  static constexpr int SK_BYTES = R * 2 * SK_PLANE;           // two skip planes per step
  static constexpr int IN_BYTES = (R + 1) * IN_PLANE;         // a step reads input planes J and J + 1: one more slot
  // u9 ring: [6 planes][4 channel pairs][HH rows][32 x][2 floats]
  static constexpr int U9_PLANE = 4 * HH * RS * 8, U9_N = 6, U9_BYTES = U9_N * U9_PLANE;
  static constexpr int WP_BYTES = 4 * 14 * 16;                // the prob weights [4 pairs][27 taps (+1)][2]
  static constexpr int LDS_BYTES = SK_BYTES + IN_BYTES + U9_BYTES + WP_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct C9ZParams {
  const void* u8;      // conv8 output, split layout [n][2 groups][hi, lo][D/2][H/2][W/2] 16-byte slots
  const void* c0;      // conv0 output (skip), split layout [n][hi, lo][D][H][W] 16-byte slots of 8 channels
  const void* wbf;     // [9 blocks][hi, lo][64 lanes][4 words] (costreg.hip, c9bf)
  const float* bias9;  // [8]
  const float* wprob;  // [4 channel pairs][27 taps][2]
  const float* bprob;  // [1]
  float* out;          // [n, D, H, W]
  int n, D, H, W, nty, ntx, nseg, seg_len, n_tasks;
};

#ifdef V3D_PHASE_TIMING
// developer build only: wave 0 (producer: marks 0-4) and wave 4 (consumer: marks 5-7) of a workgroup write their cycle counts
__device__ unsigned long long g_c9z_phase[8 * 1024];
#define C9Z_PHASE_DECL long long ph_t = __builtin_readcyclecounter(); long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define C9Z_PHASE_MARK(i) do { const long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#define C9Z_PHASE_FLUSH do { if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x < 1024) for (int i_ = 0; i_ < 8; ++i_) if ((threadIdx.x == 0) == (i_ < 5)) g_c9z_phase[blockIdx.x * 8 + i_] = (unsigned long long)ph_acc[i_]; } while (0)
#else
#define C9Z_PHASE_DECL
#define C9Z_PHASE_MARK(i)
#define C9Z_PHASE_FLUSH
#endif

// f(integral_constant<int, I>) for I = B .. E - 1, fully unrolled with compile-time indices
template <int B, int E, class F>
__device__ __forceinline__ void c9z_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    c9z_static_for<B + 1, E>(f);
  }
}
// The transposed conv's products per accumulator (pz, py) of a cell row, in (dz, dy) order: {B fragment = dz * 2 + dy, weight
// block = (pz ? 2 : dz) * 3 + (py ? 2 : dy)}; every entry is three MFMAs (hi*hi, hi*lo, lo*hi) into that accumulator.
struct C9ZJobs { int n; int bsel[4]; int blk[4]; };
constexpr C9ZJobs kC9ZJobs[4] = {{4, {0, 1, 2, 3}, {0, 1, 3, 4}}, {2, {1, 3, 0, 0}, {2, 5, 0, 0}},
                                 {2, {2, 3, 0, 0}, {6, 7, 0, 0}}, {1, {3, 0, 0, 0}, {8, 0, 0, 0}}};

#ifndef V3D_C9Z_ABLATE
#define V3D_C9Z_ABLATE 0     // developer ablations of the consumer: 1 no FMAs, 2 no row reads, 3 no weight reads, 4 no stores, 5 no prob at all
#endif

// one LDS-DMA piece = 1 KB = 64 lanes x 16 bytes (two ring rows of 32 slots); exec = the lanes that load
__device__ __forceinline__ void c9z_dma(unsigned lds_dst, const char* base, unsigned voff, unsigned long long mask) {
  unsigned long long sv;
  unsigned m0v;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_mov_b32 %[m0v], m0\n\t"
      "s_mov_b32 m0, %[dst]\n\t"
      "s_mov_b64 exec, %[k]\n\t"
      "global_load_lds_dwordx4 %[v], %[b]\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      "s_mov_b32 m0, %[m0v]"
      : [sv] "=&s"(sv), [m0v] "=&s"(m0v)
      : [dst] "s"(lds_dst), [b] "s"(base), [v] "v"(voff), [k] "s"(mask)
      : "memory");
}

// `wprob_r` = p.wprob as a __restrict__ kernel argument: the consumers read the prob weights inside the step loop, after their
// stores to `out` -- only if the compiler can exclude that those stores alias the weights do the reads stay scalar loads of
// loop invariants (without it: 54 vector loads per step and wave, and the step took 5 000 instead of 1 500 cycles).
__global__ __launch_bounds__(512, 2) void conv9z_kernel(C9ZParams p, const float* __restrict__ wprob_r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sk_ring = smem;                                  // conv0 skip planes
  unsigned char* const in_ring = smem + C9Z::SK_BYTES;                  // u8 planes
  float* const u9 = reinterpret_cast<float*>(smem + C9Z::SK_BYTES + C9Z::IN_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave8 < 4;
  const int wave = wave8 & 3;
  const int kq = lane >> 4, jn = lane & 15;
  const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const int D2 = p.D >> 1, H2 = p.H >> 1, W2 = p.W >> 1;
  const size_t HWo = (size_t)p.H * p.W, DHW = (size_t)p.D * HWo;
  const size_t HW2 = (size_t)H2 * W2, DHW2 = (size_t)D2 * HW2;
  const v3d::TileWalk walk = v3d::xcd_tile_walk(p.n_tasks);
  C9Z_PHASE_DECL;

  struct Task { int n, oy0, ox0, J0, nsteps, z0, z1; };
  auto decode = [&](int t) __attribute__((always_inline)) {
    Task q;                                                             // (view, z segment, y tile, x tile), x fastest
    const int tx = t % p.ntx; t /= p.ntx;
    const int ty = t % p.nty; t /= p.nty;
    const int seg = t % p.nseg;
    q.n = t / p.nseg;
    q.oy0 = ty * C9Z::TH; q.ox0 = tx * C9Z::TW;
    q.z0 = seg * p.seg_len; q.z1 = min(q.z0 + p.seg_len, p.D);         // out planes [z0, z1), both even
    q.J0 = q.z0 / 2 - 1;                                                // first cell layer: u9 planes z0 - 1, z0
    q.nsteps = (q.z1 - q.z0) / 2 + 2;                                   // producers: steps 0 .. nsteps - 2, consumers: 2 .. nsteps - 1
    return q;
  };
  // Step s of a task handles cell layer J = J0 + s: inputs u8 planes J, J + 1 and skip planes 2J + 1, 2J + 2 -> u9 planes
  // 2J + 1, 2J + 2 (producers); the consumers compute out planes 2J - 2, 2J - 1 from u9 planes 2J - 3 .. 2J.  One barrier B(s)
  // per step: behind it the DMA pieces of step s have landed (every issuing wave waited for its own), the u9 planes of step
  // s - 1 are written, and the ring slots step s - 1 read are free for the pieces of step s + 2.
  auto u9_slot = [](int gz) { return (gz + 6) % C9Z::U9_N; };          // gz >= -1

  if (producer) {
    // ================= producers: LDS-DMA streams, transposed conv on the matrix cores, BN + ReLU + skip -> u9 ring =======
    bf16x8 a_hi[9], a_lo[9];
    {
      const u32x4* wq = reinterpret_cast<const u32x4*>(p.wbf) + lane;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        a_hi[k] = __builtin_bit_cast(bf16x8, wq[(k * 2) * 64]);
        a_lo[k] = __builtin_bit_cast(bf16x8, wq[(k * 2 + 1) * 64]);
      }
    }
    const int px = kq >> 1, cbase = 4 * (kq & 1);
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = p.bias9[cbase + r];

#pragma unroll 1
    for (int t = walk.t; t < walk.end; t += walk.step) {
      const Task q = decode(t);
      // ---- DMA role of this wave: waves 0 / 1 the skip plane 2J + 1 / 2J + 2 (5 pieces of hi rows + 5 of lo rows), waves
      // 2 / 3 channel group 0 / 1 of the u8 plane J + 1 (3 pieces of hi rows + 3 of lo rows) ----------------------------------
      const bool dma_skip = wave < 2;
      unsigned voff[5];
      unsigned long long vmask[5];
      {
        const int j = lane >> 5, col = lane & 31;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          bool ok;
          if (dma_skip) {
            const int gy = q.oy0 - 1 + 2 * i + j, gx = q.ox0 - 1 + col;
            ok = col < C9Z::HW && gx >= 0 && gx < p.W && gy >= 0 && gy < p.H;
            voff[i] = ok ? (unsigned)((gy * p.W + gx) * 16) : 0u;
            vmask[i] = __ballot(ok || col >= 30);                       // slots 30, 31 are never read: keep-alive lanes
          } else {
            const int row = lane / C9Z::IN_RS, c21 = lane % C9Z::IN_RS;  // 3 rows of 21 slots per piece, lane 63 keeps alive
            const int gy = (q.oy0 >> 1) - 1 + 3 * i + row, gx = (q.ox0 >> 1) - 1 + c21;
            ok = i < C9Z::IN_PIECES && lane < 3 * C9Z::IN_RS && c21 <= C9Z::CX + 1 && gx >= 0 && gx < W2 && gy >= 0 && gy < H2;
            voff[i] = ok ? (unsigned)((gy * W2 + gx) * 16) : 0u;
            vmask[i] = __ballot(ok || lane == 63);
          }
        }
      }
      const char* const sk_base = reinterpret_cast<const char*>(p.c0) + ((size_t)q.n * 2) * DHW * 16;
      const char* const in_base = reinterpret_cast<const char*>(p.u8) + ((size_t)(q.n * 2 + (wave & 1)) * 2) * DHW2 * 16;
      // the pieces of step s (cell layer J): issued only when the plane exists; returns whether they were
      auto issue = [&](int s) __attribute__((always_inline)) {
        const int J = q.J0 + s;
        if (dma_skip) {
          const int gz = 2 * J + 1 + wave;
          if (s > q.nsteps - 2 || gz < 0 || gz >= p.D) return false;
          const char* bh = sk_base + (size_t)gz * HWo * 16;
          const char* bl = bh + DHW * 16;
          const unsigned dst = smem_lds + (unsigned)(((s % C9Z::R) * 2 + wave) * C9Z::SK_PLANE);
#pragma unroll
          for (int i = 0; i < 5; ++i) c9z_dma(dst + i * 1024, bh, voff[i], vmask[i]);
#pragma unroll
          for (int i = 0; i < 5; ++i) c9z_dma(dst + C9Z::SK_HL + i * 1024, bl, voff[i], vmask[i]);
        } else {
          const int jz = J + 1;                                         // plane J itself arrived with step s - 1
          if (s > q.nsteps - 2 || jz < 0 || jz >= D2) return false;
          const char* bh = in_base + (size_t)jz * HW2 * 16;
          const char* bl = bh + DHW2 * 16;
          const unsigned dst = smem_lds + C9Z::SK_BYTES + (unsigned)(((s + 1) % (C9Z::R + 1)) * C9Z::IN_PLANE + (wave & 1) * 2 * C9Z::IN_GH);
#pragma unroll
          for (int i = 0; i < C9Z::IN_PIECES; ++i) c9z_dma(dst + i * 1024, bh, voff[i], vmask[i]);
#pragma unroll
          for (int i = 0; i < C9Z::IN_PIECES; ++i) c9z_dma(dst + C9Z::IN_GH + i * 1024, bl, voff[i], vmask[i]);
        }
        return true;
      };
      // prologue: the u8 ring starts as zeros (out-of-volume voxels are never written by the DMA; planes -1 and D/2 stay
      // zero slots), plane J0 of the first step is fetched like a "step -1", then the pieces of steps 0 and 1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                                  // P0: the consumers are done with the previous task
      {
        u32x4* const rz = reinterpret_cast<u32x4*>(in_ring);
        for (int i = wave * 64 + lane; i < C9Z::IN_BYTES / 16; i += 256) rz[i] = (u32x4){0u, 0u, 0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __syncthreads();                                                  // P1: zeros are in place before any DMA lands
      if (!dma_skip) {
        const int jz = q.J0;                                            // input plane J0 -> slot 0 (the slot "step -1" fills)
        if (jz >= 0 && jz < D2) {
          const char* bh = in_base + (size_t)jz * HW2 * 16;
          const char* bl = bh + DHW2 * 16;
          const unsigned dst = smem_lds + C9Z::SK_BYTES + (unsigned)((wave & 1) * 2 * C9Z::IN_GH);
#pragma unroll
          for (int i = 0; i < C9Z::IN_PIECES; ++i) c9z_dma(dst + i * 1024, bh, voff[i], vmask[i]);
#pragma unroll
          for (int i = 0; i < C9Z::IN_PIECES; ++i) c9z_dma(dst + C9Z::IN_GH + i * 1024, bl, voff[i], vmask[i]);
        }
      }
      issue(0);
      bool fl1 = issue(1);                                              // are the pieces of step s + 1 in flight?

#pragma unroll 1
      for (int s = 0; s < q.nsteps; ++s) {
        // my pieces of step s have landed when at most those of step s + 1 are outstanding
        if (fl1) { if (dma_skip) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C9Z_PHASE_MARK(0);
        __syncthreads();                                                // B(s)
        C9Z_PHASE_MARK(1);
        fl1 = issue(s + 2);
        C9Z_PHASE_MARK(2);
        if (s > q.nsteps - 2) continue;                                 // the consumers' last step
        const int J = q.J0 + s;
        // an input plane outside the volume is a zero slot: plane J + 1 of this step goes to slot (s + 1) % 4, which a
        // skipped DMA leaves with the data of step s - 3 -> clear it (only the first / last cell layer of the volume)
        if (J + 1 < 0 || J + 1 >= D2) {
          u32x4* const rz = reinterpret_cast<u32x4*>(in_ring + ((s + 1) % (C9Z::R + 1)) * C9Z::IN_PLANE);
          for (int i = wave * 64 + lane; i < C9Z::IN_PLANE / 16; i += 256) rz[i] = (u32x4){0u, 0u, 0u, 0u};
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // (the waves read each other's part: one more rendezvous, only in this rare case; the consumers take part)
          asm volatile("s_barrier" ::: "memory");
        }
        const unsigned char* const in0 = in_ring + (s % (C9Z::R + 1)) * C9Z::IN_PLANE;            // plane J  (dz = 0)
        const unsigned char* const in1 = in_ring + ((s + 1) % (C9Z::R + 1)) * C9Z::IN_PLANE;      // plane J + 1 (dz = 1)
        // ---- transposed conv: this wave owns cell row cy = wave (4 accumulators) and accumulator `wave` of cell row 4.  One
        // wave per SIMD has nobody to hide an LDS round trip behind: every read of the step -- the B fragments of both cell
        // rows and the skip values of the five accumulators -- is issued up front, then the MFMAs, then the u9 stores --------
        const int pz4 = wave >> 1, py4 = wave & 1;                      // this wave's accumulator of cell row 4
        // B fragment of (cell row cy, dz, dy): input voxel (row cy + dy, x = jn + (kq >> 1)), channel group kq & 1
        auto bfrag = [&](int cy, int dz, int dy, u32x4& b_hi, u32x4& b_lo) __attribute__((always_inline)) {
          const unsigned char* const pl = dz ? in1 : in0;
          const int r = cy + dy;
          const unsigned off = (unsigned)((kq & 1) * 2 * C9Z::IN_GH + (r / 3) * 1024 + ((r % 3) * C9Z::IN_RS + jn + (kq >> 1)) * 16);
          b_hi = *reinterpret_cast<const u32x4*>(pl + off);
          b_lo = *reinterpret_cast<const u32x4*>(pl + off + C9Z::IN_GH);
        };
        u32x4 bh[2][2][2], bl[2][2][2];                                 // [own row / row 4][dz][dy]
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            bfrag(wave, dz, dy, bh[0][dz][dy], bl[0][dz][dy]);
            bfrag(C9Z::CY - 1, dz, dy, bh[1][dz][dy], bl[1][dz][dy]);
          }
        // skip values of accumulator (cy, pz, py): this lane's 4 channels of voxel (2 cy + py, 2 jn + px) of plane 2J + 1 + pz
        u32x2 skh[5], skl[5];
        auto skip_ld = [&](int i, int cy, int pz, int py) __attribute__((always_inline)) {
          const int hy = 2 * cy + py, hx = 2 * min(jn, C9Z::CX - 1) + px;
          const unsigned char* const sk = sk_ring + ((s % C9Z::R) * 2 + pz) * C9Z::SK_PLANE + (hy * C9Z::RS + hx) * 16 + (kq & 1) * 8;
          skh[i] = *reinterpret_cast<const u32x2*>(sk);
          skl[i] = *reinterpret_cast<const u32x2*>(sk + C9Z::SK_HL);
        };
#pragma unroll
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll
          for (int py = 0; py < 2; ++py) skip_ld(pz * 2 + py, wave, pz, py);
        skip_ld(4, C9Z::CY - 1, pz4, py4);

        f32x4 acc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Five accumulator streams (the own row's four + one of cell row 4) of 3 .. 12 dependent MFMAs each: issued round-robin
        // -- MFMA r of every stream that still has one -- so that consecutive matrix instructions never share an accumulator
        // (one wave per SIMD: a dependent MFMA waits for its predecessor's passes).  Every accumulator sees its products in the
        // tile kernel's order.  The stream of cell row 4 depends on the wave: four straight-line instances, no dynamic
        // register indexing.
        auto mfmas = [&](auto w_c) __attribute__((always_inline)) {
          constexpr int W = decltype(w_c)::value;
          c9z_static_for<0, 12>([&](auto r_c) __attribute__((always_inline)) {
            constexpr int r = decltype(r_c)::value, job = r / 3, typ = r % 3;
            c9z_static_for<0, 5>([&](auto st_c) __attribute__((always_inline)) {
              constexpr int st = decltype(st_c)::value;
              constexpr int a = st < 4 ? st : W;                       // job table of this stream's (pz, py)
              if constexpr (job < kC9ZJobs[a].n) {
                constexpr int bs = kC9ZJobs[a].bsel[job], blk = kC9ZJobs[a].blk[job];
                const u32x4 bhv = bh[st < 4 ? 0 : 1][bs >> 1][bs & 1], blv = bl[st < 4 ? 0 : 1][bs >> 1][bs & 1];
                const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bhv), b_lo = __builtin_bit_cast(bf16x8, blv);
                if constexpr (typ == 0) acc[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[blk], b_hi, acc[st], 0, 0, 0);
                else if constexpr (typ == 1) acc[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[blk], b_lo, acc[st], 0, 0, 0);
                else acc[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[blk], b_hi, acc[st], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              }
            });
          });
        };
        switch (wave) {
          case 0: mfmas(std::integral_constant<int, 0>{}); break;
          case 1: mfmas(std::integral_constant<int, 1>{}); break;
          case 2: mfmas(std::integral_constant<int, 2>{}); break;
          default: mfmas(std::integral_constant<int, 3>{}); break;
        }
        C9Z_PHASE_MARK(3);
        // ---- BN bias + ReLU + conv0 skip -> u9 planes 2J + 1 (pz = 0) and 2J + 2 (pz = 1); zero outside the volume ----------
        auto emit = [&](const f32x4& a, int i, int cy, int pz, int py) __attribute__((always_inline)) {
          if (jn >= C9Z::CX) return;
          const int hy = 2 * cy + py, hx = 2 * jn + px;
          const int gz = 2 * J + 1 + pz, gy = q.oy0 - 1 + hy, gx = q.ox0 - 1 + hx;
          const bool inside = gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
          const u32x2 sh = skh[i], sl = skl[i];
          const float skv[4] = {__uint_as_float(sh.x << 16) + __uint_as_float(sl.x << 16),
                                __uint_as_float(sh.x & 0xffff0000u) + __uint_as_float(sl.x & 0xffff0000u),
                                __uint_as_float(sh.y << 16) + __uint_as_float(sl.y << 16),
                                __uint_as_float(sh.y & 0xffff0000u) + __uint_as_float(sl.y & 0xffff0000u)};
          float val[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) val[r] = inside ? fmaxf(a[r] + bias[r], 0.f) + skv[r] : 0.f;
          float* const dst = u9 + (size_t)u9_slot(gz) * (C9Z::U9_PLANE / 4);
#pragma unroll
          for (int rp = 0; rp < 2; ++rp)
            *reinterpret_cast<f32x2*>(dst + ((((cbase >> 1) + rp) * C9Z::HH + hy) * C9Z::RS + hx) * 2) = (f32x2){val[2 * rp], val[2 * rp + 1]};
        };
#pragma unroll
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll
          for (int py = 0; py < 2; ++py) emit(acc[pz * 2 + py], pz * 2 + py, wave, pz, py);
        emit(acc[4], 4, C9Z::CY - 1, pz4, py4);
        C9Z_PHASE_MARK(4);
      }
    }
  } else {
    // ================= consumers: the prob conv, one step behind; wave = (plane of the step, y half) ==========================
    const int zsel = wave >> 1, y = (wave & 1) * 4 + (lane >> 4), xp = lane & 15;
    // The prob weights live in LDS (one copy per workgroup, written here): 216 floats are more than a wave's SGPRs hold, and as
    // scalar loads inside the step loop they cost an SMEM round trip per channel pair and step that one wave per SIMD cannot
    // hide; as uniform (broadcast) LDS reads they travel with the row reads of the same batch.
    f32x2* const wl = reinterpret_cast<f32x2*>(smem + C9Z::SK_BYTES + C9Z::IN_BYTES + C9Z::U9_BYTES);   // [4][28] (27 used)
    for (int i = tid - 256; i < 4 * 28; i += 256)
      wl[i] = i % 28 < 27 ? reinterpret_cast<const f32x2*>(wprob_r)[(i / 28) * 27 + i % 28] : (f32x2){0.f, 0.f};
    const float bsv = p.bprob[0];
#pragma unroll 1
    for (int t = walk.t; t < walk.end; t += walk.step) {
      const Task q = decode(t);
      __syncthreads();                                                  // P0
      __syncthreads();                                                  // P1
#pragma unroll 1
      for (int s = 0; s < q.nsteps; ++s) {
        C9Z_PHASE_MARK(5);
        __syncthreads();                                                // B(s)
        C9Z_PHASE_MARK(6);
        const int J = q.J0 + s;
        if ((J + 1 < 0 || J + 1 >= D2) && s <= q.nsteps - 2)           // the producers' rare zero-slot rendezvous (same condition,
          asm volatile("s_barrier" ::: "memory");                       // same steps): take part in it
        if (s < 2 || V3D_C9Z_ABLATE == 5) continue;
        const int gz = 2 * J - 2 + zsel;                                // out plane; u9 planes gz - 1 .. gz + 1 are complete
        if (xp < C9Z::TW / 2) {
          // One wave per SIMD: the tile kernel's loop -- a row read, waited for and multiplied at a time, every FMA into the same
          // two accumulators -- is an LDS round trip per 6 FMAs and two dependent chains of 108 packed FMAs: 5 000 cycles per
          // plane half (measured), which four waves per SIMD used to hide.  Here the 18 row reads of a channel pair are issued
          // together and the sums run in six independent chains (one pair of accumulators per kz, added at the end).
          f32x2 o0[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, o1[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll 1
          for (int cp = 0; cp < 4; ++cp) {                              // (rolled: one channel pair's 18 rows + 27 weights live)
            f32x4 r0[9], r1[9];
            f32x2 wv[28];                                               // two taps per 16-byte (broadcast) read
#pragma unroll
            for (int i = 0; i < 14; ++i) {
              const f32x4 w4 = V3D_C9Z_ABLATE == 3 ? (f32x4){(float)i, (float)cp, 1.f, 2.f}
                                                   : reinterpret_cast<const f32x4*>(wl)[cp * 14 + i];
              wv[2 * i] = (f32x2){w4.x, w4.y};
              wv[2 * i + 1] = (f32x2){w4.z, w4.w};
            }
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) {
              const float* const pl = u9 + (size_t)u9_slot(gz - 1 + kz) * (C9Z::U9_PLANE / 4);
#pragma unroll
              for (int ky = 0; ky < 3; ++ky) {
                const f32x2* row = reinterpret_cast<const f32x2*>(pl) + (cp * C9Z::HH + (y + ky)) * C9Z::RS + 2 * xp;
                if (V3D_C9Z_ABLATE == 2) { r0[kz * 3 + ky] = (f32x4){(float)xp, 1.f, (float)kz, 2.f}; r1[kz * 3 + ky] = (f32x4){(float)ky, 1.f, (float)y, 2.f}; continue; }
                r0[kz * 3 + ky] = *reinterpret_cast<const f32x4*>(row);
                r1[kz * 3 + ky] = *reinterpret_cast<const f32x4*>(row + 2);
              }
            }
            // k visits the three kz accumulator pairs in rotation and the two outputs alternate inside a row, so that
            // consecutive FMAs never hit the same accumulator (a dependent v_pk_fma_f32 costs ~20 cycles: left alone the
            // scheduler lines up each chain's 36 FMAs back to back); the scheduling barriers pin that order.  Every
            // accumulator still sees its products in the order (cp, ky, x tap).
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) {
              const int k = (kk % 3) * 3 + kk / 3, kz = k / 3;
              const f32x4 q0 = r0[k], q1 = r1[k];
              const f32x2 a0 = {q0.x, q0.y}, a1 = {q0.z, q0.w}, a2 = {q1.x, q1.y}, a3 = {q1.z, q1.w};
              const f32x2 w0 = wv[k * 3], w1 = wv[k * 3 + 1], w2 = wv[k * 3 + 2];
              if (V3D_C9Z_ABLATE == 1) { o0[kz] += a0 + w0; o1[kz] += a3 + w2 + a1 + a2 + w1; continue; }
              o0[kz] += a0 * w0; o1[kz] += a1 * w0;
              o0[kz] += a1 * w1; o1[kz] += a2 * w1;
              o0[kz] += a2 * w2; o1[kz] += a3 * w2;
              __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);       // 6 VALU, in this order
            }
          }
          const f32x2 s0 = (o0[0] + o0[1]) + o0[2], s1 = (o1[0] + o1[1]) + o1[2];
          const int gy = q.oy0 + y, gx = q.ox0 + 2 * xp;
          if (gz < q.z1 && gy < p.H && gx < p.W && (V3D_C9Z_ABLATE != 4 || s0.x == 1234.5f)) {
            const f32x2 res = {s0.x + s0.y + bsv, s1.x + s1.y + bsv};
            float* o = p.out + (size_t)q.n * DHW + ((size_t)gz * p.H + gy) * p.W + gx;
            if (gx + 1 < p.W && (p.W & 1) == 0) {
              *reinterpret_cast<f32x2*>(o) = res;
            } else {
              o[0] = res.x;
              if (gx + 1 < p.W) o[1] = res.y;
            }
          }
        }
      }
      C9Z_PHASE_MARK(7);
    }
  }
  C9Z_PHASE_FLUSH;
}

}  // namespace

int v3d::launch_conv9z(const void* u8_split, const void* c0_split, const float* wbf, const float* bias9, const float* wprob,
                       const float* bprob, float* out, int n, int D, int H, int W, hipStream_t s) {
  V3D_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, V3D_ERR_BAD_SHAPE, "conv9+prob: D, H, W must be even");
  V3D_REQUIRE((long long)D * H * W * 16 < (1ll << 32), V3D_ERR_BAD_SHAPE, "conv9+prob: volume too large for 32-bit plane offsets");
  C9ZParams p;
  p.u8 = u8_split; p.c0 = c0_split; p.wbf = wbf; p.bias9 = bias9; p.wprob = wprob; p.bprob = bprob; p.out = out;
  p.n = n; p.D = D; p.H = H; p.W = W;
  p.nty = (H + C9Z::TH - 1) / C9Z::TH; p.ntx = (W + C9Z::TW - 1) / C9Z::TW;
  int dev = 0, n_cu = 0;
  V3D_CHECK_HIP(hipGetDevice(&dev));
  V3D_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu <= 0) n_cu = 256;
  // z segments (even lengths): a task runs len / 2 + 3 steps
  const long long tiles = (long long)n * p.nty * p.ntx;
  long long best = -1;
  for (int nseg = 1; nseg <= D / 2; ++nseg) {
    int len = (D + nseg - 1) / nseg;
    len += len & 1;
    if ((long long)len * (nseg - 1) >= D) continue;
    const long long rounds = (tiles * nseg + n_cu - 1) / n_cu;
    const long long cost = rounds * (len / 2 + 2) + rounds;
    if (best < 0 || cost < best) { best = cost; p.nseg = nseg; p.seg_len = len; }
  }
  const long long tasks = tiles * p.nseg;
  V3D_REQUIRE(tasks > 0 && tasks < (1ll << 31), V3D_ERR_BAD_SHAPE, "conv9+prob: bad grid");
  p.n_tasks = (int)tasks;
  static bool attr_set[64] = {false};
  V3D_REQUIRE(dev >= 0 && dev < 64, V3D_ERR_UNSUPPORTED, "conv9+prob: device ordinal %d", dev);
  if (!attr_set[dev]) {
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)conv9z_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C9Z::LDS_BYTES));
    attr_set[dev] = true;
  }
  {
    v3d::TimedScope ts("costreg_conv9_prob", s);
    conv9z_kernel<<<v3d::persistent_grid(tasks, 1), 512, C9Z::LDS_BYTES, s>>>(p, p.wprob);
  }
  V3D_CHECK_LAUNCH("conv9z_kernel");
  return V3D_OK;
}

#ifdef V3D_PHASE_TIMING
extern "C" int v3d_debug_conv9z_phase_read(unsigned long long* out8_host, int n_blocks) {
  V3D_CHECK_HIP(hipDeviceSynchronize());
  static unsigned long long h[8 * 1024];
  V3D_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c9z_phase), sizeof(h)));
  for (int i = 0; i < 8; ++i) out8_host[i] = 0;
  for (int b = 0; b < n_blocks && b < 1024; ++b)
    for (int i = 0; i < 8; ++i) out8_host[i] += h[(size_t)b * 8 + i];
  return V3D_OK;
}
#endif
#endif  // V3D_EXPERIMENTS
