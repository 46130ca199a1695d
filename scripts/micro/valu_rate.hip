// VALU issue rates on gfx950: cycles per wave64 instruction per SIMD for scalar / packed fp32 FMA and friends,
// at 1, 2, 4 waves per SIMD on one CU (s_memtime) and chip-wide (hipEvents).  Answers: does v_pk_fma_f32 double the
// FMA rate of v_fma_f32 (it does not if the SIMD retires 32 FMAs per clock either way)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { kFma = 0, kPkFma = 1, kPkFmaBcast = 2, kPkMul = 3, kPkAdd = 4, kAdd = 5, kCndmask = 6, kFmaLds = 7, kMov = 8 };

template <int OP>
__global__ __launch_bounds__(1024) void k(long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float a[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  f2 p[8];
  f4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (float)(threadIdx.x + j); p[j] = (f2){a[j], a[j] + 1.f}; }
  float x = 1.0001f, y = 0.5f;
  f2 xx = {1.0001f, 0.9999f}, yy = {0.5f, 0.25f};
  const int ofs = (threadIdx.x & 7) * 16;
  long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (OP == kFma) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == kPkFma) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(xx), "v"(yy));
        if constexpr (OP == kPkFmaBcast)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[j]) : "v"(xx), "v"(yy));
        if constexpr (OP == kPkMul) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[j]) : "v"(xx));
        if constexpr (OP == kPkAdd) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[j]) : "v"(xx));
        if constexpr (OP == kAdd) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(x));
        if constexpr (OP == kCndmask) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[j]) : "v"(x));
        if constexpr (OP == kMov) asm volatile("v_mov_b32 %0, %1" : "+v"(a[j]) : "v"(x));
        if constexpr (OP == kFmaLds) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y));
          if (j == 0) q += *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(lds) + ofs + r * 128);
        }
      }
    }
  }
  long long c1 = __builtin_readcyclecounter();
  float s = q[0] + q[1] + q[2] + q[3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j] + p[j][0] + p[j][1];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int OP>
void run(const char* name, long long* d, float* s) {
  const int iters = 20000;
  const double n_inst = (double)iters * 32;
  printf("%-28s", name);
  for (int wps : {1, 2, 4}) {   // waves per SIMD on ONE CU
    k<OP><<<1, 256 * wps>>>(d, s, iters);
    (void)hipDeviceSynchronize();
    k<OP><<<1, 256 * wps>>>(d, s, iters);
    (void)hipDeviceSynchronize();
    long long h[16];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < 4 * wps; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("  %dw/SIMD: %5.2f cyc/inst/SIMD", wps, (double)mx / (n_inst * wps));
  }
  // chip-wide: 512 blocks x 1024 threads = 8 waves per SIMD on 256 CUs
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<OP><<<512, 1024>>>(d, s, iters);
  (void)hipEventRecord(e0);
  k<OP><<<512, 1024>>>(d, s, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double winst = n_inst * 512 * 16;   // wave instructions
  printf("  chip 8w/SIMD: %6.1f G wave-inst/s (%.2f ns-cyc@2.4GHz/inst/SIMD)\n", winst / ms / 1e6,
         ms * 1e-3 * 2.4e9 / (winst / 1024));
}

int main() {
  long long* d; float* s;
  (void)hipMalloc(&d, 512 * 16 * 8); (void)hipMalloc(&s, 512 * 1024 * 4);
  run<kFma>("v_fma_f32", d, s);
  run<kPkFma>("v_pk_fma_f32", d, s);
  run<kPkFmaBcast>("v_pk_fma_f32 op_sel bcast", d, s);
  run<kPkMul>("v_pk_mul_f32", d, s);
  run<kPkAdd>("v_pk_add_f32", d, s);
  run<kAdd>("v_add_f32", d, s);
  run<kCndmask>("v_cndmask_b32", d, s);
  run<kMov>("v_mov_b32", d, s);
  run<kFmaLds>("v_fma_f32 + ds_read_b128/8", d, s);
  return 0;
}
