// calibrate __builtin_readcyclecounter (s_memtime) against wall_clock64 (100 MHz) and v_mfma issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(long long* out, float* sink) {
  long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  f32x4 acc[4] = {};
  bf16x8 a = {}, b = {};
  for (int i = 0; i < 100000; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  }
  long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  sink[threadIdx.x] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
}
int main() {
  long long* d; float* s; (void)hipMalloc(&d, 16); (void)hipMalloc(&s, 1024);
  for (int r = 0; r < 2; ++r) { k<<<1, 64>>>(d, s); (void)hipDeviceSynchronize(); }
  long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("400000 MFMA 16x16x32 bf16: cyclecounter %lld ticks, wall %lld ticks (100 MHz) = %.1f us -> %.2f cyc-ticks/ns, %.2f ticks per MFMA\n",
         h[0], h[1], h[1] / 100.0, h[0] / (h[1] * 10.0), h[0] / 400000.0);
  return 0;
}
