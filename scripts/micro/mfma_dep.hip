// cycles per v_mfma_f32_16x16x32_bf16 when NACC independent accumulators are cycled (dependent-issue latency)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void k(long long* out, float* sink) {
  f32x4 acc[NACC] = {};
  bf16x8 a = {}, b = {};
  long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < 20000; ++i) {
#pragma unroll
    for (int r = 0; r < 12 / NACC; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  }
  long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = c1 - c0;
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0];
  sink[threadIdx.x] = s;
}
template <int NACC> void run(long long* d, float* s) {
  for (int r = 0; r < 2; ++r) { k<NACC><<<1, 64>>>(d, s); (void)hipDeviceSynchronize(); }
  long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%d independent accumulators: %.2f cycles per MFMA\n", NACC, h / (20000.0 * 12));
}
int main() {
  long long* d; float* s; (void)hipMalloc(&d, 16); (void)hipMalloc(&s, 1024);
  run<1>(d, s); run<2>(d, s); run<3>(d, s); run<4>(d, s); run<6>(d, s);
  return 0;
}
