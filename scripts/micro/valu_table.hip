// Issue cost table of the VALU instructions the warp kernel uses (gfx950): cycles per wave64 instruction per SIMD at 1, 2, 4
// waves per SIMD on one CU (s_memtime), 8 independent destination registers per wave, no clobber-induced s_nops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void k(long long* out, float* sink, int iters, float sxv) {
  float a[8];
  f2 p[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (float)(threadIdx.x + j) + 1.5f; p[j] = (f2){a[j], a[j] + 1.f}; }
  float x = 1.0001f, y = 0.5f;
  f2 xx = {1.0001f, 0.9999f}, yy = {0.5f, 0.25f};
  const unsigned long long m = __builtin_amdgcn_read_exec();
  unsigned long long mo = 0;
  const float sx = __builtin_amdgcn_readfirstlane(sxv);
  long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 2) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[j]) : "v"(x), "v"(y): "vcc");
        if constexpr (OP == 3) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %3" : "+v"(a[j]) : "v"(x), "v"(y), "s"(m));
        if constexpr (OP == 4) asm volatile("v_cmp_ge_f32 vcc, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y): "vcc");
        if constexpr (OP == 5) asm volatile("v_cmp_ge_f32_e64 %1, %2, %0" : "+v"(a[j]), "=s"(mo) : "v"(x), "v"(y));
        if constexpr (OP == 6) asm volatile("v_floor_f32 %0, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 7) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 9) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 10) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 11) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 12) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 13) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 14) asm volatile("v_div_scale_f32 %0, vcc, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y): "vcc");
        if constexpr (OP == 15) asm volatile("v_div_fmas_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y): "vcc");
        if constexpr (OP == 16) asm volatile("v_div_fixup_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 17) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 18) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 19) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 20) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 21) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 22) asm volatile("v_mul_f32 %0, %4, %0" : "+v"(a[j]) : "v"(x), "v"(y), "s"(m), "s"(sx));
        if constexpr (OP == 23) asm volatile("v_cmp_ge_f32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[j]) : "v"(x), "v"(y): "vcc");
        if constexpr (OP == 24) asm volatile("v_cmp_ge_f32_e64 %1, %2, %0\n v_cndmask_b32_e64 %0, %2, %0, %1" : "+v"(a[j]), "=s"(mo) : "v"(x), "v"(y));
        if constexpr (OP == 25) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 26) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 27) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 28) asm volatile("v_trunc_f32 %0, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 29) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 30) asm volatile("v_fract_f32 %0, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 31) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 32) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 33) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 34) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(xx), "v"(yy));
        if constexpr (OP == 35) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[j]) : "v"(xx), "v"(yy));
        if constexpr (OP == 36) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 37) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 38) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 39) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 40) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 41) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(a[j]) : "v"(x), "v"(y));
        if constexpr (OP == 42) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(xx), "s"(m));          // SGPR pair operand
        if constexpr (OP == 43) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j & 1]) : "v"(xx), "v"(yy));     // 2 dependent chains
        if constexpr (OP == 44) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j & 1]) : "v"(xx), "s"(m));      // both
        if constexpr (OP == 45) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j & 1]) : "v"(x), "v"(y));          // scalar FMA, 2 chains
        if constexpr (OP == 46) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x), "s"(sx));             // scalar FMA, SGPR operand
      }
    }
  }
  long long c1 = __builtin_readcyclecounter();
  float s = (float)(mo & 1);
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j] + p[j][0] + p[j][1];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int OP>
void run(const char* name, long long* d, float* s) {
  const int iters = 5000;
  const double n_inst = (double)iters * 32;
  printf("%-72s", name);
  for (int wps : {1, 2, 4}) {
    k<OP><<<1, 256 * wps>>>(d, s, iters, 1.f);
    (void)hipDeviceSynchronize();
    k<OP><<<1, 256 * wps>>>(d, s, iters, 1.f);
    (void)hipDeviceSynchronize();
    long long h[16];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < 4 * wps; ++w) mx = h[w] > mx ? h[w] : mx;
    printf(" %dw %6.2f", wps, (double)mx / (n_inst * wps));
  }
  printf("\n");
}
int main() {
  long long* d; float* s;
  (void)hipMalloc(&d, 16 * 8); (void)hipMalloc(&s, 1024 * 4);
  run<0>("v_fma_f32 %0, %1, %2, %0", d, s);
  run<1>("v_mul_f32 %0, %1, %0", d, s);
  run<2>("v_cndmask_b32 %0, %1, %0, vcc", d, s);
  run<3>("v_cndmask_b32_e64 %0, %1, %0, %3", d, s);
  run<4>("v_cmp_ge_f32 vcc, %1, %0", d, s);
  run<5>("v_cmp_ge_f32_e64 %3, %1, %0", d, s);
  run<6>("v_floor_f32 %0, %0", d, s);
  run<7>("v_cvt_i32_f32 %0, %0", d, s);
  run<8>("v_rcp_f32 %0, %0", d, s);
  run<9>("v_max_f32 %0, %1, %0", d, s);
  run<10>("v_and_b32 %0, %1, %0", d, s);
  run<11>("v_lshlrev_b32 %0, 3, %0", d, s);
  run<12>("v_mad_u32_u24 %0, %1, %2, %0", d, s);
  run<13>("v_mul_lo_u32 %0, %1, %0", d, s);
  run<14>("v_div_scale_f32 %0, vcc, %1, %2, %0", d, s);
  run<15>("v_div_fmas_f32 %0, %1, %2, %0", d, s);
  run<16>("v_div_fixup_f32 %0, %1, %2, %0", d, s);
  run<17>("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", d, s);
  run<18>("v_bfe_u32 %0, %0, 3, 5", d, s);
  run<19>("v_perm_b32 %0, %1, %0, %2", d, s);
  run<20>("v_add_u32 %0, %1, %0", d, s);
  run<21>("v_fmac_f32 %0, %1, %2", d, s);
  run<22>("v_mul_f32 %0, %4, %0", d, s);
  run<23>("v_cmp_ge_f32 vcc, %1, %0 ; v_cndmask_b32 %0, %1, %0, vcc", d, s);
  run<24>("v_cmp_ge_f32_e64 %3, %1, %0 ; v_cndmask_b32_e64 %0, %1, %0, %3", d, s);
  run<25>("v_and_or_b32 %0, %1, %2, %0", d, s);
  run<26>("v_med3_f32 %0, %1, %2, %0", d, s);
  run<27>("v_sub_f32 %0, %1, %0", d, s);
  run<28>("v_trunc_f32 %0, %0", d, s);
  run<29>("v_cvt_f32_i32 %0, %0", d, s);
  run<30>("v_fract_f32 %0, %0", d, s);
  run<31>("v_min_f32 %0, %1, %0", d, s);
  run<32>("v_mul_u32_u24 %0, %1, %0", d, s);
  run<33>("v_lshl_add_u32 %0, %0, 3, %1", d, s);
  run<34>("v_pk_fma_f32 %5, %6, %7, %5", d, s);
  run<35>("v_pk_mul_f32 %5, %6, %5", d, s);
  run<36>("v_cvt_pk_bf16_f32 %0, %1, %0", d, s);
  run<37>("v_add_f32 %0, %1, %0", d, s);
  run<38>("v_xor_b32 %0, %1, %0", d, s);
  run<39>("v_or_b32 %0, %1, %0", d, s);
  run<40>("v_add3_u32 %0, %1, %2, %0", d, s);
  run<41>("v_lshrrev_b32 %0, 16, %0", d, s);
  run<42>("v_pk_fma_f32 v, v, s[pair], v   (8 independent accumulators)", d, s);
  run<43>("v_pk_fma_f32 v, v, v, v         (2 dependent chains)", d, s);
  run<44>("v_pk_fma_f32 v, v, s[pair], v   (2 dependent chains)", d, s);
  run<45>("v_fma_f32 v, v, v, v            (2 dependent chains)", d, s);
  run<46>("v_fma_f32 v, v, s, v            (8 independent accumulators)", d, s);
  return 0;
}
