// diagnostic: semantics of v_permlane16_swap / v_permlane32_swap as a cross-row reduction step (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, float *y16, float *y32, float *a16, float *b16) {
  float p = x[threadIdx.x];
  unsigned pa = __builtin_bit_cast(unsigned, p), pb = __builtin_bit_cast(unsigned, p);
  asm volatile("" : "+v"(pb));   // keep the two operands distinct values for the compiler
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(pa), "+v"(pb));
  a16[threadIdx.x] = __builtin_bit_cast(float, pa);
  b16[threadIdx.x] = __builtin_bit_cast(float, pb);
  y16[threadIdx.x] = __builtin_bit_cast(float, pa) + __builtin_bit_cast(float, pb);
  unsigned qa = __builtin_bit_cast(unsigned, p), qb = qa;
  asm volatile("" : "+v"(qb));
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(qa), "+v"(qb));
  y32[threadIdx.x] = __builtin_bit_cast(float, qa) + __builtin_bit_cast(float, qb);
}
int main() {
  float h[64], o16[64], o32[64], a[64], b[64], *d, *e, *f, *g, *hh;
  for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i / 16)) * 100 + i % 16;   // row r: 100 * 2^r + lane
  hipMalloc(&d, 256); hipMalloc(&e, 256); hipMalloc(&f, 256); hipMalloc(&g, 256); hipMalloc(&hh, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, f, g, hh);
  hipMemcpy(o16, e, 256, hipMemcpyDeviceToHost); hipMemcpy(o32, f, 256, hipMemcpyDeviceToHost);
  hipMemcpy(a, g, 256, hipMemcpyDeviceToHost); hipMemcpy(b, hh, 256, hipMemcpyDeviceToHost);
  int bad16 = 0, bad32 = 0;
  for (int i = 0; i < 64; ++i) { bad16 += o16[i] != h[i] + h[i ^ 16]; bad32 += o32[i] != h[i] + h[i ^ 32]; }
  printf("xor16 mismatches %d, xor32 mismatches %d\n", bad16, bad32);
  for (int r = 0; r < 4; ++r) printf("row %d: in %.0f vdst' %.0f src' %.0f sum %.0f\n", r, h[16 * r], a[16 * r], b[16 * r], o16[16 * r]);
  return 0;
}
