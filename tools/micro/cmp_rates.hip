// Dev tool (GPU): issue rate of 64-bit comparisons against 32-bit ones (inline assembly, so nothing is folded).
// hipcc --offload-arch=gfx950 -O3 -o cmp_rates tools/micro/cmp_rates.hip && ./cmp_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int OP>
__global__ void k(int *out, double a, double b) {
  double x = a + threadIdx.x, y = b + threadIdx.x * 0.5;
  int acc = threadIdx.x;
  for (int it = 0; it < N_IT; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
      if (OP == 1) asm volatile("v_cmp_lt_i64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
      if (OP == 2) asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"((int)threadIdx.x), "v"(acc) : "vcc");
      if (OP == 3) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : : "vcc");
      if (OP == 4) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"((float)threadIdx.x), "v"(1.5f) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int OP>
void run(const char *name, int *d) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4 * 2, threads = 256;
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 0.5);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 0.5);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double groups = (double)blocks * threads / 64 * N_IT * 8;
  printf("%-28s %7.3f ms  %5.2f clk per group at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (groups / 1024));
}
int main() {
  int *d;
  (void)hipMalloc(&d, sizeof(int) * 256 * 4 * 2 * 256);
  run<3>("addc alone", d);
  run<0>("v_cmp_lt_f64 + addc", d);
  run<1>("v_cmp_lt_i64 + addc", d);
  run<2>("v_cmp_lt_i32 + addc", d);
  run<4>("v_cmp_lt_f32 + addc", d);
  return 0;
}
