// Dev tool (hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o valu_rate): issue rate of independent vector
// instructions, one workgroup of 1-16 waves on one CU; every wave times its own loop with s_memtime.  DESIGN.md section 7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void k(uint32_t* out, int iters) {
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0x7feb352du));
        if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(0x7feb35u));
        if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(0x7feb352du));
        if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(0x7feb35u));
        if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %0, 16" : "+v"(a[i]));
      }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ((long long*)out)[4096 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  uint32_t* d; hipMalloc(&d, 1 << 20);
  const char* names[5] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_xor_b32", "v_mad_u32_u24", "v_alignbit_b32"};
  for (int threads : {64, 256, 512, 768, 1024}) for (int op = 0; op < 5; ++op) {
    const int iters = 200;
    if (op == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, d, iters);
    if (op == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, d, iters);
    if (op == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, d, iters);
    if (op == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, d, iters);
    if (op == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    long long c[16]; hipMemcpy(c, (char*)d + 4096 * 8, 8 * 16, hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1ll << 60; for (int w = 0; w < threads / 64; ++w) { mx = c[w] > mx ? c[w] : mx; mn = c[w] < mn ? c[w] : mn; }
    printf("%d threads  %-16s fastest wave %.2f  slowest wave %.2f cycles per instruction; per SIMD one instruction per %.2f cycles\n", threads, names[op], (double)mn / (iters * 64.0), (double)mx / (iters * 64.0), (double)mx / (iters * 64.0) / (threads < 256 ? 1 : threads / 256));
  }
  return 0;
}
