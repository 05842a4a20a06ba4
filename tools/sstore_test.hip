// Dev check: do scalar stores (s_store_dwordx2 + s_dcache_wb) work on gfx950, and are the data SGPRs read at issue?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned long long* p, int n) {
  unsigned long long* dst = p + (size_t)blockIdx.x * 64 * n;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  dst += wave * n;
  for (int i = 0; i < n; ++i) {
    unsigned long long m = __ballot(((threadIdx.x * 2654435761u + i * 40503u) >> 7) & 1);
    asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(m), "s"(dst + i) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}
int main() {
  const int nb = 512, n = 64;
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * nb * 64 * n);
  hipMemset(d, 0, sizeof(unsigned long long) * nb * 64 * n);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, n);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  std::vector<unsigned long long> h((size_t)nb * 64 * n);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 4; ++w)
      for (int i = 0; i < n; ++i) {
        unsigned long long want = 0;
        for (int l = 0; l < 64; ++l) { unsigned t = (unsigned)(w * 64 + l); if (((t * 2654435761u + i * 40503u) >> 7) & 1) want |= 1ull << l; }
        if (h[((size_t)b * 64 + w) * n + i] != want) ++bad;
      }
  printf("scalar stores: %ld mismatches of %d\n", bad, nb * 4 * n);
  return bad != 0;
}
