// Dev tool (hipcc --offload-arch=gfx950 -O3 tools/rw_width_test.hip -o rw_width_test): the LayerNorm backward's memory
// pattern without its arithmetic -- a wave per 1536-byte row, NR tensors read and NW written per row, 65536 rows -- with
// the row cut into per-lane pieces three ways:
//   W8  : 3 instructions of 8 B per lane per tensor-row (512 B each: what the row kernels did up to round 3)
//   W16 : 1 instruction of 16 B per lane (columns 0-511) + 1 of 8 B per lane (columns 512-767)
//   W12 : 2 instructions of 12 B per lane (768 B each)
// Which part of "4.6-4.8 of 8 TB/s" is the instruction shape?  DESIGN.md section 7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
typedef __attribute__((ext_vector_type(3))) unsigned u3;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int MODE, int NR, int NW>
__global__ __launch_bounds__(256, 4) void k(const char* const* in, char* const* out, int rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const size_t base = (size_t)row * 1536;
    if (MODE == 0) {
      u2 a[3] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
#pragma unroll
      for (int t = 0; t < NR; ++t)
#pragma unroll
        for (int it = 0; it < 3; ++it) {
          const u2 v = __builtin_nontemporal_load((const u2*)(in[t] + base + (lane + 64 * it) * 8));
          a[it] += v * (unsigned)(t + 1);
        }
#pragma unroll
      for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int it = 0; it < 3; ++it) *(u2*)(out[t] + base + (lane + 64 * it) * 8) = a[it] + (unsigned)t;
      if (NW == 0 && a[0][0] + a[1][0] + a[2][1] == 0x12345678u) *(u2*)(out[0] + base) = a[0];   // (keeps the loads alive)
    } else if (MODE == 1) {
      u4 a = {0u, 0u, 0u, 0u};
      u2 b = {0u, 0u};
#pragma unroll
      for (int t = 0; t < NR; ++t) {
        const u4 v = __builtin_nontemporal_load((const u4*)(in[t] + base + lane * 16));
        const u2 w = __builtin_nontemporal_load((const u2*)(in[t] + base + 1024 + lane * 8));
        a += v * (unsigned)(t + 1);
        b += w * (unsigned)(t + 1);
      }
#pragma unroll
      for (int t = 0; t < NW; ++t) {
        *(u4*)(out[t] + base + lane * 16) = a + (unsigned)t;
        *(u2*)(out[t] + base + 1024 + lane * 8) = b + (unsigned)t;
      }
      if (NW == 0 && a[0] + a[3] + b[1] == 0x12345678u) *(u2*)(out[0] + base) = b;
    } else {
      u3 a[2] = {{0u, 0u, 0u}, {0u, 0u, 0u}};
#pragma unroll
      for (int t = 0; t < NR; ++t)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const u3 v = __builtin_nontemporal_load((const u3*)(in[t] + base + (lane + 64 * it) * 12));
          a[it] += v * (unsigned)(t + 1);
        }
#pragma unroll
      for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int it = 0; it < 2; ++it) *(u3*)(out[t] + base + (lane + 64 * it) * 12) = a[it] + (unsigned)t;
      if (NW == 0 && a[0][0] + a[1][2] == 0x12345678u) *(u2*)(out[0] + base) = (u2){a[0][0], a[1][1]};
    }
  }
}

template <int MODE, int NR, int NW>
static double run(const char* const* din, char* const* dout, int rows, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int rep = 0; rep < 7; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, NR, NW>), dim3(grid), dim3(256), 0, 0, din, dout, rows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms / 10);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2] * 1e3;
}

int main() {
  const int rows = 65536;
  const size_t bytes = (size_t)rows * 1536;
  char* bufs[8];
  for (int i = 0; i < 8; ++i) { hipMalloc(&bufs[i], bytes); hipMemset(bufs[i], i + 1, bytes); }
  const char** din; char** dout;
  hipMalloc(&din, 8 * sizeof(char*)); hipMalloc(&dout, 8 * sizeof(char*));
  hipMemcpy(din, bufs, 4 * sizeof(char*), hipMemcpyHostToDevice);
  hipMemcpy(dout, bufs + 4, 4 * sizeof(char*), hipMemcpyHostToDevice);
  const char* names[3] = {"3 x 8 B per lane", "16 B + 8 B per lane", "2 x 12 B per lane"};
  for (int grid : {1024, 2048}) {
#define CASE(NR, NW)                                                                                                   \
    {                                                                                                                  \
      const double t[3] = {run<0, NR, NW>(din, dout, rows, grid), run<1, NR, NW>(din, dout, rows, grid), run<2, NR, NW>(din, dout, rows, grid)}; \
      for (int m = 0; m < 3; ++m)                                                                                      \
        printf("grid %4d  %d read + %d written  %-20s %7.1f us  %6.0f GB/s (writes %5.0f GB/s)\n", grid, NR, NW, names[m], t[m],        \
               (NR + NW) * (double)bytes / t[m] / 1e3, NW * (double)bytes / t[m] / 1e3);                              \
    }
    CASE(3, 2) CASE(2, 1) CASE(1, 1) CASE(3, 0) CASE(0, 2)
  }
  // one CU's own store / load path: a few workgroups on an otherwise idle chip (4 waves each; 16 waves = one full CU's worth
  // when they land on one CU -- the dispatcher spreads workgroups, so read "per workgroup")
  for (int grid : {1, 2, 4, 8}) {
    const int r = 8192;
    const double tw = run<1, 0, 1>(din, dout, r, grid), tr = run<1, 1, 0>(din, dout, r, grid), tw8 = run<0, 0, 1>(din, dout, r, grid);
    printf("grid %d (chip idle)  write-only 16+8 B: %7.1f us = %5.1f GB/s per workgroup; 8 B pieces: %5.1f GB/s; read-only: %5.1f GB/s per workgroup\n", grid, tw,
           r * 1536.0 / tw / 1e3 / grid, r * 1536.0 / tw8 / 1e3 / grid, r * 1536.0 / tr / 1e3 / grid);
  }
  return 0;
}
