// Dev check of the czc_common.h wave reductions against lane-by-lane sums (run on the GPU box).
#include "../czech_contriever_amd/csrc/czc_common.h"
#include <stdio.h>
using namespace czc;
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float v = (float)(l * l % 37) + 0.25f * l;
  out[l] = wave_sum(v);
  out[64 + l] = half_sum(v);
  out[128 + l] = row_sum16(v);
  out[192 + l] = wave_max(v);
  float a, b; xor16_pair(v, a, b); out[256 + l] = a; out[320 + l] = b;
  xor32_pair(v, a, b); out[384 + l] = a; out[448 + l] = b;
  out[512 + l] = dpp_mov<CZC_DPP_HALF_MIRROR>(v);
  out[576 + l] = dpp_mov<CZC_DPP_MIRROR>(v);
  out[640 + l] = dpp_mov<0x128>(v);
}
int main() {
  float* d; hipMalloc(&d, 704 * 4);
  k<<<1, 64>>>(d);
  float h[704]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  float v[64]; for (int l = 0; l < 64; ++l) v[l] = (float)(l * l % 37) + 0.25f * l;
  float tot = 0, mx = -1e9; for (int l = 0; l < 64; ++l) { tot += v[l]; mx = v[l] > mx ? v[l] : mx; }
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    float hs = 0, rs = 0;
    for (int j = 0; j < 64; ++j) { if ((j >> 5) == (l >> 5)) hs += v[j]; if ((j >> 4) == (l >> 4)) rs += v[j]; }
    if (h[l] != tot || h[64 + l] != hs || h[128 + l] != rs || h[192 + l] != mx) { ++bad; if (bad < 6) printf("lane %d: wave %g/%g half %g/%g row %g/%g max %g/%g\n", l, h[l], tot, h[64 + l], hs, h[128 + l], rs, h[192 + l], mx); }
  }
  printf("xor16 a:"); for (int l = 0; l < 64; l += 8) printf(" %g(%g)", h[256 + l], v[l]); printf("\nxor16 b:"); for (int l = 0; l < 64; l += 8) printf(" %g(%g)", h[320 + l], v[l ^ 16]);
  printf("\nxor32 a:"); for (int l = 0; l < 64; l += 8) printf(" %g", h[384 + l]); printf("\nxor32 b:"); for (int l = 0; l < 64; l += 8) printf(" %g", h[448 + l]);
  printf("\nhalf_mirror:"); for (int l = 0; l < 16; ++l) printf(" %g(%g)", h[512 + l], v[(l & 8) | (7 - (l & 7))]);
  printf("\nmirror:"); for (int l = 0; l < 16; ++l) printf(" %g(%g)", h[576 + l], v[15 - l]);
  printf("\nror8:"); for (int l = 0; l < 16; ++l) printf(" %g(%g)", h[640 + l], v[l ^ 8]);
  printf("\nbad=%d\n", bad);
  return bad != 0;
}
