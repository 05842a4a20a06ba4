// gemm_args.h — argument block and epilogue ids of the bf16 MFMA GEMM (gemm_bf16.hip).
#pragma once
#include "czc_common.h"

namespace czc {

struct GemmArgs {
  const bf16_t* A; const bf16_t* B;
  // optional SECOND product in the same launch (EPI_F32 only): output rows [g_rows, M) are A2^T-side x B2 with leading
  // dimensions of their own and the same N, K; g_rows is a multiple of the row tile.  The encoder's backward uses it for
  // dW[qkv | o] = [dQKV | dY]^T . [x | ctx]: the two weight gradients are neighbours in the flat gradient array, and the
  // 768 x 768 one alone has 9 output tiles (28-way split-K, 37 K-tiles per workgroup: 745 TFLOP/s against 1 100-1 230)
  const bf16_t* A2; const bf16_t* B2; int lda2, ldb2, g_rows;
  void* C; void* C2;
  const float* bias; const bf16_t* aux;
  const float* rowv;     // per-row vector (head: lse)
  float* part;           // head partials [3][tilesN][M]
  float* colsum;         // optional [tiles_m * WM][N] per-wave-row partial column sums of C (DGELU epilogue)
  int M, N, K;
  int lda, ldb, ldc;
  int tiles_m, tiles_n;
  int group_n;           // column tiles per sweep of the tile order (swizzle_tile)
  int ktiles_per_split;  // K tiles handled by one split
  int fast_addr;         // both operands span < 2 GiB: DMA pieces may use 32-bit lane offsets from a scalar base
  int store_pol;         // cache policy of the bf16 output stores: 0 plain, 1 sc1, 2 nt
  int cfg;               // tile configuration: 0 = default, 1 = 256x256 (8 waves), 2 = 128x128 (4 waves, 2 WG/CU)
  int xcd8;              // 1 = the device dispatches workgroups round-robin over 8 XCDs (256 CUs): split-K launches are laid out by XCD
  float alpha;           // head: 1/T
  float beta;            // head: eps/(K+1)
  // EPI_FILTER (flat index, coarse bf16 scan): row m = query, column n = passage row of this chunk.  An entry is appended to
  // query m's candidate list when its upper bound acc + f_qe[m] * f_dn[n] exceeds the query's lower bound f_lb[m].
  const float* f_lb; const float* f_qe; const float* f_dn;
  float* f_cs; float* f_ce; int* f_ci; int* f_cnt;
  long long f_cap; int f_idbase; int f_nvalid;
};

#define EPI_STORE 0
#define EPI_BIAS 1
#define EPI_BIAS_GELU 2
#define EPI_DGELU 3
#define EPI_ADD 4
#define EPI_F32 5
#define EPI_LSE 6       // head forward: per-row (max, sumexp, sum) partials of alpha*acc
#define EPI_SOFTGRAD 7  // head backward: C = exp(alpha*acc - rowv[m]) - beta   (bf16)
#define EPI_BIAS_GELU1 8 // C = gelu(acc + bias) only (inference / key encoder: the pre-activation is not kept)
#define EPI_FILTER 9     // nothing is stored: accumulators that pass a per-row threshold are appended to per-row lists


// checks shapes, picks the instantiation and launches; returns a CZC_* code
int gemm_dispatch(int a_layout, int b_layout, int epi, GemmArgs a, int splitk, hipStream_t st);

}  // namespace czc
