// distill.hip — the similarity-matrix distillation loss of the teacher/student exchange (gfx950).
//
// Ajchler/czech-contriever distill.py:44-47 (compute_sim_matrix: L2-normalise rows, E.E^T) and :270-274
// (teacher_sim / student_sim / F.mse_loss):   loss = mean_ij (Sn_i.Sn_j - Tn_i.Tn_j)^2
// with Sn, Tn the row-normalised student [B, D] and teacher [B, Dt] embeddings.  One call gives the loss and its
// gradient w.r.t. the (un-normalised) student embeddings; the caller scales both (DISTILL_LOSS_MULTIPLIER, weight).
//   G = Sn Sn^T - Tn Tn^T (symmetric)      dL/dSn = (4 / B^2) G Sn      dL/dS_i = (dSn_i - Sn_i (Sn_i.dSn_i)) / |S_i|
// fp32 FMA throughout (B = 256, D = 768, Dt = 3584: 0.5 GFLOP, microseconds): the sums are deterministic (no atomics).
#include "czc_common.h"

namespace czc {

__global__ __launch_bounds__(256) void distill_normalize_kernel(const float* __restrict__ x, int dim, float* __restrict__ xn,
                                                                float* __restrict__ inv_norm) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (int64_t)row * dim;
  float ss = 0.f;
  for (int d = tid; d < dim; d += 256) ss += xr[d] * xr[d];
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float inv = 1.0f / fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);   // F.normalize: x / max(|x|, eps)
  for (int d = tid; d < dim; d += 256) xn[(int64_t)row * dim + d] = xr[d] * inv;
  if (inv_norm && tid == 0) inv_norm[row] = inv;
}

// G tile 16x16 per block: G[i][j] = <Sn_i, Sn_j> - <Tn_i, Tn_j>; partial[block] = sum of the tile's G^2
__global__ __launch_bounds__(256) void distill_simdiff_kernel(const float* __restrict__ sn, int ds, const float* __restrict__ tn,
                                                              int dt, int batch, float* __restrict__ g,
                                                              float* __restrict__ partial) {
  __shared__ float a[16][17], b[16][17];
  __shared__ float red[4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  float acc = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const float* e = pass ? tn : sn;
    const int dim = pass ? dt : ds;
    float s = 0.f;
    for (int d0 = 0; d0 < dim; d0 += 16) {
      const int d = d0 + tx;
      a[ty][tx] = (i0 + ty < batch && d < dim) ? e[(int64_t)(i0 + ty) * dim + d] : 0.f;
      b[ty][tx] = (j0 + ty < batch && d < dim) ? e[(int64_t)(j0 + ty) * dim + d] : 0.f;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; ++k) s = fmaf(a[ty][k], b[tx][k], s);
      __syncthreads();
    }
    acc += pass ? -s : s;
  }
  const bool in = (i0 + ty < batch) && (j0 + tx < batch);
  if (in) g[(int64_t)(i0 + ty) * batch + j0 + tx] = acc;
  float sq = wave_sum(in ? acc * acc : 0.f);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one block per row i: dSn_i = (4/B^2) sum_j G[i][j] Sn_j, projected through the normalisation; block 0 also sums the loss
__global__ __launch_bounds__(256) void distill_grad_kernel(const float* __restrict__ g, const float* __restrict__ sn,
                                                           const float* __restrict__ inv_norm, int batch, int dim,
                                                           const float* __restrict__ partial, int n_partial,
                                                           float* __restrict__ loss, float* __restrict__ d_student) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  if (i == 0 && tid == 0) {
    float s = 0.f;
    for (int p = 0; p < n_partial; ++p) s += partial[p];
    loss[0] = s / ((float)batch * (float)batch);
  }
  if (!d_student) return;
  constexpr int MAXD = 16;                 // dim <= 4096
  float acc[MAXD];
  const int nd = (dim + 255) / 256;
#pragma unroll
  for (int u = 0; u < MAXD; ++u) acc[u] = 0.f;
  for (int j = 0; j < batch; ++j) {
    const float w = g[(int64_t)i * batch + j];
#pragma unroll
    for (int u = 0; u < MAXD; ++u)
      if (u < nd) { const int d = tid + 256 * u; if (d < dim) acc[u] = fmaf(w, sn[(int64_t)j * dim + d], acc[u]); }
  }
  float dot = 0.f;
#pragma unroll
  for (int u = 0; u < MAXD; ++u)
    if (u < nd) { const int d = tid + 256 * u; if (d < dim) dot = fmaf(sn[(int64_t)i * dim + d], acc[u], dot); }
  dot = wave_sum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  dot = red[0] + red[1] + red[2] + red[3];
  const float coef = 4.0f / ((float)batch * (float)batch) * inv_norm[i];
#pragma unroll
  for (int u = 0; u < MAXD; ++u)
    if (u < nd) {
      const int d = tid + 256 * u;
      if (d < dim) d_student[(int64_t)i * dim + d] = coef * (acc[u] - sn[(int64_t)i * dim + d] * dot);
    }
}

static int64_t distill_floats(int b, int ds, int dt) {
  const int64_t tiles = (int64_t)ceil_div(b, 16) * ceil_div(b, 16);
  return (int64_t)b * ds + (int64_t)b * dt + b + (int64_t)b * b + tiles;
}

}  // namespace czc

using namespace czc;

extern "C" int64_t czc_distill_workspace_floats(int32_t batch, int32_t dim_student, int32_t dim_teacher) {
  if (batch < 1 || dim_student < 1 || dim_teacher < 1) return -1;
  return distill_floats(batch, dim_student, dim_teacher);
}

extern "C" int czc_distill_sim_loss(const float* student, const float* teacher, int32_t batch, int32_t dim_student,
                                    int32_t dim_teacher, float* workspace, int64_t workspace_floats, float* loss,
                                    float* d_student, czc_stream_t stream) {
  CZC_REQUIRE(student && teacher && workspace && loss, "czc_distill_sim_loss: null pointer");
  CZC_REQUIRE(batch >= 1 && dim_student >= 1 && dim_student <= 4096 && dim_teacher >= 1,
              "czc_distill_sim_loss: batch=%d dim_student=%d (<= 4096) dim_teacher=%d", batch, dim_student, dim_teacher);
  CZC_REQUIRE(workspace_floats >= distill_floats(batch, dim_student, dim_teacher), "czc_distill_sim_loss: workspace %lld < %lld floats",
              (long long)workspace_floats, (long long)distill_floats(batch, dim_student, dim_teacher));
  hipStream_t st = (hipStream_t)stream;
  float* sn = workspace;
  float* tn = sn + (int64_t)batch * dim_student;
  float* inv = tn + (int64_t)batch * dim_teacher;
  float* g = inv + batch;
  float* partial = g + (int64_t)batch * batch;
  const int tiles = ceil_div(batch, 16);
  hipLaunchKernelGGL(distill_normalize_kernel, dim3(batch), dim3(256), 0, st, student, dim_student, sn, inv);
  hipLaunchKernelGGL(distill_normalize_kernel, dim3(batch), dim3(256), 0, st, teacher, dim_teacher, tn, (float*)nullptr);
  hipLaunchKernelGGL(distill_simdiff_kernel, dim3(tiles, tiles), dim3(256), 0, st, sn, dim_student, tn, dim_teacher, batch, g, partial);
  hipLaunchKernelGGL(distill_grad_kernel, dim3(batch), dim3(256), 0, st, g, sn, inv, batch, dim_student, partial, tiles * tiles, loss, d_student);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}
