// head.hip — MoCo InfoNCE head against the negative queue, for gfx950.
//
// Replaces src/moco.py:289-294 (q.queue logits), :323-330 (/T + cross_entropy, label smoothing),
// :340-351 (loss / accuracy / stdq / stdk) and :277-287 (ring-buffer enqueue).
//
// The queue lives slot-major in bf16, queue_bf16[K][C] (row j = key j).  Forward: the logits
// l_neg = q . queue^T are a bf16 MFMA GEMM (M = batch, N = K slots, reduction C) whose epilogue
// never writes the [batch, K] matrix: every 256-slot tile leaves per-row (max, sum exp, sum)
// partials and a tiny kernel folds them with the positive logit into lse, loss and accuracy.
// Backward: P = softmax - smoothing is produced tile by tile in bf16 by the same GEMM with a
// different epilogue (recomputing the logits from the saved lse), then dq = P . queue is a second
// GEMM that reads the SAME queue rows through the transposing LDS read (queue as a [k][n]
// operand), split over the K slots into fp32 slabs that one small kernel sums.
#include "czc_common.h"
#include "gemm_args.h"

namespace czc {

// q f32 [B,C] -> q_bf16 ; lpos[b] = sum_c q*k   (one wave per row)
__global__ __launch_bounds__(256) void head_prep_kernel(const float* __restrict__ q, const float* __restrict__ k, int B,
                                                        int C, bf16_t* __restrict__ qb, float* __restrict__ lpos) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  float s = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 a = *(const f32x4*)(q + (size_t)b * C + c), kk = *(const f32x4*)(k + (size_t)b * C + c);
    s += a[0] * kk[0] + a[1] * kk[1] + a[2] * kk[2] + a[3] * kk[3];
    uint2 w;
    w.x = pack2bf(a[0], a[1]);
    w.y = pack2bf(a[2], a[3]);
    *(uint2*)(qb + (size_t)b * C + c) = w;
  }
  s = wave_sum(s);
  if (lane == 0) lpos[b] = s;
}

// fold the per-tile partials with the positive logit.  One thread per row.
__global__ __launch_bounds__(256) void head_final_kernel(const float* __restrict__ part, int tiles, int B, int Kq,
                                                         const float* __restrict__ lpos, float invT, float ls,
                                                         float* __restrict__ lse_out, float* __restrict__ rowloss,
                                                         float* __restrict__ rowok) {
  // one wave per batch row: the lanes split the column tiles (a thread per row walked 2 x 513 dependent loads)
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const size_t pl = (size_t)tiles * B;
  float mx = -1e30f;
  for (int t = lane; t < tiles; t += 64) mx = fmaxf(mx, part[(size_t)t * B + b]);
  mx = wave_max(mx);
  float se = 0.f, sm = 0.f;
  for (int t = lane; t < tiles; t += 64) {
    se += part[pl + (size_t)t * B + b] * __expf(part[(size_t)t * B + b] - mx);
    sm += part[2 * pl + (size_t)t * B + b];
  }
  se = wave_sum(se);
  sm = wave_sum(sm);
  if (lane != 0) return;
  // the positive logit as ONE rounded fp32 value, as the reference's `logits / T` tensor holds it: never fused into the
  // subtractions below.  With a bare `lpos * invT` the compiler contracted `z0 - M` (and `z0 - lse` in the backward) into
  // fma(lpos, invT, -M), i.e. subtracted from the UNROUNDED product: when the positive dominates (lse == z0; every early
  // step with the README flags) exp(z0 - M) came out as exp(rounding residual) instead of 1, loss and coefficient as
  // +-1e-4 instead of exactly 0, and AdamW -- which normalises any non-zero gradient to a full-size step -- moved
  // every weight by lr along that noise.
  float z0 = lpos[b] * invT;
  asm volatile("" : "+v"(z0));   // (__fmul_rn is a plain `*` in this toolchain's headers: the empty asm is what keeps the product rounded)
  const float M = fmaxf(mx, z0);
  const float tot = se * __expf(mx - M) + __expf(z0 - M);
  const float lse = M + __logf(tot);
  lse_out[b] = lse;
  rowloss[b] = (1.0f - ls) * (lse - z0) + ls * (lse - (sm + z0) / (float)(Kq + 1));
  rowok[b] = (z0 >= mx) ? 1.0f : 0.f;   // argmax picks column 0 on ties, as torch.argmax does
}

// stats[0] = mean loss, stats[1] = accuracy %, stats[2] = stats[3] = 0 (filled by colstd)
__global__ __launch_bounds__(256) void head_mean_kernel(const float* __restrict__ rowloss, const float* __restrict__ rowok,
                                                        int B, float* __restrict__ stats) {
  __shared__ float red[8];
  float a = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) { a += rowloss[i]; c += rowok[i]; }
  a = wave_sum(a); c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[0] = (red[0] + red[1] + red[2] + red[3]) / (float)B;
    stats[1] = ((red[4] + red[5] + red[6] + red[7]) / (float)B) * 100.0f;   // mean first, then percent: the reference's rounding
    stats[2] = 0.f; stats[3] = 0.f; stats[4] = 0.f; stats[5] = 0.f; stats[6] = 0.f; stats[7] = 0.f;
  }
}

// out[0] += mean_c std_b(x[:, c])  (unbiased, torch.std(x, dim=0).mean())
__global__ __launch_bounds__(256) void colstd_kernel(const float* __restrict__ x, int B, int C, float* out) {
  // 64 columns per workgroup, the four waves split the rows; two passes (mean, then squared deviations) as torch.std
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  // sixteen rows' loads in flight per thread (a rolled loop pays one memory round trip per row: 28 us for 0.8 MB)
  float m = 0.f;
  if (c < C) {
    int b = w;
    for (; b + 60 < B; b += 64) {
      float t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = x[(size_t)(b + 4 * i) * C + c];
#pragma unroll
      for (int i = 0; i < 16; ++i) m += t[i];
    }
    for (; b < B; b += 4) m += x[(size_t)b * C + c];
  }
  red[w][lane] = m;
  __syncthreads();
  m = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)B;
  __syncthreads();
  float v = 0.f;
  if (c < C) {
    int b = w;
    for (; b + 60 < B; b += 64) {
      float t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = x[(size_t)(b + 4 * i) * C + c];
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float d = t[i] - m; v += d * d; }
    }
    for (; b < B; b += 4) { const float d = x[(size_t)b * C + c] - m; v += d * d; }
  }
  red[w][lane] = v;
  __syncthreads();
  if (w == 0) {
    v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float sd = (c < C) ? sqrtf(v / (float)(B - 1)) : 0.f;
    sd = wave_sum(sd);
    if (lane == 0) atomicAdd(out, sd / (float)C);
  }
}

// dq = scale * (sum_z slab[z] + coef0[b] * k)
__global__ __launch_bounds__(256) void head_bwd_final_kernel(const float* __restrict__ slabs, int nslabs, int B, int C,
                                                             const float* __restrict__ k, const float* __restrict__ lpos,
                                                             const float* __restrict__ lse, float invT, float ls, int Kq,
                                                             float scale, float* __restrict__ dq) {
  const int64_t n4 = (int64_t)B * C / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int b = (int)((i * 4) / C);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < nslabs; ++z) s += ((const f32x4*)slabs)[(int64_t)z * n4 + i];
  float z0 = lpos[b] * invT;   // the same rounded value head_final_kernel folded into lse (see there)
  asm volatile("" : "+v"(z0));
  const float coef0 = __expf(z0 - lse[b]) - (1.0f - ls) - ls / (float)(Kq + 1);
  ((f32x4*)dq)[i] = (s + ((const f32x4*)k)[i] * coef0) * scale;
}

// ---- enqueue: 64 keys x 64 columns per block, transposed through LDS for the [C][K] fp32 copy ----
__global__ __launch_bounds__(256) void enqueue_kernel(const float* __restrict__ keys, int n, int C, int Kq,
                                                      bf16_t* __restrict__ qb, float* __restrict__ qf,
                                                      const int64_t* __restrict__ ptr_dev) {
  __shared__ float tile[64][65];
  const int i0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int64_t ptr = ptr_dev[0];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int i = i0 + r, c = c0 + tx;
    float v = 0.f;
    if (i < n && c < C) {
      v = keys[(size_t)i * C + c];
      const int64_t slot = (ptr + i) % Kq;
      qb[(size_t)slot * C + c] = f2bf(v);   // 64 lanes -> 128 contiguous bytes of one queue row
    }
    tile[r][tx] = v;
  }
  if (!qf) return;
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {       // r = column, tx = key: 64 lanes -> 256 contiguous bytes of queue_f32[c][.]
    const int c = c0 + r, i = i0 + tx;
    if (i < n && c < C) {
      const int64_t slot = (ptr + i) % Kq;
      qf[(size_t)c * Kq + slot] = tile[tx][r];
    }
  }
}
__global__ void advance_ptr_kernel(int64_t* ptr, int n, int Kq) { ptr[0] = (ptr[0] + n) % Kq; }

// queue_bf16[K][C] = transpose(queue_f32[C][K])
__global__ __launch_bounds__(256) void queue_import_kernel(const float* __restrict__ qf, int C, int Kq, bf16_t* __restrict__ qb) {
  __shared__ float tile[64][65];
  const int j0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, j = j0 + tx;
    tile[r][tx] = (c < C && j < Kq) ? qf[(size_t)c * Kq + j] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int j = j0 + r, c = c0 + tx;
    if (j < Kq && c < C) qb[(size_t)j * C + c] = f2bf(tile[tx][r]);
  }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct HeadWs {
  int64_t qb, lpos, part, rowloss, rowok, P, slabs, total;
  int tiles, splits;
};
static HeadWs head_ws(int B, int C, int Kq) {
  HeadWs w;
  w.tiles = ceil_div(Kq, 256);
  int kt = ceil_div(Kq, 64);
  w.splits = kt < 84 ? kt : 84;
  int64_t o = 0;
  w.qb = o; o += align256((int64_t)B * C * 2);
  w.lpos = o; o += align256((int64_t)B * 4);
  w.part = o; o += align256((int64_t)3 * w.tiles * B * 4);
  w.rowloss = o; o += align256((int64_t)B * 4);
  w.rowok = o; o += align256((int64_t)B * 4);
  w.P = o; o += align256((int64_t)B * Kq * 2);
  w.slabs = o; o += align256((int64_t)w.splits * B * C * 4);
  w.total = o;
  return w;
}


// ---------------------------------------------------------------------------------------------------------------------
// In-batch head (reference src/inbatch.py:71-75 + its autograd): scores = (q / T) . keys^T over the keys of ALL ranks
// [N = W*B, C], cross-entropy with the positive of local query i at column label_offset + i, label smoothing, and the
// gradients w.r.t. BOTH operands (queries and keys are differentiable there).  B <= a few hundred, N <= a few thousand:
// 0.8 GFLOP forward, fp32 FMA throughout (what the reference's fp32 matmul gives), every sum in a fixed order.
// ---------------------------------------------------------------------------------------------------------------------
// 16 x 16 tile of scores per block: s[i][j] = invT * <q_i, k_j>
__global__ __launch_bounds__(256) void inbatch_scores_kernel(const float* __restrict__ q, const float* __restrict__ k, int B, int N,
                                                             int C, float invT, float* __restrict__ s) {
  __shared__ float a[16][17], b[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  float acc = 0.f;
  for (int d0 = 0; d0 < C; d0 += 16) {
    const int d = d0 + tx;
    a[ty][tx] = (i0 + ty < B && d < C) ? q[(int64_t)(i0 + ty) * C + d] * invT : 0.f;   // (q / T) first, as the reference
    b[ty][tx] = (j0 + ty < N && d < C) ? k[(int64_t)(j0 + ty) * C + d] : 0.f;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = fmaf(a[ty][kk], b[tx][kk], acc);
    __syncthreads();
  }
  if (i0 + ty < B && j0 + tx < N) s[(int64_t)(i0 + ty) * N + j0 + tx] = acc;
}

// one block per query row: log-sum-exp, loss, first arg-max, and the row of d loss / d scores written over the scores:
// g_j = (p_j - (1 - ls) [j = y] - ls / N) / B      (F.cross_entropy with label_smoothing, mean over the batch)
__global__ __launch_bounds__(256) void inbatch_row_kernel(float* __restrict__ s, int B, int N, int label_offset, float ls,
                                                          float* __restrict__ rowloss, float* __restrict__ rowok) {
  __shared__ float redf[4];
  __shared__ int redi[4];
  const int i = blockIdx.x, tid = threadIdx.x, y = label_offset + i;
  float* row = s + (int64_t)i * N;
  float mx = -INFINITY, sm = 0.f;
  int am = 0x7fffffff;
  for (int j = tid; j < N; j += 256) {
    const float z = row[j];
    if (z > mx) { mx = z; am = j; }     // first index of the maximum in this thread's strided run
    sm += z;
  }
  // block maximum, then the smallest index that attains it (torch.argmax returns the first)
  float wm = xor32_max(xor16_max(row_max16(mx)));
  if ((tid & 63) == 0) redf[tid >> 6] = wm;
  __syncthreads();
  const float bm = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  __syncthreads();
  int cand = (mx == bm) ? am : 0x7fffffff;
  for (int o = 32; o >= 1; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
  float se = 0.f;
  for (int j = tid; j < N; j += 256) se += __expf(row[j] - bm);
  se = wave_sum(se);
  sm = wave_sum(sm);
  if ((tid & 63) == 0) { redi[tid >> 6] = cand; redf[tid >> 6] = se; }
  __syncthreads();
  const int first = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
  const float sumexp = redf[0] + redf[1] + redf[2] + redf[3];
  __syncthreads();
  if ((tid & 63) == 0) redf[tid >> 6] = sm;
  __syncthreads();
  const float sumz = redf[0] + redf[1] + redf[2] + redf[3];
  const float lse = bm + __logf(sumexp);
  if (tid == 0) {
    const float zy = row[y];
    rowloss[i] = (1.0f - ls) * (lse - zy) + ls * (lse - sumz / (float)N);
    rowok[i] = (first == y) ? 1.0f : 0.f;
  }
  __syncthreads();   // row[y] has been read before its owner overwrites it below
  const float invB = 1.0f / (float)B, sm_eps = ls / (float)N;
  for (int j = tid; j < N; j += 256) {
    const float p = __expf(row[j] - lse);
    row[j] = (p - (j == y ? 1.0f - ls : 0.f) - sm_eps) * invB;
  }
}

// out_r = coef * sum_j w[r][j] x_j  (rows of `w` contiguous: d q_i = invT sum_j g[i][j] k_j), one block per output row
__global__ __launch_bounds__(256) void inbatch_rowmix_kernel(const float* __restrict__ w, int ldw, const float* __restrict__ x, int n,
                                                             int C, float coef, float* __restrict__ out) {
  const int r = blockIdx.x, tid = threadIdx.x;
  constexpr int MAXD = 16;                 // C <= 4096
  float acc[MAXD];
  const int nd = (C + 255) / 256;
#pragma unroll
  for (int u = 0; u < MAXD; ++u) acc[u] = 0.f;
  for (int j = 0; j < n; ++j) {
    const float wj = w[(int64_t)r * ldw + j];
#pragma unroll
    for (int u = 0; u < MAXD; ++u)
      if (u < nd) { const int d = tid + 256 * u; if (d < C) acc[u] = fmaf(wj, x[(int64_t)j * C + d], acc[u]); }
  }
#pragma unroll
  for (int u = 0; u < MAXD; ++u)
    if (u < nd) { const int d = tid + 256 * u; if (d < C) out[(int64_t)r * C + d] = coef * acc[u]; }
}
// out_c = coef * sum_i w[i][c] x_i  (a COLUMN of `w`: d k_j = invT sum_i g[i][j] q_i), four output rows per block so that the
// x rows are read once for the four
__global__ __launch_bounds__(256) void inbatch_colmix_kernel(const float* __restrict__ w, int ldw, int ncols, const float* __restrict__ x,
                                                             int n, int C, float coef, float* __restrict__ out) {
  const int c0 = blockIdx.x * 4, tid = threadIdx.x;
  constexpr int MAXD = 16;
  const int nd = (C + 255) / 256;
  float acc[4][MAXD];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int u = 0; u < MAXD; ++u) acc[r][u] = 0.f;
  for (int i = 0; i < n; ++i) {
    float wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wv[r] = (c0 + r < ncols) ? w[(int64_t)i * ldw + c0 + r] : 0.f;
#pragma unroll
    for (int u = 0; u < MAXD; ++u)
      if (u < nd) {
        const int d = tid + 256 * u;
        if (d < C) {
          const float xv = x[(int64_t)i * C + d];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r][u] = fmaf(wv[r], xv, acc[r][u]);
        }
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (c0 + r < ncols)
#pragma unroll
      for (int u = 0; u < MAXD; ++u)
        if (u < nd) { const int d = tid + 256 * u; if (d < C) out[(int64_t)(c0 + r) * C + d] = coef * acc[r][u]; }
}

}  // namespace czc

using namespace czc;

extern "C" int64_t czc_infonce_workspace_bytes(int batch, int dim, int queue_size) {
  return head_ws(batch, dim, queue_size).total;
}

static int head_check(const char* who, int B, int C, int Kq, float T, float ls, void* ws, int64_t ws_bytes) {
  CZC_REQUIRE(B > 1 && C > 0 && (C % 8) == 0 && Kq > 0 && (Kq % 8) == 0, "%s: batch=%d dim=%d queue=%d (dim, queue multiples of 8; batch >= 2)", who, B, C, Kq);
  CZC_REQUIRE(T > 0.f && ls >= 0.f && ls < 1.f, "%s: temperature=%f label_smoothing=%f", who, T, ls);
  CZC_REQUIRE(ws && ws_bytes >= head_ws(B, C, Kq).total && ((uintptr_t)ws % 256) == 0, "%s: workspace too small or unaligned (%lld < %lld)",
              who, (long long)ws_bytes, (long long)head_ws(B, C, Kq).total);
  return CZC_OK;
}

extern "C" int czc_infonce_forward(const float* q, const float* k, const uint16_t* queue_bf16, int batch, int dim,
                                   int queue_size, float temperature, float label_smoothing, void* workspace,
                                   int64_t workspace_bytes, float* lse_out, float* stats_out, czc_stream_t stream) {
  if (int rc = head_check("czc_infonce_forward", batch, dim, queue_size, temperature, label_smoothing, workspace, workspace_bytes)) return rc;
  ProfScope prof((hipStream_t)stream, "infonce_fwd", 2.0 * batch * (double)dim * (queue_size + 1), 2.0 * queue_size * (double)dim);
  hipStream_t st = (hipStream_t)stream;
  const HeadWs w = head_ws(batch, dim, queue_size);
  char* base = (char*)workspace;
  bf16_t* qb = (bf16_t*)(base + w.qb);
  float* lpos = (float*)(base + w.lpos);
  float* part = (float*)(base + w.part);
  float* rowloss = (float*)(base + w.rowloss);
  float* rowok = (float*)(base + w.rowok);
  const float invT = 1.0f / temperature;
  hipLaunchKernelGGL(head_prep_kernel, dim3(ceil_div(batch, 4)), dim3(256), 0, st, q, k, batch, dim, qb, lpos);
  GemmArgs a{};
  a.A = qb; a.B = queue_bf16; a.M = batch; a.N = queue_size; a.K = dim;
  a.lda = dim; a.ldb = dim; a.ldc = queue_size; a.C = part; a.part = part; a.alpha = invT;
  if (int rc = gemm_dispatch(0, 0, EPI_LSE, a, 1, st)) return rc;
  hipLaunchKernelGGL(head_final_kernel, dim3(ceil_div(batch, 4)), dim3(256), 0, st, part, w.tiles, batch, queue_size,
                     lpos, invT, label_smoothing, lse_out, rowloss, rowok);
  hipLaunchKernelGGL(head_mean_kernel, dim3(1), dim3(256), 0, st, rowloss, rowok, batch, stats_out);
  hipLaunchKernelGGL(colstd_kernel, dim3(ceil_div(dim, 64)), dim3(256), 0, st, q, batch, dim, stats_out + 2);
  hipLaunchKernelGGL(colstd_kernel, dim3(ceil_div(dim, 64)), dim3(256), 0, st, k, batch, dim, stats_out + 3);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_infonce_backward(const float* q, const float* k, const uint16_t* queue_bf16, int batch, int dim,
                                    int queue_size, float temperature, float label_smoothing, const float* lse,
                                    float loss_scale, void* workspace, int64_t workspace_bytes, float* dq,
                                    czc_stream_t stream) {
  if (int rc = head_check("czc_infonce_backward", batch, dim, queue_size, temperature, label_smoothing, workspace, workspace_bytes)) return rc;
  CZC_REQUIRE(((int64_t)batch * dim) % 4 == 0, "czc_infonce_backward: batch*dim must be a multiple of 4");
  ProfScope prof((hipStream_t)stream, "infonce_bwd", 4.0 * batch * (double)dim * queue_size, 4.0 * queue_size * (double)dim + 4.0 * batch * (double)queue_size);
  hipStream_t st = (hipStream_t)stream;
  const HeadWs w = head_ws(batch, dim, queue_size);
  char* base = (char*)workspace;
  bf16_t* qb = (bf16_t*)(base + w.qb);
  float* lpos = (float*)(base + w.lpos);
  bf16_t* P = (bf16_t*)(base + w.P);
  float* slabs = (float*)(base + w.slabs);
  const float invT = 1.0f / temperature;
  hipLaunchKernelGGL(head_prep_kernel, dim3(ceil_div(batch, 4)), dim3(256), 0, st, q, k, batch, dim, qb, lpos);
  GemmArgs a{};
  a.A = qb; a.B = queue_bf16; a.M = batch; a.N = queue_size; a.K = dim;
  a.lda = dim; a.ldb = dim; a.ldc = queue_size; a.C = P; a.rowv = lse; a.alpha = invT;
  a.beta = label_smoothing / (float)(queue_size + 1);
  if (int rc = gemm_dispatch(0, 0, EPI_SOFTGRAD, a, 1, st)) return rc;
  GemmArgs g{};
  g.A = P; g.B = queue_bf16; g.M = batch; g.N = dim; g.K = queue_size;
  g.lda = queue_size; g.ldb = dim; g.ldc = dim; g.C = slabs;
  if (int rc = gemm_dispatch(0, 1, EPI_F32, g, w.splits, st)) return rc;
  const int64_t n4 = (int64_t)batch * dim / 4;
  hipLaunchKernelGGL(head_bwd_final_kernel, dim3((int)((n4 + 255) / 256)), dim3(256), 0, st, slabs, w.splits, batch, dim,
                     k, lpos, lse, invT, label_smoothing, queue_size, loss_scale * invT / (float)batch, dq);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_enqueue(const float* keys, int n, int dim, int queue_size, uint16_t* queue_bf16, float* queue_f32,
                           int64_t* queue_ptr, czc_stream_t stream) {
  CZC_REQUIRE(n > 0 && dim > 0 && queue_size > 0 && n <= queue_size, "czc_enqueue: n=%d dim=%d queue=%d", n, dim, queue_size);
  CZC_REQUIRE(queue_size % n == 0, "czc_enqueue: %d, %d (queue_size must be divisible by the gathered batch)", n, queue_size);
  ProfScope prof((hipStream_t)stream, "enqueue", 0.0, (double)n * dim * (4.0 + 2.0 + (queue_f32 ? 4.0 : 0.0)));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(enqueue_kernel, dim3(ceil_div(n, 64), ceil_div(dim, 64)), dim3(256), 0, st, keys, n, dim,
                     queue_size, queue_bf16, queue_f32, queue_ptr);
  hipLaunchKernelGGL(advance_ptr_kernel, dim3(1), dim3(1), 0, st, queue_ptr, n, queue_size);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_queue_import(const float* queue_f32, int dim, int queue_size, uint16_t* queue_bf16,
                                czc_stream_t stream) {
  CZC_REQUIRE(dim > 0 && queue_size > 0, "czc_queue_import: dim=%d queue=%d", dim, queue_size);
  hipLaunchKernelGGL(queue_import_kernel, dim3(ceil_div(queue_size, 64), ceil_div(dim, 64)), dim3(256), 0,
                     (hipStream_t)stream, queue_f32, dim, queue_size, queue_bf16);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int64_t czc_inbatch_workspace_floats(int batch, int n_keys) {
  if (batch < 1 || n_keys < 1) return -1;
  return (int64_t)batch * n_keys + 2 * (int64_t)batch;
}

extern "C" int czc_inbatch_head(const float* q, const float* keys, int batch, int n_keys, int dim, int label_offset,
                                float temperature, float label_smoothing, float* workspace, int64_t workspace_floats,
                                float* stats_out, float* dq, float* dkeys, czc_stream_t stream) {
  CZC_REQUIRE(q && keys && workspace && stats_out, "czc_inbatch_head: null pointer");
  CZC_REQUIRE(batch >= 2 && n_keys >= batch && dim >= 1 && dim <= 4096, "czc_inbatch_head: batch=%d (>= 2) n_keys=%d dim=%d (<= 4096)",
              batch, n_keys, dim);
  CZC_REQUIRE(label_offset >= 0 && label_offset + batch <= n_keys, "czc_inbatch_head: labels [%d, %d) outside the %d keys", label_offset,
              label_offset + batch, n_keys);
  CZC_REQUIRE(temperature > 0.f && label_smoothing >= 0.f && label_smoothing < 1.f, "czc_inbatch_head: temperature=%f label_smoothing=%f",
              temperature, label_smoothing);
  CZC_REQUIRE(workspace_floats >= czc_inbatch_workspace_floats(batch, n_keys), "czc_inbatch_head: workspace %lld < %lld floats",
              (long long)workspace_floats, (long long)czc_inbatch_workspace_floats(batch, n_keys));
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(st, "inbatch_head", 2.0 * batch * (double)n_keys * dim * ((dq ? 1 : 0) + (dkeys ? 1 : 0) + 1), 0.0);
  const float invT = 1.0f / temperature;
  float* s = workspace;
  float* rowloss = s + (int64_t)batch * n_keys;
  float* rowok = rowloss + batch;
  hipLaunchKernelGGL(inbatch_scores_kernel, dim3(ceil_div(n_keys, 16), ceil_div(batch, 16)), dim3(256), 0, st, q, keys, batch, n_keys, dim,
                     invT, s);
  hipLaunchKernelGGL(inbatch_row_kernel, dim3(batch), dim3(256), 0, st, s, batch, n_keys, label_offset, label_smoothing, rowloss, rowok);
  hipLaunchKernelGGL(head_mean_kernel, dim3(1), dim3(256), 0, st, rowloss, rowok, batch, stats_out);
  // the reference's stdq / stdk: torch.std(qemb, dim=0).mean() and the same over THIS rank's keys (src/inbatch.py:84-85)
  hipLaunchKernelGGL(colstd_kernel, dim3(ceil_div(dim, 64)), dim3(256), 0, st, q, batch, dim, stats_out + 2);
  hipLaunchKernelGGL(colstd_kernel, dim3(ceil_div(dim, 64)), dim3(256), 0, st, keys + (int64_t)label_offset * dim, batch, dim, stats_out + 3);
  if (dq) hipLaunchKernelGGL(inbatch_rowmix_kernel, dim3(batch), dim3(256), 0, st, s, n_keys, keys, n_keys, dim, invT, dq);
  if (dkeys) hipLaunchKernelGGL(inbatch_colmix_kernel, dim3(ceil_div(n_keys, 4)), dim3(256), 0, st, s, n_keys, n_keys, q, batch, dim, invT, dkeys);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}
