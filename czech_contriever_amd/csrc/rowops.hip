// rowops.hip — HBM-bound row kernels of the BERT encoder for gfx950:
//   embedding gather + LayerNorm (+dropout) fwd/bwd, dropout + residual + LayerNorm fwd/bwd,
//   masked mean / CLS pooling fwd/bwd, bf16 column sums (bias gradients), dropout-mask export.
// One 64-lane wave owns one row at a time: lane l holds the float4 chunks l, l+64, l+128, l+192 of
// the row (hidden <= 1024), so a row's mean / variance are two wave reductions, every global access
// is 8 B (bf16) or 16 B (f32) per lane, contiguous across the wave, and the per-column gradient
// sums (d gamma, d beta, d bias) stay in the SAME lane's registers across all rows a wave visits.
#include "czc_common.h"

namespace czc {

// NIT = float4 chunks per lane: hidden <= 64 lanes * NIT * 4 elements (NIT = 3 for 768, 4 up to 1024)

__device__ __forceinline__ f32x4 ld_bf4(const bf16_t* p) {
  const uint2 w = *(const uint2*)p;
  return (f32x4){bflo(w.x), bfhi(w.x), bflo(w.y), bfhi(w.y)};
}
// 4 bf16 at column c of a row of H, zeros past the end.  The ADDRESS is clamped and the load is unconditional: a load
// under `if (c < H)` is waited for before the next one is issued (one memory round trip per chunk and tensor),
// clamped loads of a row are all in flight together.
// NT: a streaming (nt) read.  The LayerNorm kernels read every input row exactly once; without those rows in its way the
// output rows stay in the 256 MiB Infinity Cache for the GEMM that reads them next (interleaved in the step: QKV 222.4 ->
// 216.2 us, FFN-up 387.4 -> 381.8, the LayerNorm backward itself 107.7 -> 105.7, the forward 51.4 -> 52.0)
template <bool NT = false>
__device__ __forceinline__ f32x4 ld_row4(const bf16_t* row, int c, int H) {
  const bool ok = c < H;
  typedef __attribute__((ext_vector_type(2))) unsigned u2v;
  uint2 w;
  if (NT) { const u2v t = __builtin_nontemporal_load((const u2v*)(row + (ok ? c : 0))); w = make_uint2(t[0], t[1]); }
  else w = *(const uint2*)(row + (ok ? c : 0));
  const f32x4 v = {bflo(w.x), bfhi(w.x), bflo(w.y), bfhi(w.y)};
  return ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 ld_vec4(const float* vec, int c, int H) {
  const bool ok = c < H;
  const f32x4 v = *(const f32x4*)(vec + (ok ? c : 0));
  return ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void st_bf4(bf16_t* p, f32x4 v) {
  uint2 w;
  w.x = pack2bf(v[0], v[1]);
  w.y = pack2bf(v[2], v[3]);
  *(uint2*)p = w;
}
// keep-scale for the 4 consecutive elements e .. e+3 (e a multiple of 4: one RNG quad)
__device__ __forceinline__ f32x4 drop_scale4(uint32_t seed, uint32_t site, uint32_t e, uint32_t thr, float inv_keep) {
  const czc_quad q = czc_rng_quad(czc_rng_key(seed, site), e >> 2);
  f32x4 s;
  s[0] = ((q.lo & 0xffffu) >= thr) ? inv_keep : 0.f;
  s[1] = (q.lo >= (thr << 16)) ? inv_keep : 0.f;
  s[2] = ((q.hi & 0xffffu) >= thr) ? inv_keep : 0.f;
  s[3] = (q.hi >= (thr << 16)) ? inv_keep : 0.f;
  return s;
}

// the same draws as 4 keep bits (bit j = element e + j kept), and back to a scale vector: the LayerNorm backward
// carries a row's dropout decisions from its first to its second phase in one register instead of twelve
__device__ __forceinline__ uint32_t drop_bits4(uint32_t seed, uint32_t site, uint32_t e, uint32_t thr) {
  const czc_quad q = czc_rng_quad(czc_rng_key(seed, site), e >> 2);
  return ((q.lo & 0xffffu) >= thr ? 1u : 0u) | (q.lo >= (thr << 16) ? 2u : 0u) |
         ((q.hi & 0xffffu) >= thr ? 4u : 0u) | (q.hi >= (thr << 16) ? 8u : 0u);
}
__device__ __forceinline__ f32x4 bits_scale4(uint32_t bits, float inv_keep) {
  return (f32x4){(bits & 1u) ? inv_keep : 0.f, (bits & 2u) ? inv_keep : 0.f, (bits & 4u) ? inv_keep : 0.f,
                 (bits & 8u) ? inv_keep : 0.f};
}

struct LnStat { float mean, rstd; };

// mean / rstd of a row held as v[NIT] chunks (invalid chunks must be zero)
template <int NIT>
__device__ __forceinline__ LnStat row_stats(const f32x4* v, int H, int lane, float eps) {
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if ((lane + 64 * it) * 4 < H) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[it][j] - mean; q += d * d; }
    }
  }
  const float var = wave_sum(q) / (float)H;
  return {mean, rsqrtf(var + eps)};
}

// ---------------------------------------------------------------------------------------------
// x = LN(dropout(y) + resid)
// ---------------------------------------------------------------------------------------------
// EXACT: the row is exactly NIT * 256 elements (768, 1024 ...): every "column < H" guard is true at compile time, the
// row's loads, arithmetic and stores form one basic block
template <int NIT, bool EXACT>
__global__ __launch_bounds__(256) void residual_ln_fwd_kernel(
    const bf16_t* __restrict__ y, const bf16_t* __restrict__ resid, int rows, int H_,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    uint32_t thr, float inv_keep, uint32_t seed, uint32_t site, bf16_t* __restrict__ x, float* __restrict__ stats) {
  const int H = EXACT ? NIT * 256 : H_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gm[NIT], bt[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    gm[it] = ld_vec4(gamma, (lane + 64 * it) * 4, H);
    bt[it] = ld_vec4(beta, (lane + 64 * it) * 4, H);
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const size_t base = (size_t)row * H;
    f32x4 z[NIT], r[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      z[it] = ld_row4<true>(y + base, (lane + 64 * it) * 4, H);
      r[it] = ld_row4<true>(resid + base, (lane + 64 * it) * 4, H);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (thr) z[it] *= drop_scale4(seed, site, (uint32_t)(base + c), thr, inv_keep);
      z[it] += r[it];
    }
    const LnStat st = row_stats<NIT>(z, H, lane, eps);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (c < H) st_bf4(x + base + c, (z[it] - st.mean) * st.rstd * gm[it] + bt[it]);
    }
    if (lane == 0) { stats[2 * (size_t)row] = st.mean; stats[2 * (size_t)row + 1] = st.rstd; }
  }
}

// ---------------------------------------------------------------------------------------------
// backward of the above.  dz_out: gradient w.r.t. (dropout(y) + resid); dy_out: w.r.t. y.
// ---------------------------------------------------------------------------------------------
template <int NIT>
__device__ __forceinline__ void flush_colsums(float* lds_acc, const f32x4 (&acc)[3][NIT], int H, int lane,
                                              float* d0, float* d1, float* d2, float* part = nullptr) {
  // lds_acc [3][1024] zeroed by the caller; waves add their lane-private partial sums
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (c < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&lds_acc[s * 1024 + c + j], acc[s][it][j]);
      }
    }
  __syncthreads();
  if (part) {
    // one row of per-workgroup partial sums; reduce_partials_kernel folds the rows afterwards.  (Thousands of
    // workgroups adding to the same few hundred addresses serialise at the memory side: ~0.1 us per add.)
    float* row = part + (size_t)blockIdx.x * 3 * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
      row[c] = lds_acc[c];
      row[H + c] = lds_acc[1024 + c];
      row[2 * H + c] = lds_acc[2048 + c];
    }
    return;
  }
  for (int c = threadIdx.x; c < H; c += blockDim.x) {
    if (d0) unsafeAtomicAdd(d0 + c, lds_acc[c]);
    if (d1) unsafeAtomicAdd(d1 + c, lds_acc[1024 + c]);
    if (d2) unsafeAtomicAdd(d2 + c, lds_acc[2048 + c]);
  }
}

// dst[c] += sum_r part[r * row_stride + c]   for c < ncols; blockIdx.y splits the rows (<= 32-deep atomic chains)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nrows, int row_stride,
                                                              int ncols, float* dst) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  // eight rows' loads in flight per thread: a rolled loop paid one memory round trip per row (6.9 us for 2.4 MB)
  float s = 0.f;
  int r = blockIdx.y;
  const int step = gridDim.y;
  for (; r + 7 * step < nrows; r += 8 * step) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = part[(size_t)(r + i * step) * row_stride + c];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  for (; r < nrows; r += step) s += part[(size_t)r * row_stride + c];
  unsafeAtomicAdd(dst + c, s);
}
// LN kernels: rows of [3][H] partials -> three destinations in one launch
__global__ __launch_bounds__(256) void reduce_partials3_kernel(const float* __restrict__ part, int nrows, int H, float* d0,
                                                               float* d1, float* d2) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;          // column in [0, 3H)
  float s = 0.f;
  if (c < 3 * H) {
    int r = blockIdx.y * 4 + ry;
    const int step = gridDim.y * 4;
    for (; r + 7 * step < nrows; r += 8 * step) {   // eight loads in flight (see reduce_partials_kernel)
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = part[(size_t)(r + i * step) * 3 * H + c];
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; r < nrows; r += step) s += part[(size_t)r * 3 * H + c];
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < 3 * H) {
    s = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
    const int seg = c / H, cc = c - seg * H;
    float* d = seg == 0 ? d0 : (seg == 1 ? d1 : d2);
    if (d) unsafeAtomicAdd(d + cc, s);
  }
}
struct MultiReduceArgs { ReduceJob j[4]; int first[5]; int chunks[4]; };
// block -> (job, 64-column block, row chunk); 64 columns x 4 row lanes per block, eight rows' loads in flight per thread
__global__ __launch_bounds__(256) void multi_reduce_kernel(const MultiReduceArgs a) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  int k = 0;
  while (k < 3 && (int)blockIdx.x >= a.first[k + 1]) ++k;
  const ReduceJob& jb = a.j[k];
  const int local = blockIdx.x - a.first[k], chunks = a.chunks[k];
  const int c = (local / chunks) * 64 + cx, chunk = local % chunks;
  float s = 0.f;
  if (c < jb.ncols) {
    const float* src = jb.src + c;
    int r = chunk * 4 + ry;
    const int step = chunks * 4;
    for (; r + 7 * step < jb.nrows; r += 8 * step) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(r + i * step) * jb.row_stride];
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; r < jb.nrows; r += step) s += src[(size_t)r * jb.row_stride];
  }
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < jb.ncols) {
    s = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
    const int sg = c / jb.seg, cc = c - sg * jb.seg;
    float* d = jb.dst[sg];
    if (d) unsafeAtomicAdd(d + cc, s);
  }
}
int multi_reduce(const ReduceJob* jobs, int njobs, hipStream_t st) {
  CZC_REQUIRE(njobs >= 1 && njobs <= 4, "multi_reduce: %d jobs", njobs);
  MultiReduceArgs a{};
  int nb = 0;
  for (int k = 0; k < 4; ++k) {
    a.first[k] = nb;
    if (k < njobs) {
      a.j[k] = jobs[k];
      int ch = jobs[k].nrows / 32;
      ch = ch < 1 ? 1 : (ch > 16 ? 16 : ch);
      a.chunks[k] = ch;
      nb += ceil_div(jobs[k].ncols, 64) * ch;
    } else {
      a.chunks[k] = 1;
    }
  }
  a.first[4] = nb;
  for (int k = njobs; k < 4; ++k) a.first[k] = nb;
  hipLaunchKernelGGL(multi_reduce_kernel, dim3(nb), dim3(256), 0, st, a);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

static void launch_reduce_partials3(const float* part, int nrows, int H, float* d0, float* d1, float* d2, hipStream_t st) {
  int chunks = nrows / 32;
  if (chunks < 1) chunks = 1;
  if (chunks > 16) chunks = 16;
  hipLaunchKernelGGL(reduce_partials3_kernel, dim3(ceil_div(3 * H, 64), chunks), dim3(256), 0, st, part, nrows, H, d0, d1, d2);
}
static void launch_reduce_partials(const float* part, int nrows, int row_stride, int ncols, float* dst, hipStream_t st) {
  if (!dst) return;
  const int chunks = nrows < 32 ? nrows : 32;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(ceil_div(ncols, 256), chunks), dim3(256), 0, st, part, nrows, row_stride,
                     ncols, dst);
}

template <int NIT, bool EXACT>
__global__ __launch_bounds__(256, NIT <= 3 ? 4 : 2) void residual_ln_bwd_kernel(
    const bf16_t* __restrict__ dx, const bf16_t* __restrict__ dx2, const bf16_t* __restrict__ y,
    const bf16_t* __restrict__ resid, const float* __restrict__ stats, int rows, int H_,
    const float* __restrict__ gamma, uint32_t thr, float inv_keep, uint32_t seed, uint32_t site,
    bf16_t* __restrict__ dz_out, bf16_t* __restrict__ dy_out, float* d_gamma, float* d_beta, float* d_bias, float* part) {
  const int H = EXACT ? NIT * 256 : H_;
  __shared__ float lds_acc[3 * 1024];
  for (int i = threadIdx.x; i < 3 * 1024; i += blockDim.x) lds_acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[3][NIT];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int it = 0; it < NIT; ++it) acc[s][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 gm[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (lane + 64 * it) * 4;
    gm[it] = (c < H) ? *(const f32x4*)(gamma + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const size_t base = (size_t)row * H;
    const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
    f32x4 xh[NIT], gh[NIT], rs_[NIT];
    uint32_t keep = 0;
    float s1 = 0.f, s2 = 0.f;
    // every load of the row first (clamped addresses, see ld_row4), then the arithmetic
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      xh[it] = ld_row4<true>(y + base, c, H);
      rs_[it] = ld_row4<true>(resid + base, c, H);
      gh[it] = ld_row4<true>(dx + base, c, H);
    }
    if (dx2) {
      f32x4 g2[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) g2[it] = ld_row4<true>(dx2 + base, (lane + 64 * it) * 4, H);
#pragma unroll
      for (int it = 0; it < NIT; ++it) gh[it] += g2[it];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      const bool ok = c < H;
      f32x4 a = xh[it];
      if (thr) {
        const uint32_t kb = drop_bits4(seed, site, (uint32_t)(base + c), thr);
        keep |= kb << (4 * it);
        a *= bits_scale4(kb, inv_keep);
      }
      const f32x4 z = a + rs_[it];
      xh[it] = ok ? (z - mean) * rstd : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 g = gh[it];
      acc[0][it] += g * xh[it];
      acc[1][it] += g;
      gh[it] = g * gm[it];
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1 += gh[it][j]; s2 += gh[it][j] * xh[it][j]; }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (c < H) {
        const f32x4 dz = (gh[it] - s1 - xh[it] * s2) * rstd;
        st_bf4(dz_out + base + c, dz);
        f32x4 dy = dz;
        if (thr) dy *= bits_scale4(keep >> (4 * it), inv_keep);
        if (thr || dy_out != dz_out) st_bf4(dy_out + base + c, dy);
        acc[2][it] += dy;
      }
    }
  }
  flush_colsums<NIT>(lds_acc, acc, H, lane, d_gamma, d_beta, d_bias, part);
}

// ---------------------------------------------------------------------------------------------
// embeddings: x = dropout(LN(word[id] + pos[s] + type[0]))
// ---------------------------------------------------------------------------------------------
template <int NIT>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(
    const int64_t* __restrict__ ids, int rows, int S, int H, int vocab, const float* __restrict__ word,
    const float* __restrict__ pos, const float* __restrict__ type0, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, uint32_t thr, float inv_keep, uint32_t seed, uint32_t site,
    bf16_t* __restrict__ x, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int s = row % S;
    const size_t base = (size_t)row * H;
    f32x4 z[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      z[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (c < H) {
        // same association as HF: (word + type) + pos
        z[it] = (*(const f32x4*)(word + (size_t)id * H + c) + *(const f32x4*)(type0 + c)) +
                *(const f32x4*)(pos + (size_t)s * H + c);
      }
    }
    const LnStat st = row_stats<NIT>(z, H, lane, eps);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (c < H) {
        const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
        f32x4 o = (z[it] - st.mean) * st.rstd * g + b;
        if (thr) o *= drop_scale4(seed, site, (uint32_t)(base + c), thr, inv_keep);
        st_bf4(x + base + c, o);
      }
    }
    if (lane == 0) { stats[2 * (size_t)row] = st.mean; stats[2 * (size_t)row + 1] = st.rstd; }
  }
}

template <int NIT>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(
    const int64_t* __restrict__ ids, int rows, int S, int H, int vocab, const float* __restrict__ word,
    const float* __restrict__ pos, const float* __restrict__ type0, const float* __restrict__ gamma,
    const float* __restrict__ stats, uint32_t thr, float inv_keep, uint32_t seed, uint32_t site,
    const bf16_t* __restrict__ dx, float* d_word, float* d_pos, float* d_type0, float* d_gamma, float* d_beta, float* part) {
  __shared__ float lds_acc[3 * 1024];
  __shared__ __attribute__((aligned(16))) float tr_all[4][1024];
  __shared__ int order_all[4][64];
  float* tr = tr_all[threadIdx.x >> 6];
  int* order = order_all[threadIdx.x >> 6];
  for (int i = threadIdx.x; i < 3 * 1024; i += blockDim.x) lds_acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[3][NIT];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int it = 0; it < NIT; ++it) acc[s][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 gm[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (lane + 64 * it) * 4;
    gm[it] = (c < H) ? *(const f32x4*)(gamma + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // A wave walks over units of work = one position s x a chunk of the batch (rows b*S + s for b in the chunk): the
  // position-embedding gradient of the chunk stays in registers and is added to d_pos ONCE per unit.  Float atomics run
  // at one 256-byte instruction per ~120 cycles per CU whatever the address pattern: a row of atomics per token into
  // d_pos doubled the kernel's atomic traffic for a tensor that has 256 rows.
  const int Bt = rows / S;
  const int CH = Bt >= 256 ? 16 : (Bt >= 16 ? Bt / 16 : 1), per = (Bt + CH - 1) / CH;
  for (int u = blockIdx.x * 4 + wave; u < S * CH; u += gridDim.x * 4) {
   const int s = u / CH, b0 = (u - s * CH) * per, b1 = min(Bt, b0 + per);
   f32x4 pacc[NIT];
#pragma unroll
   for (int it = 0; it < NIT; ++it) pacc[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
   // Word rows: consecutive tokens of the unit with the SAME id (the [CLS] column, runs of padding) are summed in
   // registers and leave as one row of atomics, and a run whose incoming gradient rows are all exactly zero (padded
   // positions: nothing attends to them and the pooling masks them, so their gradient is 0.0 all the way down) adds
   // nothing at all.  On text-like ids the per-token form serialised on the hot rows in L2: 233 us with uniform ids,
   // 441 us with a Zipf head, 766 us with 30 % padding (tools/bench_embed_bwd.py).
   f32x4 wacc[NIT];
#pragma unroll
   for (int it = 0; it < NIT; ++it) wacc[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
   int cur_id = -1;
   bool wany = false;
   auto flush_word = [&]() {
    if (cur_id >= 0 && wany) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = (lane + 64 * it) * 4;
        if (c < H) *(f32x4*)(tr + c) = wacc[it];     // wave-private row image in LDS
      }
      // scatter-add with every atomic wave-instruction covering 256 CONTIGUOUS bytes (lane l -> element 64k + l):
      // four 4-byte adds 16 B apart per lane ran at a quarter of this rate (MI355X float-atomic access shape)
      for (int c = lane; c < H; c += 64) unsafeAtomicAdd(d_word + (size_t)cur_id * H + c, tr[c]);
    }
   };
   // the unit's tokens in the order of their ids (a rank by counting, one token per lane): equal ids become neighbours,
   // so a frequent word costs the unit one row of atomics, not one per occurrence
   const int nu = b1 - b0;
   const bool sorted = nu <= 64;
   if (sorted) {
    int64_t m64 = lane < nu ? ids[(size_t)(b0 + lane) * S + s] : 0;
    const int mine = (int)(m64 < 0 ? 0 : (m64 >= vocab ? vocab - 1 : m64));
    int rank = 0;
    for (int j = 0; j < nu; ++j) {
      const int other = __builtin_amdgcn_readlane(mine, j);
      rank += (other < mine || (other == mine && j < lane)) ? 1 : 0;
    }
    if (lane < nu) order[rank] = lane;
   }
   for (int k = 0; k < nu; ++k) {
    const int b = b0 + (sorted ? order[k] : k);
    const int row = b * S + s;
    int64_t id64 = ids[row];
    id64 = id64 < 0 ? 0 : (id64 >= vocab ? vocab - 1 : id64);
    const int id = __builtin_amdgcn_readfirstlane((int)id64);
    if (id != cur_id) {
      flush_word();
#pragma unroll
      for (int it = 0; it < NIT; ++it) wacc[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      cur_id = id;
      wany = false;
    }
    bool nz = false;
    const size_t base = (size_t)row * H;
    const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
    f32x4 xh[NIT], gh[NIT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      xh[it] = gh[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (c < H) {
        const f32x4 z = (*(const f32x4*)(word + (size_t)id * H + c) + *(const f32x4*)(type0 + c)) +
                        *(const f32x4*)(pos + (size_t)s * H + c);
        xh[it] = (z - mean) * rstd;
        f32x4 g = ld_bf4(dx + base + c);
        nz = nz || g[0] != 0.f || g[1] != 0.f || g[2] != 0.f || g[3] != 0.f;
        if (thr) g *= drop_scale4(seed, site, (uint32_t)(base + c), thr, inv_keep);
        acc[0][it] += g * xh[it];
        acc[1][it] += g;
        gh[it] = g * gm[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1 += gh[it][j]; s2 += gh[it][j] * xh[it][j]; }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (c < H) {
        const f32x4 dz = (gh[it] - s1 - xh[it] * s2) * rstd;
        acc[2][it] += dz;
        pacc[it] += dz;
        wacc[it] += dz;
      }
    }
    wany = wany || __any(nz);
   }
   flush_word();
   if (b1 > b0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (c < H) *(f32x4*)(tr + c) = pacc[it];
    }
    for (int c = lane; c < H; c += 64) unsafeAtomicAdd(d_pos + (size_t)s * H + c, tr[c]);
   }
  }
  flush_colsums<NIT>(lds_acc, acc, H, lane, d_gamma, d_beta, d_type0, part);
}

// ---------------------------------------------------------------------------------------------
// pooling.  One 256-thread block per sequence, thread t owns columns 4t..4t+3.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void pool_fwd_kernel(const bf16_t* __restrict__ x, const uint8_t* __restrict__ mask,
                                                       int S, int H, int pooling, int normalize,
                                                       float* __restrict__ emb, float* __restrict__ raw) {
  __shared__ float red[4];
  __shared__ uint8_t m_s[512];
  const int b = blockIdx.x, c = threadIdx.x * 4;
  for (int s = threadIdx.x; s < S; s += 256) m_s[s] = mask[(size_t)b * S + s];
  __syncthreads();
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  const bf16_t* xb = x + (size_t)b * S * H;
  if (pooling == CZC_POOL_AVERAGE) {
    // sixteen rows in flight per thread (a load under `if (mask)` was waited for before the next was issued: one memory
    // round trip per token; one workgroup per CU runs here, so the depth per thread is what fills the memory pipe);
    // masked rows are loaded and multiplied by 0
    constexpr int U = 16;
    f32x4 part[U];
#pragma unroll
    for (int u = 0; u < U; ++u) part[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cc = (c < H) ? c : 0;
    for (int s = 0; s < S; s += U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld_bf4(xb + (size_t)min(s + u, S - 1) * H + cc);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool on = (s + u < S) && m_s[min(s + u, S - 1)];
        cnt += on ? 1 : 0;
        part[u] += v[u] * (on ? 1.0f : 0.0f);
      }
    }
    // the same summation tree as with eight partial sums per thread: tokens s, s + 8 share a partial sum
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] += part[u + 8];
    sum = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    // torch: sum / count  (count == 0 gives nan/inf exactly as the reference's 0/0 would)
    const float inv = 1.0f / (float)cnt;
    sum = sum * inv;
    if (cnt == 0) sum = sum * __builtin_nanf("");
  } else {
    if (m_s[0] && c < H) sum = ld_bf4(xb + c);
  }
  float nrm = 1.f;
  if (normalize) {
    const float ss = block_sum_256((c < H) ? (sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2] + sum[3] * sum[3]) : 0.f, red);
    nrm = fmaxf(sqrtf(ss), 1e-12f);
  }
  if (c < H) {
    if (raw) *(f32x4*)(raw + (size_t)b * H + c) = sum;
    *(f32x4*)(emb + (size_t)b * H + c) = normalize ? sum / nrm : sum;
  }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ d_emb, const float* __restrict__ raw,
                                                       const uint8_t* __restrict__ mask, int S, int H, int pooling,
                                                       int normalize, bf16_t* __restrict__ dx) {
  __shared__ float red[4];
  __shared__ uint8_t m_s[512];
  const int b = blockIdx.x, c = threadIdx.x * 4;
  for (int s = threadIdx.x; s < S; s += 256) m_s[s] = mask[(size_t)b * S + s];
  __syncthreads();
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  if (c < H) g = *(const f32x4*)(d_emb + (size_t)b * H + c);
  if (normalize) {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (c < H) r = *(const f32x4*)(raw + (size_t)b * H + c);
    const float ss = block_sum_256(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], red);
    const float dot = block_sum_256(r[0] * g[0] + r[1] * g[1] + r[2] * g[2] + r[3] * g[3], red);
    const float nrm = sqrtf(ss);
    if (nrm >= 1e-12f) g = (g - r * (dot / ss)) / nrm;   // d/dr of r/|r|
    else g = g / 1e-12f;                                   // clamp_min branch of F.normalize
  }
  int cnt = 0;
  for (int s = 0; s < S; ++s) cnt += m_s[s] ? 1 : 0;
  bf16_t* db = dx + (size_t)b * S * H;
  if (c < H) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (pooling == CZC_POOL_AVERAGE) {
      const f32x4 gs = g * (1.0f / (float)cnt);
      for (int s = 0; s < S; ++s) st_bf4(db + (size_t)s * H + c, m_s[s] ? gs : zero);
    } else {
      for (int s = 0; s < S; ++s) st_bf4(db + (size_t)s * H + c, (s == 0 && m_s[0]) ? g : zero);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// column sums of a bf16 matrix: col[n] += sum_m x[m][n]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int rows, int cols, int ld,
                                                     int rows_per_block, float* col) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = r0; r < r1; ++r) {
    const uint4 w = *(const uint4*)(x + (size_t)r * ld + c);
    a[0] += bflo(w.x); a[1] += bfhi(w.x); a[2] += bflo(w.y); a[3] += bfhi(w.y);
    a[4] += bflo(w.z); a[5] += bfhi(w.z); a[6] += bflo(w.w); a[7] += bfhi(w.w);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) unsafeAtomicAdd(col + c + j, a[j]);
}

__global__ void dropout_mask_kernel(uint32_t seed, uint32_t site, uint32_t thr, int64_t n, uint8_t* keep) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    keep[i] = czc_keep(seed, site, (uint32_t)i, thr) ? 1 : 0;
}

static int row_grid(int rows) {
  int g = ceil_div(rows, 4);
  return g > 2048 ? 2048 : (g < 1 ? 1 : g);
}
static bool hidden_ok(int H) { return H > 0 && H <= 1024 && (H % 8) == 0; }
int ln_bwd_partial_rows(int rows) { const int g = row_grid(rows); return g > 1024 ? 1024 : g; }

}  // namespace czc

using namespace czc;

#define LAUNCH_NIT(kern, H, grid, st, ...)                                                         \
  do {                                                                                              \
    if ((H) <= 256) hipLaunchKernelGGL(kern<1>, grid, dim3(256), 0, st, __VA_ARGS__);              \
    else if ((H) <= 512) hipLaunchKernelGGL(kern<2>, grid, dim3(256), 0, st, __VA_ARGS__);         \
    else if ((H) <= 768) hipLaunchKernelGGL(kern<3>, grid, dim3(256), 0, st, __VA_ARGS__);         \
    else hipLaunchKernelGGL(kern<4>, grid, dim3(256), 0, st, __VA_ARGS__);                         \
  } while (0)

#define LAUNCH_NIT_EXACT(kern, H, grid, st, ...)                                                   \
  do {                                                                                              \
    if ((H) == 768) hipLaunchKernelGGL((kern<3, true>), grid, dim3(256), 0, st, __VA_ARGS__);       \
    else if ((H) == 1024) hipLaunchKernelGGL((kern<4, true>), grid, dim3(256), 0, st, __VA_ARGS__); \
    else if ((H) <= 256) hipLaunchKernelGGL((kern<1, false>), grid, dim3(256), 0, st, __VA_ARGS__); \
    else if ((H) <= 512) hipLaunchKernelGGL((kern<2, false>), grid, dim3(256), 0, st, __VA_ARGS__); \
    else if ((H) <= 768) hipLaunchKernelGGL((kern<3, false>), grid, dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((kern<4, false>), grid, dim3(256), 0, st, __VA_ARGS__);                 \
  } while (0)

#define DROP_ARGS(p) \
  const uint32_t thr = (p) > 0.f ? czc_drop_thr(p) : 0u; \
  const float inv_keep = (p) > 0.f ? 1.0f / (1.0f - (p)) : 1.0f

extern "C" int czc_residual_ln_forward(const uint16_t* y, const uint16_t* resid, int rows, int hidden,
                                       const float* gamma, const float* beta, float eps, float dropout_p,
                                       uint32_t seed, uint32_t site, uint16_t* x, float* stats, czc_stream_t stream) {
  CZC_REQUIRE(hidden_ok(hidden), "czc_residual_ln_forward: hidden=%d must be a multiple of 8 and <= 1024", hidden);
  CZC_REQUIRE(rows > 0 && dropout_p >= 0.f && dropout_p < 1.f, "czc_residual_ln_forward: rows=%d p=%f", rows, dropout_p);
  ProfScope prof((hipStream_t)stream, "residual_ln_fwd", 0.0, 2.0 * rows * (double)hidden * 3.0 + 8.0 * rows);
  DROP_ARGS(dropout_p);
  LAUNCH_NIT_EXACT(residual_ln_fwd_kernel, hidden, dim3(row_grid(rows)), (hipStream_t)stream, y, resid, rows, hidden, gamma, beta,
             eps, thr, inv_keep, seed, site, x, stats);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_residual_ln_backward(const uint16_t* dx, const uint16_t* dx2, const uint16_t* y,
                                        const uint16_t* resid, const float* stats, int rows, int hidden,
                                        const float* gamma, float dropout_p, uint32_t seed, uint32_t site,
                                        uint16_t* dz_out, uint16_t* dy_out, float* d_gamma, float* d_beta,
                                        float* d_bias, float* partials, czc_stream_t stream) {
  CZC_REQUIRE(hidden_ok(hidden), "czc_residual_ln_backward: hidden=%d must be a multiple of 8 and <= 1024", hidden);
  CZC_REQUIRE(rows > 0 && dropout_p >= 0.f && dropout_p < 1.f, "czc_residual_ln_backward: rows=%d p=%f", rows, dropout_p);
  CZC_REQUIRE(dz_out && dy_out, "czc_residual_ln_backward: outputs missing");
  CZC_REQUIRE(!(dropout_p > 0.f && dz_out == dy_out), "czc_residual_ln_backward: dy_out must differ from dz_out when p > 0");
  ProfScope prof((hipStream_t)stream, "residual_ln_bwd", 0.0,
                 2.0 * rows * (double)hidden * (3.0 + (dx2 ? 1.0 : 0.0) + 1.0 + (dropout_p > 0.f ? 1.0 : 0.0)) + 8.0 * rows);
  DROP_ARGS(dropout_p);
  // 119 VGPRs -> 4 workgroups per CU are resident: exactly one round of 1024 (measured: 2048 -> 206 us, 1024 -> 155 us,
  // 512 -> 200 us at 65536 x 768)
  int g = row_grid(rows);
  if (g > 1024) g = 1024;
  LAUNCH_NIT_EXACT(residual_ln_bwd_kernel, hidden, dim3(g), (hipStream_t)stream, dx, dx2, y, resid, stats, rows, hidden, gamma, thr,
             inv_keep, seed, site, dz_out, dy_out, d_gamma, d_beta, d_bias, partials);
  if (partials && !g_defer_partials) launch_reduce_partials3(partials, g, hidden, d_gamma, d_beta, d_bias, (hipStream_t)stream);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_embed_ln_forward(const int64_t* ids, int batch, int seq, int hidden, int vocab, const float* word,
                                    const float* pos, const float* type0, const float* gamma, const float* beta,
                                    float eps, float dropout_p, uint32_t seed, uint32_t site, uint16_t* x,
                                    float* stats, czc_stream_t stream) {
  CZC_REQUIRE(hidden_ok(hidden), "czc_embed_ln_forward: hidden=%d must be a multiple of 8 and <= 1024", hidden);
  CZC_REQUIRE(batch > 0 && seq > 0 && vocab > 0, "czc_embed_ln_forward: batch=%d seq=%d vocab=%d", batch, seq, vocab);
  ProfScope prof((hipStream_t)stream, "embed_ln_fwd", 0.0, (double)batch * seq * hidden * (4.0 + 2.0) + 8.0 * batch * seq);
  DROP_ARGS(dropout_p);
  const int rows = batch * seq;
  LAUNCH_NIT(embed_ln_fwd_kernel, hidden, dim3(row_grid(rows)), (hipStream_t)stream, ids, rows, seq, hidden, vocab, word, pos,
             type0, gamma, beta, eps, thr, inv_keep, seed, site, x, stats);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_embed_ln_backward(const int64_t* ids, int batch, int seq, int hidden, int vocab, const float* word,
                                     const float* pos, const float* type0, const float* gamma, const float* stats,
                                     float dropout_p, uint32_t seed, uint32_t site, const uint16_t* dx, float* d_word,
                                     float* d_pos, float* d_type0, float* d_gamma, float* d_beta, float* partials,
                                     czc_stream_t stream) {
  CZC_REQUIRE(hidden_ok(hidden), "czc_embed_ln_backward: hidden=%d must be a multiple of 8 and <= 1024", hidden);
  CZC_REQUIRE(batch > 0 && seq > 0 && vocab > 0, "czc_embed_ln_backward: batch=%d seq=%d vocab=%d", batch, seq, vocab);
  ProfScope prof((hipStream_t)stream, "embed_ln_bwd", 0.0, (double)batch * seq * hidden * (4.0 + 2.0 + 4.0) + 8.0 * batch * seq);
  DROP_ARGS(dropout_p);
  const int rows = batch * seq;
  int g = row_grid(rows);
  if (g > 1024) g = 1024;
  LAUNCH_NIT(embed_ln_bwd_kernel, hidden, dim3(g), (hipStream_t)stream, ids, rows, seq, hidden, vocab, word, pos, type0,
             gamma, stats, thr, inv_keep, seed, site, dx, d_word, d_pos, d_type0, d_gamma, d_beta, partials);
  if (partials) launch_reduce_partials3(partials, g, hidden, d_gamma, d_beta, d_type0, (hipStream_t)stream);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_pool_forward(const uint16_t* x, const uint8_t* mask, int batch, int seq, int hidden, int pooling,
                                int normalize, float* emb, float* raw_out, czc_stream_t stream) {
  CZC_REQUIRE(hidden_ok(hidden), "czc_pool_forward: hidden=%d must be a multiple of 8 and <= 1024", hidden);
  CZC_REQUIRE(batch > 0 && seq > 0 && seq <= 512, "czc_pool_forward: batch=%d seq=%d (seq <= 512)", batch, seq);
  CZC_REQUIRE(pooling == CZC_POOL_AVERAGE || pooling == CZC_POOL_CLS, "czc_pool_forward: pooling=%d", pooling);
  ProfScope prof((hipStream_t)stream, "pool_fwd", 0.0, 2.0 * batch * (double)seq * hidden);
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, x, mask, seq, hidden, pooling,
                     normalize, emb, raw_out);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_pool_backward(const float* d_emb, const float* raw, const uint8_t* mask, int batch, int seq,
                                 int hidden, int pooling, int normalize, uint16_t* dx, czc_stream_t stream) {
  CZC_REQUIRE(hidden_ok(hidden), "czc_pool_backward: hidden=%d must be a multiple of 8 and <= 1024", hidden);
  CZC_REQUIRE(batch > 0 && seq > 0 && seq <= 512, "czc_pool_backward: batch=%d seq=%d (seq <= 512)", batch, seq);
  CZC_REQUIRE(!normalize || raw, "czc_pool_backward: normalize needs the raw embedding");
  ProfScope prof((hipStream_t)stream, "pool_bwd", 0.0, 2.0 * batch * (double)seq * hidden);
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, d_emb, raw, mask, seq, hidden,
                     pooling, normalize, dx);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int64_t czc_colsum_scratch_floats(int hidden) { return (int64_t)2048 * 3 * hidden; }

extern "C" int czc_reduce_partials(const float* partials, int nrows, int row_stride, int ncols, float* dst,
                                   czc_stream_t stream) {
  CZC_REQUIRE(partials && dst && nrows > 0 && ncols > 0 && row_stride >= ncols, "czc_reduce_partials: bad arguments");
  ProfScope prof((hipStream_t)stream, "reduce_partials", 0.0, 4.0 * nrows * (double)ncols);
  launch_reduce_partials(partials, nrows, row_stride, ncols, dst, (hipStream_t)stream);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_colsum_bf16(const uint16_t* x, int rows, int cols, int ld, float* col, czc_stream_t stream) {
  CZC_REQUIRE(rows > 0 && cols > 0 && (cols % 8) == 0 && (ld % 8) == 0, "czc_colsum_bf16: rows=%d cols=%d ld=%d", rows, cols, ld);
  const int rpb = 128;
  dim3 grid(ceil_div(cols, 2048), ceil_div(rows, rpb));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, rpb, col);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_dropout_mask(uint32_t seed, uint32_t site, float p, int64_t n, uint8_t* keep, czc_stream_t stream) {
  CZC_REQUIRE(p >= 0.f && p < 1.f && n > 0 && n < (1ll << 32), "czc_dropout_mask: p=%f n=%lld", p, (long long)n);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed, site,
                     p > 0.f ? czc_drop_thr(p) : 0u, n, keep);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}
