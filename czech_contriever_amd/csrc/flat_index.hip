// flat_index.hip — exact inner-product top-k over a flat fp32 passage index, HBM-resident (gfx950).
//
// Replaces the `faiss.IndexFlatIP.search` call of Ajchler/czech-contriever src/index.py:43 (driven from
// passage_retrieval.py:65-91,184-186).  Same contract: fp32 inner products, the k best per query in
// descending score order.  Ties are ordered by ascending passage row (faiss leaves them unspecified).
//
// Two ways through the index, same answer (tests run both against the oracle):
//   czc_index_scan          every score on the fp32 matrix cores (below);
//   czc_index_coarse_scan   scores on the bf16 matrix cores with a rigorous error margin per (query, row), survivors
//                           re-scored in fp32 (second half of this file) — the default of the host side, ~6x faster.
// The fp32 scan: two kernels per chunk of CZC_INDEX_CHUNK passage rows, all on the caller's stream, no host round trip:
//   ip_scan_kernel     scores[128 queries x 256 rows] per workgroup on the fp32 matrix cores
//                      (v_mfma_f32_32x32x2_f32: products and sums in fp32, as an sgemm would), and in the epilogue
//                      every score ABOVE the query's running k-th best (tau) is appended to the query's candidate
//                      list.  The first chunk of a search has no tau yet and is written densely (no atomics).
//   topk_compact_kernel one workgroup per query: exact k best of (kept k + appended) — a bitonic sort in LDS when
//                      the list is short (every chunk but the first), a 3-level radix select first when it is not —
//                      written back to the head of the list in order, and tau := the k-th score.
// Anything in the final top k beats the running k-th best at every point of the scan, so filtering by tau loses
// nothing: the result equals a full sort of all N scores.
#include "czc_common.h"
#include "gemm_args.h"

namespace czc {

constexpr int IX_TQ = 128;     // queries per workgroup
constexpr int IX_TR = 256;     // passage rows per workgroup
constexpr int IX_BK = 16;      // K (embedding dim) step per LDS stage
constexpr int IX_SORT = 4096;  // LDS sort capacity (composite 64-bit keys)
constexpr int IX_KMAX = 2048;  // largest k

typedef __attribute__((ext_vector_type(16))) float acc16;

// position (in 16-B units) of chunk c of row `row` in a [rows][16 floats] LDS tile: the 4 chunks of a row are
// permuted by the row's bits 1..2 so that 8 consecutive rows reading the same chunk touch 8 different bank groups
// 16-byte global load issued where it is written: a plain load was sunk down to its LDS store, behind the 64 MFMAs it
// is meant to overlap, and waited for on the spot (a volatile one is waited for right after issue).  The compiler does
// not see the load, so ix_wait_loads() names every destination before its first use.
typedef __attribute__((ext_vector_type(4))) float ix_f4;
__device__ __forceinline__ void ix_ld16_async(ix_f4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ix_wait_loads(ix_f4& a, ix_f4& b, ix_f4& c, ix_f4& d, ix_f4& e, ix_f4& f) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "memory");
}
__device__ __forceinline__ int ix_slot(int row, int c) { return row * 4 + (c ^ ((row >> 1) & 3)); }

// ascending-ordered unsigned image of a float
__device__ __forceinline__ uint32_t ix_key(float s) {
  const uint32_t u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ix_unkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// 64-bit sort key: larger = better score, then smaller row
__device__ __forceinline__ uint64_t ix_comp(float s, int id) {
  return ((uint64_t)ix_key(s) << 32) | (uint32_t)(~(uint32_t)id);
}

template <bool DENSE>
__global__ __launch_bounds__(256, 2) void ip_scan_kernel(
    const float* __restrict__ db, int rows, int dim, int id_base, const float* __restrict__ queries, int nq,
    const float* __restrict__ tau, float* __restrict__ cand_score, int* __restrict__ cand_id,
    int* __restrict__ cand_count, int64_t cap, int q_tiles, int r_tiles) {
  __shared__ ix_f4 sA[2][IX_TQ * 4];
  __shared__ ix_f4 sB[2][IX_TR * 4];

  // workgroups go round-robin over the 8 XCDs: give each XCD its own passage tiles (all query tiles of one passage
  // tile run back to back on the same L2), so that the index streams from HBM once
  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int r_tile = (local / q_tiles) * 8 + xcd;
  const int q_tile = local % q_tiles;
  if (r_tile >= r_tiles) return;
  const int q0 = q_tile * IX_TQ, r0 = r_tile * IX_TR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // staging: thread -> (row lr + 64 j, 16-B chunk lc); out-of-range rows re-read the last valid row (masked later)
  const int lc = tid & 3, lr = tid >> 2;
  const float* ap[2];
  const float* bp[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) ap[j] = queries + (int64_t)min(q0 + lr + 64 * j, nq - 1) * dim + lc * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) bp[j] = db + (int64_t)min(r0 + lr + 64 * j, rows - 1) * dim + lc * 4;
  int sa[2], sb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) sa[j] = ix_slot(lr + 64 * j, lc);
#pragma unroll
  for (int j = 0; j < 4; ++j) sb[j] = ix_slot(lr + 64 * j, lc);

  acc16 acc[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

  ix_f4 ra[2], rb[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) ra[j] = *(const ix_f4*)(ap[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) rb[j] = *(const ix_f4*)(bp[j]);
#pragma unroll
  for (int j = 0; j < 2; ++j) sA[0][sa[j]] = ra[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) sB[0][sb[j]] = rb[j];
  __syncthreads();

  const int nk = dim / IX_BK;
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    // the last step re-reads its own stage into the idle buffer: no branch around loads and LDS stores
    const int k1 = min(t + 1, nk - 1) * IX_BK;
#pragma unroll
    for (int j = 0; j < 2; ++j) ix_ld16_async(ra[j], ap[j] + k1);
#pragma unroll
    for (int j = 0; j < 4; ++j) ix_ld16_async(rb[j], bp[j] + k1);
    __builtin_amdgcn_sched_barrier(0);
    // K is consumed in a permuted order (the same for both operands): half-wave h takes floats 8g+4h .. 8g+4h+3 of
    // the stage with ONE 16-byte read; MFMA (g, j) then contracts k = 8g+j (h = 0) and 8g+4+j (h = 1)
    ix_f4 fa[2][2], fb[4][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt][g] = sA[cur][ix_slot(wm * 64 + mt * 32 + r, 2 * g + h)];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) fb[nt][g] = sB[cur][ix_slot(wn * 128 + nt * 32 + r, 2 * g + h)];
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][g][j], fb[nt][g][j], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    ix_wait_loads(ra[0], ra[1], rb[0], rb[1], rb[2], rb[3]);
#pragma unroll
    for (int j = 0; j < 2; ++j) sA[cur ^ 1][sa[j]] = ra[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) sB[cur ^ 1][sb[j]] = rb[j];
    __syncthreads();
  }

  // acc[mt][nt][i] = <query q0 + wm*64 + mt*32 + (i/4)*8 + h*4 + i%4 , passage r0 + wn*128 + nt*32 + r>
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = q0 + wm * 64 + mt * 32 + (i >> 2) * 8 + h * 4 + (i & 3);
      const bool qok = q < nq;
      const float th = (qok && !DENSE) ? tau[q] : -INFINITY;
      float* cs = cand_score + (int64_t)min(q, nq - 1) * cap;
      int* ci = cand_id + (int64_t)min(q, nq - 1) * cap;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int row = r0 + wn * 128 + nt * 32 + r;
        const float s = acc[mt][nt][i];
        if (DENSE) {
          if (qok && row < rows) { cs[row] = s; ci[row] = id_base + row; }
          continue;
        }
        const bool pred = qok && row < rows && s > th;
        const uint64_t mask = __ballot(pred);
        if (mask == 0) continue;
        const uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
        const uint32_t mh = h ? hi : lo;
        int base = 0;
        if (pred && r == __builtin_ctz(mh)) base = atomicAdd(&cand_count[q], __builtin_popcount(mh));
        const int blo = lo ? __builtin_amdgcn_readlane(base, __builtin_ctz(lo | 0x80000000u)) : 0;
        const int bhi = hi ? __builtin_amdgcn_readlane(base, 32 + __builtin_ctz(hi | 0x80000000u)) : 0;
        if (pred) {
          const int64_t pos = (int64_t)(h ? bhi : blo) + __builtin_popcount(mh & ((1u << r) - 1u));
          if (pos < cap) { cs[pos] = s; ci[pos] = id_base + row; }
        }
      }
    }
  }
}

// ---- exact top-k of one query's candidate list ---------------------------------------------------------------
__device__ __forceinline__ void ix_bitonic_desc(uint64_t* buf, int P, int tid, int nthreads) {
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (P >> 1); i += nthreads) {
        const int lo = 2 * i - (i & (stride - 1));   // index with the `stride` bit clear
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const uint64_t a = buf[lo], b = buf[hi];
        if ((a < b) == desc) { buf[lo] = b; buf[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// k-th largest among the entries whose key matches `prefix` above `shift+bits`: fills hist, returns the bin and how many
// of the k are still to be found inside it.  One call per radix level.
template <int BITS>
__device__ __forceinline__ void ix_radix_level(const float* cs, int n, uint32_t prefix, int prefix_shift, int shift,
                                               int* hist, int* part, int* sh, int tid, int nthreads,
                                               const float* ce = nullptr) {
  constexpr int BINS = 1 << BITS;
  for (int i = tid; i < BINS; i += nthreads) hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += nthreads) {
    const uint32_t key = ix_key(ce ? cs[i] - ce[i] : cs[i]);
    if (prefix_shift >= 32 || (key >> prefix_shift) == prefix) atomicAdd(&hist[(key >> shift) & (BINS - 1)], 1);
  }
  __syncthreads();
  // sums of groups of 4 bins, their suffix sums (from the top bin down) by doubling, then the one thread whose group
  // holds the wanted rank walks its 4 bins.  (A single thread walking down from the top bin took ~130 dependent LDS
  // reads per level for ordinary positive scores: 15-20 us per call.)
  constexpr int NG = BINS / 4;
  for (int i = tid; i < NG; i += nthreads) part[i] = hist[4 * i] + hist[4 * i + 1] + hist[4 * i + 2] + hist[4 * i + 3];
  __syncthreads();
  for (int off = 1; off < NG; off <<= 1) {
    const int v = (tid < NG && tid + off < NG) ? part[tid + off] : 0;
    __syncthreads();
    if (tid < NG) part[tid] += v;
    __syncthreads();
  }
  const int want = sh[0];       // how many of the best are still wanted at this level (>= 1)
  __syncthreads();
  if (tid < NG) {
    const int incl = part[tid], excl = (tid + 1 < NG) ? part[tid + 1] : 0;
    if (excl < want && want <= incl) {
      int need = want - excl;
      int b = 4 * tid + 3;
      while (b > 4 * tid && hist[b] < need) { need -= hist[b]; --b; }
      sh[0] = need;
      sh[1] = b;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void topk_compact_kernel(float* __restrict__ cand_score, int* __restrict__ cand_id,
                                                          int* __restrict__ cand_count, float* __restrict__ tau,
                                                          int64_t cap, int k, int dense_n) {
  __shared__ uint64_t buf[IX_SORT];
  __shared__ int hist[2048];
  __shared__ int part[512];
  __shared__ int sh[4];
  const int q = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
  float* cs = cand_score + (int64_t)q * cap;
  int* ci = cand_id + (int64_t)q * cap;
  int n = dense_n >= 0 ? dense_n : cand_count[q];
  if ((int64_t)n > cap) n = (int)cap;
  int m;   // entries placed in buf
  if (n <= IX_SORT) {
    for (int i = tid; i < n; i += nthreads) buf[i] = ix_comp(cs[i], ci[i]);
    m = n;
  } else {
    // exact key T of the k-th best by three histogram passes over the list (11 + 11 + 10 bits)
    if (tid == 0) sh[0] = k;
    __syncthreads();
    ix_radix_level<11>(cs, n, 0, 32, 21, hist, part, sh, tid, nthreads);
    const uint32_t b1 = (uint32_t)sh[1];
    __syncthreads();
    ix_radix_level<11>(cs, n, b1, 21, 10, hist, part, sh, tid, nthreads);
    const uint32_t p2 = (b1 << 11) | (uint32_t)sh[1];
    __syncthreads();
    ix_radix_level<10>(cs, n, p2, 10, 0, hist, part, sh, tid, nthreads);
    const uint32_t T = (p2 << 10) | (uint32_t)sh[1];
    __syncthreads();
    // everything above T (fewer than k entries), then the entries equal to T while the sort capacity lasts
    if (tid == 0) { sh[2] = 0; sh[3] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += nthreads) {
      const float s = cs[i];
      if (ix_key(s) > T) buf[atomicAdd(&sh[2], 1)] = ix_comp(s, ci[i]);
    }
    __syncthreads();
    const int above = sh[2];
    for (int i = tid; i < n; i += nthreads) {
      const float s = cs[i];
      if (ix_key(s) == T) {
        const int p = above + atomicAdd(&sh[3], 1);
        if (p < IX_SORT) buf[p] = ix_comp(s, ci[i]);
      }
    }
    __syncthreads();
    m = min(above + sh[3], IX_SORT);
  }
  int P = 2;
  while (P < m) P <<= 1;
  for (int i = m + tid; i < P; i += nthreads) buf[i] = 0;   // below every real entry
  __syncthreads();
  ix_bitonic_desc(buf, P, tid, nthreads);
  const int keep = min(m, k);
  for (int i = tid; i < keep; i += nthreads) {
    const uint64_t c = buf[i];
    cs[i] = ix_unkey((uint32_t)(c >> 32));
    ci[i] = (int)(~(uint32_t)c);
  }
  if (tid == 0) {
    cand_count[q] = keep;
    tau[q] = keep >= k ? ix_unkey((uint32_t)(buf[k - 1] >> 32)) : -INFINITY;
  }
}

__global__ void index_begin_kernel(int* cand_count, float* tau, int nq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) { cand_count[i] = 0; tau[i] = -INFINITY; }
}

__global__ void index_finish_kernel(const float* cand_score, const int* cand_id, const int* cand_count, int64_t cap,
                                    int nq, int k, float* out_score, int64_t* out_id) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)nq * k) return;
  const int q = (int)(i / k), j = (int)(i % k);
  const bool have = j < cand_count[q];
  // faiss pads a short result with label -1 and the lowest float
  out_score[i] = have ? cand_score[(int64_t)q * cap + j] : -3.4028234663852886e38f;
  out_id[i] = have ? (int64_t)cand_id[(int64_t)q * cap + j] : -1;
}


// ---- coarse (bf16) scan with error margins, then exact rescoring ----------------------------------------------
// The scan above spends its time in fp32 MFMAs (157 TFLOP/s peak).  The same search with the SCORING done in bf16
// (2.5 PFLOP/s class, the training GEMM of gemm_bf16.hip with a filter epilogue) and exactness kept by interval
// arithmetic: a bf16 score c of query q and row d differs from the true inner product s by at most
//   E(q, d) = IX_EPS * |q| * |d|      (two roundings to 8 significant bits: 2^-7 (1 + 2^-9) sum|q_i d_i| <= ... |q||d| by
//                                      Cauchy-Schwarz, plus the fp32 accumulation of 768 terms; IX_EPS = 0.008)
// so s lies in [c - E, c + E].  LB(q) = k-th largest lower bound seen so far is a lower bound of the final k-th score;
// a row can only be in the top k if its upper bound reaches LB.  The scan keeps every such row (usually a few hundred
// per query), the survivors are re-scored exactly in fp32 and the exact top k is selected as before: same result as
// the fp32 scan, the matrix cores run ~6x faster.  A query whose survivor list outgrows its buffer is flagged and
// must be searched with the fp32 scan instead.
constexpr float IX_EPS = 0.008f;
constexpr int IX_KEEP = 16384;   // survivors a query may carry from chunk to chunk

__global__ __launch_bounds__(256) void index_prepare_kernel(const float* __restrict__ x, int64_t rows, int64_t rows_pad, int dim,
                                                            float scale, bf16_t* __restrict__ xb, float* __restrict__ norms) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows_pad) return;
  float ss = 0.f;
  for (int d = lane * 4; d < dim; d += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) v = *(const float4*)(x + row * dim + d);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    uint2 o;
    o.x = pack2bf(v.x, v.y);
    o.y = pack2bf(v.z, v.w);
    *(uint2*)(xb + row * dim + d) = o;
  }
  ss = wave_sum(ss);
  if (lane == 0) norms[row] = scale * sqrtf(ss);
}

// first chunk of a search: the GEMM wrote c densely into cand_score[q][0..n); fill in the margins and the ids
__global__ void coarse_fill_kernel(const float* __restrict__ qe, const float* __restrict__ dn, int n, int id_base, int64_t cap,
                                   float* __restrict__ ce, int* __restrict__ ci) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (j >= n) return;
  ce[(int64_t)q * cap + j] = qe[q] * dn[j];
  ci[(int64_t)q * cap + j] = id_base + j;
}

// one workgroup per query: raise LB to the k-th largest lower bound, drop the entries whose upper bound is below it
__global__ __launch_bounds__(512) void margin_compact_kernel(float* __restrict__ cand_score, float* __restrict__ cand_err,
                                                            int* __restrict__ cand_id, int* __restrict__ cand_count,
                                                            float* __restrict__ lb, int* __restrict__ flags,
                                                            int* __restrict__ kept, int64_t cap, int k, int dense_n, int force) {
  __shared__ int hist[2048];
  __shared__ int part[512];
  __shared__ int sh[4];
  const int q = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
  float* cs = cand_score + (int64_t)q * cap;
  float* ce = cand_err + (int64_t)q * cap;
  int* ci = cand_id + (int64_t)q * cap;
  int n = dense_n >= 0 ? dense_n : cand_count[q];
  // the selection below costs ~10x the chunk's few appends: leave the list alone until it has doubled since it was last
  // compacted (a stale bound is still a valid bound, it only lets a few more rows through) — unless the next chunk's
  // worst case would no longer fit, or this is the last chunk
  if (dense_n < 0 && !force && n <= IX_KEEP && n < 2 * kept[q] + 64) return;
  bool overflow = false;
  if ((int64_t)n > cap) { n = (int)cap; overflow = true; }
  float bound = lb[q];
  if (n >= k) {
    if (tid == 0) sh[0] = k;
    __syncthreads();
    ix_radix_level<11>(cs, n, 0, 32, 21, hist, part, sh, tid, nthreads, ce);
    const uint32_t b1 = (uint32_t)sh[1];
    __syncthreads();
    ix_radix_level<11>(cs, n, b1, 21, 10, hist, part, sh, tid, nthreads, ce);
    const uint32_t p2 = (b1 << 11) | (uint32_t)sh[1];
    __syncthreads();
    ix_radix_level<10>(cs, n, p2, 10, 0, hist, part, sh, tid, nthreads, ce);
    bound = fmaxf(bound, ix_unkey((p2 << 10) | (uint32_t)sh[1]));
    __syncthreads();
  }
  // in-place compaction, a strip of `nthreads` entries at a time: a strip is read, then its survivors are written at or
  // below where they were read
  if (tid == 0) sh[2] = 0;
  __syncthreads();
  for (int s0 = 0; s0 < n; s0 += nthreads) {
    const int i = s0 + tid;
    float c = 0.f, e = 0.f;
    int id = 0;
    bool keep = false;
    if (i < n) { c = cs[i]; e = ce[i]; id = ci[i]; keep = c + e >= bound; }
    __syncthreads();
    if (keep) {
      const int p = atomicAdd(&sh[2], 1);
      cs[p] = c; ce[p] = e; ci[p] = id;
    }
    __syncthreads();
  }
  if (tid == 0) {
    int m = sh[2];
    if (m > IX_KEEP) { m = IX_KEEP; overflow = true; }
    cand_count[q] = m;
    kept[q] = m;
    lb[q] = bound;
    if (overflow) flags[q] = 1;
  }
}

// exact fp32 score of every surviving entry: one wave per entry
__global__ __launch_bounds__(256) void rescore_kernel(const float* const* __restrict__ blocks, const int64_t* __restrict__ starts,
                                                      int nblocks, int dim, const float* __restrict__ queries,
                                                      float* __restrict__ cand_score, const int* __restrict__ cand_id,
                                                      const int* __restrict__ cand_count, int64_t cap) {
  const int q = blockIdx.y, lane = threadIdx.x & 63;
  const int n = min((int64_t)cand_count[q], cap);
  const float* qv = queries + (int64_t)q * dim;
  for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
    const int64_t id = cand_id[(int64_t)q * cap + e];
    int b = 0;
    while (b + 1 < nblocks && id >= starts[b + 1]) ++b;
    const float* row = blocks[b] + (id - starts[b]) * dim;
    float s = 0.f;
    for (int d = lane * 4; d < dim; d += 256) {
      const float4 a = *(const float4*)(qv + d), v = *(const float4*)(row + d);
      s = fmaf(a.x, v.x, s); s = fmaf(a.y, v.y, s); s = fmaf(a.z, v.z, s); s = fmaf(a.w, v.w, s);
    }
    s = wave_sum(s);
    if (lane == 0) cand_score[(int64_t)q * cap + e] = s;
  }
}

__global__ void coarse_begin_kernel(int* cand_count, float* lb, int* flags, int* kept, int nq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) { cand_count[i] = 0; lb[i] = -INFINITY; flags[i] = 0; kept[i] = 0; }
}

static int ix_kpad(int k) { int p = 2; while (p < k) p <<= 1; return p; }

}  // namespace czc

using namespace czc;

extern "C" int64_t czc_index_candidate_capacity(int32_t k) {
  if (k < 1 || k > IX_KMAX) return -1;
  return (int64_t)ix_kpad(k) + CZC_INDEX_CHUNK;
}

extern "C" int czc_index_begin(int32_t* cand_count, float* tau, int32_t nq, czc_stream_t stream) {
  CZC_REQUIRE(nq > 0 && cand_count && tau, "czc_index_begin: nq=%d", nq);
  hipLaunchKernelGGL(index_begin_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, (hipStream_t)stream, cand_count, tau, nq);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_index_scan(const float* db, int64_t rows, int32_t dim, int64_t id_base, const float* queries,
                              int32_t nq, int32_t k, int32_t fresh, float* cand_score, int32_t* cand_id,
                              int32_t* cand_count, float* tau, int64_t cap, czc_stream_t stream) {
  CZC_REQUIRE(db && queries && cand_score && cand_id && cand_count && tau, "czc_index_scan: null pointer");
  CZC_REQUIRE(rows > 0 && nq > 0, "czc_index_scan: rows=%lld nq=%d", (long long)rows, nq);
  CZC_REQUIRE(dim > 0 && dim % IX_BK == 0, "czc_index_scan: dim=%d must be a multiple of %d", dim, IX_BK);
  CZC_REQUIRE(k >= 1 && k <= IX_KMAX, "czc_index_scan: k=%d outside 1..%d", k, IX_KMAX);
  CZC_REQUIRE(cap >= czc_index_candidate_capacity(k), "czc_index_scan: candidate capacity %lld < %lld", (long long)cap,
              (long long)czc_index_candidate_capacity(k));
  CZC_REQUIRE(id_base >= 0 && id_base + rows <= 0x7fffffffLL, "czc_index_scan: passage rows beyond 2^31-1");
  hipStream_t st = (hipStream_t)stream;
  const int q_tiles = ceil_div(nq, IX_TQ);
  for (int64_t row0 = 0; row0 < rows; row0 += CZC_INDEX_CHUNK) {
    const int n = (int)((rows - row0 < CZC_INDEX_CHUNK) ? rows - row0 : CZC_INDEX_CHUNK);
    const int r_tiles = ceil_div(n, IX_TR);
    const int dense = (fresh && row0 == 0) ? 1 : 0;
    const int grid = q_tiles * ((r_tiles + 7) / 8) * 8;
    auto kern = dense ? ip_scan_kernel<true> : ip_scan_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, db + row0 * dim, n, dim, (int)(id_base + row0), queries, nq,
                       tau, cand_score, cand_id, cand_count, cap, q_tiles, r_tiles);
    CZC_LAUNCH_CHECK();
    hipLaunchKernelGGL(topk_compact_kernel, dim3(nq), dim3(512), 0, st, cand_score, cand_id, cand_count, tau, cap, k,
                       dense ? n : -1);
    CZC_LAUNCH_CHECK();
  }
  return CZC_OK;
}

extern "C" int czc_index_finish(const float* cand_score, const int32_t* cand_id, const int32_t* cand_count, int64_t cap,
                                int32_t nq, int32_t k, float* out_score, int64_t* out_id, czc_stream_t stream) {
  CZC_REQUIRE(cand_score && cand_id && cand_count && out_score && out_id, "czc_index_finish: null pointer");
  CZC_REQUIRE(nq > 0 && k >= 1 && k <= IX_KMAX, "czc_index_finish: nq=%d k=%d", nq, k);
  const int64_t total = (int64_t)nq * k;
  hipLaunchKernelGGL(index_finish_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     cand_score, cand_id, cand_count, cap, nq, k, out_score, out_id);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

// ---- coarse scan + rescoring entry points ------------------------------------------------------------------------
extern "C" int64_t czc_index_coarse_capacity(int32_t k) {
  if (k < 1 || k > IX_KMAX) return -1;
  return (int64_t)IX_KEEP + CZC_INDEX_CHUNK;
}

extern "C" int czc_index_prepare(const float* x, int64_t rows, int32_t dim, float scale, uint16_t* x_bf16, float* norms,
                                 czc_stream_t stream) {
  CZC_REQUIRE(x && x_bf16 && norms && rows > 0, "czc_index_prepare: null pointer or rows=%lld", (long long)rows);
  CZC_REQUIRE(dim > 0 && dim % IX_BK == 0, "czc_index_prepare: dim=%d must be a multiple of %d", dim, IX_BK);
  const int64_t rows_pad = (rows + 7) / 8 * 8;
  hipLaunchKernelGGL(index_prepare_kernel, dim3((unsigned)ceil_div64(rows_pad, 4)), dim3(256), 0, (hipStream_t)stream, x, rows,
                     rows_pad, dim, scale, x_bf16, norms);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" float czc_index_coarse_epsilon(void) { return IX_EPS; }

extern "C" int czc_index_coarse_begin(int32_t* cand_count, float* lower_bound, int32_t* flags, int32_t* kept, int32_t nq,
                                      czc_stream_t stream) {
  CZC_REQUIRE(nq > 0 && cand_count && lower_bound && flags && kept, "czc_index_coarse_begin: nq=%d", nq);
  hipLaunchKernelGGL(coarse_begin_kernel, dim3(ceil_div(nq, 256)), dim3(256), 0, (hipStream_t)stream, cand_count, lower_bound, flags,
                     kept, nq);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_index_coarse_scan(const uint16_t* db_bf16, const float* db_norms, int64_t rows, int32_t dim, int64_t id_base,
                                     const uint16_t* q_bf16, const float* q_margin, int32_t nq, int32_t k, int32_t fresh,
                                     float* cand_score, float* cand_err, int32_t* cand_id, int32_t* cand_count,
                                     float* lower_bound, int32_t* flags, int32_t* kept, int64_t cap, czc_stream_t stream) {
  CZC_REQUIRE(db_bf16 && db_norms && q_bf16 && q_margin && cand_score && cand_err && cand_id && cand_count && lower_bound && flags && kept,
              "czc_index_coarse_scan: null pointer");
  CZC_REQUIRE(rows > 0 && nq > 0, "czc_index_coarse_scan: rows=%lld nq=%d", (long long)rows, nq);
  CZC_REQUIRE(dim > 0 && dim % IX_BK == 0, "czc_index_coarse_scan: dim=%d must be a multiple of %d", dim, IX_BK);
  CZC_REQUIRE(k >= 1 && k <= IX_KMAX, "czc_index_coarse_scan: k=%d outside 1..%d", k, IX_KMAX);
  CZC_REQUIRE(cap >= czc_index_coarse_capacity(k) && cap % 8 == 0 && cap < 0x7fffffffLL, "czc_index_coarse_scan: candidate capacity %lld",
              (long long)cap);
  CZC_REQUIRE(id_base >= 0 && id_base + rows <= 0x7fffffffLL, "czc_index_coarse_scan: passage rows beyond 2^31-1");
  hipStream_t st = (hipStream_t)stream;
  // The first chunk of a search is written densely and selected from in full (4 passes over n entries per query): it is
  // kept short; after it the lower bounds exist and a chunk costs little more than its GEMM.  The last chunk of a call
  // always compacts, so that czc_index_rescore re-scores survivors only.
  constexpr int FIRST = 16384;
  int64_t row0 = 0;
  while (row0 < rows) {
    const int dense = (fresh && row0 == 0) ? 1 : 0;
    const int64_t lim = dense ? FIRST : CZC_INDEX_CHUNK;
    const int n = (int)((rows - row0 < lim) ? rows - row0 : lim);
    const int n_pad = (n + 7) / 8 * 8;          // the block's bf16 image is padded to 8 rows (czc_index_prepare)
    GemmArgs a{};
    a.A = q_bf16; a.B = db_bf16 + row0 * dim;
    a.M = nq; a.N = n_pad; a.K = dim;
    a.lda = dim; a.ldb = dim;
    int rc;
    if (dense) {
      a.C = cand_score; a.ldc = (int)cap;
      rc = gemm_dispatch(0, 0, EPI_F32, a, 1, st);
      if (rc != CZC_OK) return rc;
      hipLaunchKernelGGL(coarse_fill_kernel, dim3(ceil_div(n, 256), nq), dim3(256), 0, st, q_margin, db_norms + row0, n,
                         (int)(id_base + row0), cap, cand_err, cand_id);
    } else {
      a.ldc = 8;
      a.f_lb = lower_bound; a.f_qe = q_margin; a.f_dn = db_norms + row0;
      a.f_cs = cand_score; a.f_ce = cand_err; a.f_ci = cand_id; a.f_cnt = cand_count;
      a.f_cap = cap; a.f_idbase = (int)(id_base + row0); a.f_nvalid = n;
      rc = gemm_dispatch(0, 0, EPI_FILTER, a, 1, st);
      if (rc != CZC_OK) return rc;
    }
    const int last = (row0 + n >= rows) ? 1 : 0;
    hipLaunchKernelGGL(margin_compact_kernel, dim3(nq), dim3(512), 0, st, cand_score, cand_err, cand_id, cand_count, lower_bound,
                       flags, kept, cap, k, dense ? n : -1, last);
    CZC_LAUNCH_CHECK();
    row0 += n;
  }
  return CZC_OK;
}

extern "C" int czc_index_rescore(const float* const* blocks, const int64_t* block_starts, int32_t nblocks, int32_t dim,
                                 const float* queries, int32_t nq, int32_t k, float* cand_score, int32_t* cand_id,
                                 int32_t* cand_count, float* tau, int64_t cap, czc_stream_t stream) {
  CZC_REQUIRE(blocks && block_starts && queries && cand_score && cand_id && cand_count && tau, "czc_index_rescore: null pointer");
  CZC_REQUIRE(nblocks >= 1 && nq > 0 && dim > 0 && dim % 4 == 0 && k >= 1 && k <= IX_KMAX, "czc_index_rescore: nblocks=%d nq=%d dim=%d k=%d",
              nblocks, nq, dim, k);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rescore_kernel, dim3(64, nq), dim3(256), 0, st, blocks, block_starts, nblocks, dim, queries, cand_score,
                     cand_id, cand_count, cap);
  hipLaunchKernelGGL(topk_compact_kernel, dim3(nq), dim3(512), 0, st, cand_score, cand_id, cand_count, tau, cap, k, -1);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}
