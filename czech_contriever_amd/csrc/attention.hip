// attention.hip — BERT self-attention (head_dim 64, key-padding mask, prob-dropout) for gfx950.
//
// Replaces HF BertSelfAttention's softmax(QK^T/8 + mask) V and its autograd backward
// (HF modeling_bert.py eager_attention_forward; call site src/contriever.py:32-43).
// Layout: qkv bf16 [B*S, 3H] straight out of the fused QKV GEMM (Q | K | V, head h at columns
// 64h), ctx bf16 [B*S, H] straight into the output-projection GEMM: no head transposes in HBM.
//
// All three kernels are "query/key on the lane": with v_mfma_f32_32x32x16_bf16 the score tile is
// produced TRANSPOSED so that its softmax statistics (running max / sum, lse, delta) live in the
// lane that owns the column, never in another lane, and the bf16 probabilities are fed to the
// second product directly from the accumulator registers (no LDS round trip):
//   forward   S^T = K Q^T  (key rows x query cols)  ->  O^T += V^T P^T        per 64-key block
//   dQ kernel S^T, dP^T = V dO^T                     ->  dQ^T += K^T dS^T
//   dKdV      S = Q K^T, dP = dO V^T (query rows x key cols) -> dV^T += dO^T P~, dK^T += Q^T dS
// K/V (or Q/dO) blocks are staged in LDS as [rows][64] bf16 with 16-B chunks XOR-swizzled by
// swz64(row) so that both the ds_read_b128 row reads and the ds_read_b64_tr_b16 column reads are
// bank-conflict free.  Sequences longer than 256 keep a two-kernel backward (dQ; dK + dV: S, P~ and dS are rebuilt
// in each); up to 256 tokens ONE workgroup per (batch, head) makes a single pass (attn_bwd_fused_kernel).
#include <stdlib.h>

#include "czc_common.h"

namespace czc {

// value of lane J of each aligned group of four lanes, in all four (DPP quad_perm [J, J, J, J])
template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, J * 0x55, 0xf, 0xf, true);
}


constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_MASK = -2.0e30f;  // masked score (log2 domain); running max starts at -1e30

__device__ __forceinline__ int swz64(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// split staging (issue early / write late): global -> registers now, registers -> swizzled LDS later
template <int ROWS>
struct TileRegs { uint4 v[ROWS * 8 / 256]; };
template <int ROWS>
__device__ __forceinline__ void tile_fetch(TileRegs<ROWS>& t, const bf16_t* __restrict__ src, int ld, int nvalid, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int id = tid + 256 * i, row = id >> 3, c = id & 7;
    t.v[i] = (uint4){0u, 0u, 0u, 0u};
    if (row < nvalid) t.v[i] = *(const uint4*)(src + (size_t)row * ld + c * 8);
  }
}
template <int ROWS>
__device__ __forceinline__ void tile_commit(const TileRegs<ROWS>& t, unsigned char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int id = tid + 256 * i, row = id >> 3, c = id & 7;
    *(uint4*)(lds + row * 128 + ((c ^ swz64(row)) << 4)) = t.v[i];
  }
}

// 32x32x16 A/B operand by rows: lane (r = lane&31, h = lane>>5) holds tile[row0 + r][16ks + 8h .. +7]
__device__ __forceinline__ bf16x8 frag_row(const unsigned char* lds, int row0, int ks, int lane) {
  const int row = row0 + (lane & 31), c = 2 * ks + (lane >> 5);
  return *(const bf16x8*)(lds + row * 128 + ((c ^ swz64(row)) << 4));
}
// 32x32x16 operand by columns (transposing read), in the k-order of an accumulator-fed partner:
// lane (r, h) element j = tile[rbase + 8(j>>2) + 4h + (j&3)][col0 + r]      (rbase multiple of 16)
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* lds, int rbase, int col0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = g >> 1;
  const int c = (col0 >> 3) + 2 * (g & 1) + (p >> 1);
  const int row1 = rbase + 4 * h + q, row2 = row1 + 8;
  typedef s16x4 __attribute__((address_space(3)))* trp_t;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(lds + row1 * 128 + ((c ^ swz64(row1)) << 4) + 8 * (p & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(lds + row2 * 128 + ((c ^ swz64(row2)) << 4) + 8 * (p & 1)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// registers 8s .. 8s+7 of an accumulator tile as the bf16 operand of k-step s: a vector conversion compiles to
// exactly four v_cvt_pk_bf16_f32 (element-wise casts made hipcc add ~4 moves / sdwa ORs per pair; inline asm
// would hide the VALU-write -> MFMA-read hazard from the compiler and needs manual s_nops)
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
  typedef __attribute__((ext_vector_type(8))) float f32x8;
  const f32x8 v = {a[8 * s], a[8 * s + 1], a[8 * s + 2], a[8 * s + 3], a[8 * s + 4], a[8 * s + 5], a[8 * s + 6], a[8 * s + 7]};
  return __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ bf16x8 ld_frag_global(const bf16_t* p) {
  const uint4 v = *(const uint4*)p;
  return __builtin_bit_cast(bf16x8, v);
}
// dropout element index of the probability (b, head, q, key): ((b*heads + head)*S + q) * S4 + key with
// S4 = S rounded up to 4, so that 4 consecutive keys starting at a multiple of 4 are exactly one RNG quad
__device__ __forceinline__ void keep4(uint32_t key, uint32_t e, uint32_t thr, float inv_keep, float (&z)[4]) {
  const czc_quad q = czc_rng_quad(key, e >> 2);
  z[0] = ((q.lo & 0xffffu) >= thr) ? inv_keep : 0.f;
  z[1] = (q.lo >= (thr << 16)) ? inv_keep : 0.f;
  z[2] = ((q.hi & 0xffffu) >= thr) ? inv_keep : 0.f;
  z[3] = (q.hi >= (thr << 16)) ? inv_keep : 0.f;
}

struct AttnArgs {
  const bf16_t* qkv; const uint8_t* mask; const bf16_t* ctx; const bf16_t* dctx;
  const float* lse_in; float* lse; float* delta; bf16_t* out; bf16_t* dqkv;
  float* dbias;  // per-workgroup partial column sums of dqkv: [gridDim.z * gridDim.x][3H] (null = not wanted)
  // dropout keep bits of the probabilities, seq <= 256 (null = not kept): word [(b * heads + head) * 8 + q / 32][pos(key)],
  // pos(k) = 32 (k / 32) + 8 ((k % 32) / 8) + 2 (k % 4) + (k / 4) % 2, has bit (q % 32) set iff probability (q, key) is kept.  The resident forward writes them from the compare masks it
  // forms anyway, the streaming backward reads one word per lane and step instead of hashing again.
  uint32_t* bits;
  int B, S, nh, H;
  uint32_t thr, seed, site;
  float inv_keep;
};

// key-padding bias (log2 domain) for the whole (padded) sequence; padflag[t] != 0 iff 32-key tile t holds a masked
// key (most tiles do not: they take a path without the bias add)
__device__ __forceinline__ void load_mask_bias(const AttnArgs& a, int b, float* mb, int* padflag, int spad, int tid) {
  if (tid < 16) padflag[tid] = 0;
  __syncthreads();
  for (int s = tid; s < spad; s += 256) {
    const bool valid = s < a.S && a.mask[(size_t)b * a.S + s];
    mb[s] = valid ? 0.f : NEG_MASK;
    if (!valid) padflag[s >> 5] = 1;
  }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char kt[2][64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char vt[2][64 * 128];
  __shared__ __attribute__((aligned(16))) float mb[512];
  __shared__ int padflag[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, hh = blockIdx.y;
  const int S = a.S, H3 = 3 * a.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int r = lane & 31, h = lane >> 5;
  const int spad = (S + 63) & ~63;
  const bool active = q0 < S;  // wave-uniform
  const float c2 = 0.125f * LOG2E;

  load_mask_bias(a, b, mb, padflag, spad, tid);

  const bf16_t* base = a.qkv + (size_t)b * S * H3 + hh * 64;
  bf16x8 bq[4];
  {
    int qr = q0 + r; if (qr > S - 1) qr = S - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[ks] = ld_frag_global(base + (size_t)qr * H3 + 16 * ks + 8 * h);
  }
  f32x16 o[2];
  o[0] = o[1] = (f32x16)(0.f);
  float m = -1.0e30f, l = 0.f;
  const uint32_t S4 = (uint32_t)((S + 3) & ~3);
  const uint32_t rkey = czc_rng_key(a.seed, a.site);
  const uint32_t erow = DROP ? ((uint32_t)((b * a.nh + hh) * S + (q0 + r)) * S4) : 0u;

  TileRegs<64> kr, vr;
  tile_fetch<64>(kr, base + a.H, H3, S, tid);
  tile_fetch<64>(vr, base + 2 * a.H, H3, S, tid);
  tile_commit<64>(kr, kt[0], tid);
  tile_commit<64>(vr, vt[0], tid);
  int buf = 0;
  for (int kb0 = 0; kb0 < S; kb0 += 64, buf ^= 1) {
    __syncthreads();   // tile kb0 is visible; every wave is done with the other buffer
    const bool more = kb0 + 64 < S;
    if (more) {
      tile_fetch<64>(kr, base + a.H + (size_t)(kb0 + 64) * H3, H3, S - kb0 - 64, tid);
      tile_fetch<64>(vr, base + 2 * a.H + (size_t)(kb0 + 64) * H3, H3, S - kb0 - 64, tid);
    }
    if (active) {
#pragma unroll 1
      for (int tt = 0; tt < 2; ++tt) {
        f32x16 s = (f32x16)(0.f);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(kt[buf], 32 * tt, ks, lane), bq[ks], s, 0, 0, 0);
        float mx, alpha, ps = 0.f;
        if (padflag[(kb0 >> 5) + tt]) {          // wave-uniform: this tile holds masked keys
          mx = m;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 bias = *(const f32x4*)(mb + kb0 + 32 * tt + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float x = s[4 * g + j] * c2 + bias[j];
              s[4 * g + j] = x;
              mx = fmaxf(mx, x);
            }
          }
          mx = xor32_max(mx);
          alpha = __builtin_amdgcn_exp2f(m - mx);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(s[i] - mx);
            s[i] = p;
            ps += p;
          }
        } else {                                  // no padding: max on the raw scores, one fma + one exp per element
          float rm = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
          for (int i = 3; i < 15; i += 2) rm = fmaxf(fmaxf(rm, s[i]), s[i + 1]);
          rm = fmaxf(rm, s[15]);
          mx = fmaxf(m, rm * c2);
          mx = xor32_max(mx);
          alpha = __builtin_amdgcn_exp2f(m - mx);
          const float nmx = -mx;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[i], c2, nmx));
            s[i] = p;
            ps += p;
          }
        }
        m = mx;
        l = l * alpha + ps;
        if (__any(alpha != 1.0f)) {   // exact: when no row's running max moved, alpha == 1 in every lane
          o[0] *= alpha;
          o[1] *= alpha;
        }
        if (DROP) {   // drop = select 0; the 1/(1-p) scale is applied once, to the final O
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const czc_quad qd = czc_rng_quad(rkey, (erow + (uint32_t)(kb0 + 32 * tt + 8 * g + 4 * h)) >> 2);
            s[4 * g + 0] = ((qd.lo & 0xffffu) >= a.thr) ? s[4 * g + 0] : 0.f;
            s[4 * g + 1] = (qd.lo >= (a.thr << 16)) ? s[4 * g + 1] : 0.f;
            s[4 * g + 2] = ((qd.hi & 0xffffu) >= a.thr) ? s[4 * g + 2] : 0.f;
            s[4 * g + 3] = (qd.hi >= (a.thr << 16)) ? s[4 * g + 3] : 0.f;
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_frag(s, s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vt[buf], 32 * tt + 16 * s2, 32 * dt, lane), pf, o[dt], 0, 0, 0);
        }
      }
    }
    if (more) {
      tile_commit<64>(kr, kt[buf ^ 1], tid);
      tile_commit<64>(vr, vt[buf ^ 1], tid);
    }
  }
  if (!active) return;
  l = xor32_sum(l);
  const float inv = (DROP ? a.inv_keep : 1.0f) / l;
  const int q = q0 + r;
  if (q < S) {
    bf16_t* dst = a.out + ((size_t)b * S + q) * a.H + hh * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack2bf(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
        w.y = pack2bf(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *(uint2*)(dst + 32 * dt + 8 * g + 4 * h) = w;
      }
    if (h == 0) a.lse[((size_t)b * a.nh + hh) * S + q] = (m + __log2f(l)) * LN2;
  }
}

// ---------------------------------------------------------------------------------------------
// forward, sequences of at most 256 keys: one 512-thread workgroup per (batch, head) keeps ALL of K and V in LDS
// (2 x 32 KiB, filled once), so the eight waves (32 queries each) share a single barrier and then sweep the eight
// 32-key tiles independently -- no per-block barrier, no double buffer, K/V read from HBM once instead of once per
// 128 queries, and 128 VGPRs = four waves per SIMD (two workgroups per CU).  The loop body is the streaming kernel's.
// ---------------------------------------------------------------------------------------------
// KEEP (with DROP): the keep words are exported (AttnArgs::bits); a template parameter, not a test of the pointer: the
// pointer test inside the dropout block cost the plain forward 12 % (119 us against 107)
template <bool DROP, bool KEEP>
__global__ __launch_bounds__(512, 4) void attn_fwd_res_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char kt[256 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char vt[256 * 128];
  __shared__ __attribute__((aligned(16))) float mb[256];
  __shared__ int padflag[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, hh = blockIdx.y;
  const int S = a.S, H3 = 3 * a.H;
  const int q0 = wave * 32;
  const int r = lane & 31, h = lane >> 5;
  const int spad = (S + 31) & ~31;
  const bool active = q0 < S;  // wave-uniform
#ifdef ATTF_STAMPS   // diagnostic build only (tools/stamp_attn_fwd.py): s_memtime of wave ATTF_STAMPS / lane 0 and where the workgroup
                     // runs (HW_ID, XCC_ID) into the workgroup's lse row; lse itself is not written in such a build
  long long* stamp_buf = (long long*)(a.lse + ((size_t)b * a.nh + hh) * S);
#define ATTF_STAMP(k) do { if (wave == ATTF_STAMPS && lane == 0) { stamp_buf[k] = __builtin_amdgcn_s_memtime(); if ((k) == 0 || (k) == 6) stamp_buf[8 + ((k) != 0)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
  if (wave == ATTF_STAMPS && lane == 0) {
    stamp_buf[10] = __builtin_amdgcn_s_getreg(0xF804);
    stamp_buf[11] = __builtin_amdgcn_s_getreg(0xF814);
  }
#else
#define ATTF_STAMP(k) do { } while (0)
#endif
  ATTF_STAMP(0);
  const float c2 = 0.125f * LOG2E;
  const bf16_t* base = a.qkv + (size_t)b * S * H3 + hh * 64;

  // Every global load of the prologue is issued before the first wait (ONE memory round trip, one barrier): the K and V
  // rows for the LDS images (rows past S are zero: their scores are masked out below), this wave's Q fragments and the
  // key's mask byte.  (Q used to be fetched after the K / V images had been written and synchronised: a second round
  // trip with every wave of the workgroup parked -- rocprofv3: 43 % of the wave cycles in s_waitcnt / s_barrier.)
  uint4 kv[4], vv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 512 * i, row = id >> 3, c = id & 7;
    kv[i] = vv[i] = (uint4){0u, 0u, 0u, 0u};
    if (row < S) {
      // streaming (nt) reads: K and V are read once per kernel, and without them in its way the context rows this kernel
      // writes stay in the Infinity Cache for the output projection that follows (forward 112.9 -> 110.2 us, the
      // projection 88.1 -> 86.8, interleaved in the step)
      typedef __attribute__((ext_vector_type(4))) unsigned u4v;
      const u4v k_ = __builtin_nontemporal_load((const u4v*)(base + a.H + (size_t)row * H3 + c * 8));
      const u4v v_ = __builtin_nontemporal_load((const u4v*)(base + 2 * a.H + (size_t)row * H3 + c * 8));
      kv[i] = make_uint4(k_[0], k_[1], k_[2], k_[3]);
      vv[i] = make_uint4(v_[0], v_[1], v_[2], v_[3]);
    }
  }
  bf16x8 bq[4];
  {
    int qr = q0 + r; if (qr > S - 1) qr = S - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[ks] = ld_frag_global(base + (size_t)qr * H3 + 16 * ks + 8 * h);
  }
  const bool kvalid = tid < S && a.mask[(size_t)b * S + (tid < S ? tid : 0)];
  __builtin_amdgcn_sched_barrier(0);
  ATTF_STAMP(1);
#ifdef ATTF_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTF_STAMP(2);
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 512 * i, row = id >> 3, c = id & 7;
    *(uint4*)(kt + row * 128 + ((c ^ swz64(row)) << 4)) = kv[i];
    *(uint4*)(vt + row * 128 + ((c ^ swz64(row)) << 4)) = vv[i];
  }
  if (tid < 256) {   // thread t owns key t: waves 0-3; a 32-key tile is one half of a wave
    mb[tid] = kvalid ? 0.f : NEG_MASK;
    const unsigned long long bad = __ballot(!kvalid && tid < spad);
    if (lane == 0) {
      padflag[2 * wave] = (bad & 0xffffffffull) != 0;
      padflag[2 * wave + 1] = (bad >> 32) != 0;
    }
  }
  __syncthreads();
  ATTF_STAMP(3);
  if (!active) return;

  f32x16 o[2];
  o[0] = o[1] = (f32x16)(0.f);
  // Online softmax with a LAZY reference point: m follows the running maximum only when that exceeds it by more than
  // 2^DEFER (log2 units), so most tiles skip the rescaling of O and l (alpha == 1 in every lane: the __any below).
  // Probabilities reach at most 2^DEFER instead of 1: exact in fp32 / bf16 terms (same mantissa, larger exponent), and
  // O / l and lse = m + log2 l do not depend on the reference point.  (Requesting the next tile's scores before this
  // tile's softmax arithmetic was measured too: 113 us against 108, four waves per SIMD hide that latency already.)
  constexpr float DEFER = 8.0f;
  float m = -1.0e30f, l = 0.f;
  const uint32_t S4 = (uint32_t)((S + 3) & ~3);
  const uint32_t rkey = czc_rng_key(a.seed, a.site);
  const uint32_t erow = DROP ? ((uint32_t)((b * a.nh + hh) * S + (q0 + r)) * S4) : 0u;
#pragma unroll 1
  for (int k0 = 0; k0 < spad; k0 += 32) {
    f32x16 s = (f32x16)(0.f);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(kt, k0, ks, lane), bq[ks], s, 0, 0, 0);
    float mx, alpha, ps = 0.f;
    if (padflag[k0 >> 5]) {                    // wave-uniform: this tile holds masked keys
      mx = m;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bias = *(const f32x4*)(mb + k0 + 8 * g + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = s[4 * g + j] * c2 + bias[j];
          s[4 * g + j] = x;
          mx = fmaxf(mx, x);
        }
      }
      mx = xor32_max(mx);
      mx = (mx > m + DEFER) ? mx : m;          // the reference point moves only when the maximum outgrew it by 2^DEFER
      alpha = __builtin_amdgcn_exp2f(m - mx);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(s[i] - mx);
        s[i] = p;
        ps += p;
      }
    } else {                                    // no padding: max on the raw scores, one fma + one exp per element
      float rm = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) rm = fmaxf(fmaxf(rm, s[i]), s[i + 1]);
      rm = fmaxf(rm, s[15]);
      mx = fmaxf(m, rm * c2);
      mx = xor32_max(mx);
      mx = (mx > m + DEFER) ? mx : m;
      alpha = __builtin_amdgcn_exp2f(m - mx);
      const float nmx = -mx;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[i], c2, nmx));
        s[i] = p;
        ps += p;
      }
    }
    m = mx;
    l = l * alpha + ps;
    if (__any(alpha != 1.0f)) {   // exact: when no row's running max moved, alpha == 1 in every lane
      o[0] *= alpha;
      o[1] *= alpha;
    }
    if (DROP) {   // drop = select 0; the 1/(1-p) scale is applied once, to the final O
      // keep words (AttnArgs::bits): the compare mask of register (g, j) IS two of them -- lanes 0-31 = key 8g + j over this
      // wave's 32 queries, lanes 32-63 = key 8g + 4 + j -- and leaves through a scalar store: no vector instruction spent
      // (32 v_writelane per tile had cost the forward what the backward gained).  Word position of key k in its tile of 32:
      // 8 (k / 8) + 2 (k % 4) + (k / 4) % 2.
      const uint32_t* tb = KEEP ? a.bits + ((size_t)(b * a.nh + hh) * 8 + __builtin_amdgcn_readfirstlane(wave)) * 256 + k0 : nullptr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const czc_quad qd = czc_rng_quad(rkey, (erow + (uint32_t)(k0 + 8 * g + 4 * h)) >> 2);
        if (!KEEP) {
          s[4 * g + 0] = ((qd.lo & 0xffffu) >= a.thr) ? s[4 * g + 0] : 0.f;
          s[4 * g + 1] = (qd.lo >= (a.thr << 16)) ? s[4 * g + 1] : 0.f;
          s[4 * g + 2] = ((qd.hi & 0xffffu) >= a.thr) ? s[4 * g + 2] : 0.f;
          s[4 * g + 3] = (qd.hi >= (a.thr << 16)) ? s[4 * g + 3] : 0.f;
        } else {
          const bool kp[4] = {(qd.lo & 0xffffu) >= a.thr, qd.lo >= (a.thr << 16), (qd.hi & 0xffffu) >= a.thr, qd.hi >= (a.thr << 16)};
          typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            s[4 * g + j] = kp[j] ? s[4 * g + j] : 0.f;
            s[4 * g + j + 1] = kp[j + 1] ? s[4 * g + j + 1] : 0.f;
            const u64x2_t m = {__ballot(kp[j]), __ballot(kp[j + 1])};   // two registers' masks = four words, one store
            // (no "memory" clobber: nothing in this kernel reads these words, and a clobber would pin the V fragment reads
            // of the P V product behind the stores)
            asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(m), "s"(tb), "n"(8 * (4 * g + j)));
          }
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = acc_frag(s, s2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vt, k0 + 16 * s2, 32 * dt, lane), pf, o[dt], 0, 0, 0);
    }
  }
  ATTF_STAMP(4);
  l = xor32_sum(l);
  const float inv = (DROP ? a.inv_keep : 1.0f) / l;
  const int q = q0 + r;
  // A row's 8-byte pieces sit in lanes r (columns 8g .. 8g+3) and r + 32 (8g+4 .. 8g+7): one half-exchange per dword
  // (v_permlane32_swap) turns the pieces of two groups g, g+1 into 16 contiguous bytes per lane -- lanes < 32 get all of
  // group g, lanes >= 32 all of group g + 1 -- so the row leaves in 4 dwordx4 stores per lane instead of 8 dwordx2 (the
  // store tail is instruction-bound, not byte-bound)
  {
    bf16_t* dst = a.out + ((size_t)b * S + (q < S ? q : 0)) * a.H + hh * 64 + (h ? 8 : 0);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        uint32_t ax = pack2bf(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv), ay = pack2bf(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        uint32_t bx = pack2bf(o[dt][4 * g + 4] * inv, o[dt][4 * g + 5] * inv), by = pack2bf(o[dt][4 * g + 6] * inv, o[dt][4 * g + 7] * inv);
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ax), "+v"(bx));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ay), "+v"(by));
        if (q < S) *(uint4*)(dst + 32 * dt + 8 * g) = make_uint4(ax, ay, bx, by);
      }
#ifndef ATTF_STAMPS
    if (q < S && h == 0) a.lse[((size_t)b * a.nh + hh) * S + q] = (m + __log2f(l)) * LN2;
#endif
  }
  ATTF_STAMP(5);
#ifdef ATTF_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTF_STAMP(6);
#endif
  if (DROP && KEEP) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");   // the scalar stores sit in the scalar cache until written back
}

// ---------------------------------------------------------------------------------------------
// backward 1: dQ (and delta = rowsum(dO * O)).  One wave = 32 queries, sweeps 64-key blocks.
// ---------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char kt[2][64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char vt[2][64 * 128];
  __shared__ __attribute__((aligned(16))) float mb[512];
  __shared__ int padflag[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, hh = blockIdx.y;
  const int S = a.S, H3 = 3 * a.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int r = lane & 31, h = lane >> 5;
  const int spad = (S + 63) & ~63;
  const bool active = q0 < S;
  const float c2 = 0.125f * LOG2E;

  load_mask_bias(a, b, mb, padflag, spad, tid);

  const bf16_t* base = a.qkv + (size_t)b * S * H3 + hh * 64;
  int qr = q0 + r; if (qr > S - 1) qr = S - 1;
  const bool qvalid = (q0 + r) < S;
  bf16x8 bq[4], bdo[4];
  float dl = 0.f;
  {
    const bf16_t* dop = a.dctx + ((size_t)b * S + qr) * a.H + hh * 64;
    const bf16_t* op = a.ctx + ((size_t)b * S + qr) * a.H + hh * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bq[ks] = ld_frag_global(base + (size_t)qr * H3 + 16 * ks + 8 * h);
      bdo[ks] = ld_frag_global(dop + 16 * ks + 8 * h);
      const bf16x8 of = ld_frag_global(op + 16 * ks + 8 * h);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)bdo[ks][j] * (float)of[j];
    }
  }
  dl = xor32_sum(dl);
  const float lse2 = qvalid ? a.lse_in[((size_t)b * a.nh + hh) * S + qr] * LOG2E : 1.0e30f;
  if (active && qvalid && h == 0) a.delta[((size_t)b * a.nh + hh) * S + qr] = dl;

  f32x16 dq[2];
  dq[0] = dq[1] = (f32x16)(0.f);
  const uint32_t S4 = (uint32_t)((S + 3) & ~3);
  const uint32_t rkey = czc_rng_key(a.seed, a.site);
  const uint32_t erow = DROP ? ((uint32_t)((b * a.nh + hh) * S + (q0 + r)) * S4) : 0u;

  TileRegs<64> kr, vr;
  tile_fetch<64>(kr, base + a.H, H3, S, tid);
  tile_fetch<64>(vr, base + 2 * a.H, H3, S, tid);
  tile_commit<64>(kr, kt[0], tid);
  tile_commit<64>(vr, vt[0], tid);
  int buf = 0;
  for (int kb0 = 0; kb0 < S; kb0 += 64, buf ^= 1) {
    __syncthreads();
    const bool more = kb0 + 64 < S;
    if (more) {
      tile_fetch<64>(kr, base + a.H + (size_t)(kb0 + 64) * H3, H3, S - kb0 - 64, tid);
      tile_fetch<64>(vr, base + 2 * a.H + (size_t)(kb0 + 64) * H3, H3, S - kb0 - 64, tid);
    }
    if (active) {
#pragma unroll 1
    for (int tt = 0; tt < 2; ++tt) {
      f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(kt[buf], 32 * tt, ks, lane), bq[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(vt[buf], 32 * tt, ks, lane), bdo[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bias = *(const f32x4*)(mb + kb0 + 32 * tt + 8 * g + 4 * h);
        float z[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP) {
          keep4(rkey, erow + (uint32_t)(kb0 + 32 * tt + 8 * g + 4 * h), a.thr, a.inv_keep, z);
        }
        const f32x4 bl = bias - lse2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[4 * g + j], c2, bl[j]));
          s[4 * g + j] = p * (z[j] * dp[4 * g + j] - dl);   // dS^T (before the 1/8 scale)
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 df = acc_frag(s, s2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt[buf], 32 * tt + 16 * s2, 32 * dt, lane), df, dq[dt], 0, 0, 0);
      }
    }
    }
    if (more) {
      tile_commit<64>(kr, kt[buf ^ 1], tid);
      tile_commit<64>(vr, vt[buf ^ 1], tid);
    }
  }
  if (a.dbias) {
    // bias gradient of the Q projection: column sums over this wave's valid queries -> LDS -> one partial row
    // per workgroup (no global atomics: 2048 adds per address would serialise at the memory side)
    float* red = (float*)kt[0];
    __syncthreads();
    if (tid < 64) red[tid] = 0.f;
    __syncthreads();
    if (active) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = half_sum(qvalid ? dq[dt][i] * 0.125f : 0.f);
          if (r == 0) atomicAdd(&red[32 * dt + (i & 3) + 8 * (i >> 2) + 4 * h], v);
        }
    }
    __syncthreads();
    if (tid < 64) a.dbias[(size_t)(blockIdx.z * gridDim.x + blockIdx.x) * H3 + hh * 64 + tid] = red[tid];
  }
  if (!active) return;
  if (!qvalid) return;
  bf16_t* dst = a.dqkv + ((size_t)b * S + q0 + r) * H3 + hh * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = pack2bf(dq[dt][4 * g] * 0.125f, dq[dt][4 * g + 1] * 0.125f);
      w.y = pack2bf(dq[dt][4 * g + 2] * 0.125f, dq[dt][4 * g + 3] * 0.125f);
      *(uint2*)(dst + 32 * dt + 8 * g + 4 * h) = w;
    }
}

// ---------------------------------------------------------------------------------------------
// backward 2: dK, dV.  One wave = 32 keys, sweeps 32-query blocks (Q, dO, lse, delta in LDS).
// ---------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char qt[2][64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char dot[2][64 * 128];
  __shared__ __attribute__((aligned(16))) float lse_s[2][64];
  __shared__ __attribute__((aligned(16))) float del_s[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, hh = blockIdx.y;
  const int S = a.S, H3 = 3 * a.H;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const int r = lane & 31, h = lane >> 5;
  const bool active = k0 < S;
  const float c2 = 0.125f * LOG2E;

  const bf16_t* base = a.qkv + (size_t)b * S * H3 + hh * 64;
  const bf16_t* dobase = a.dctx + (size_t)b * S * a.H + hh * 64;
  const float* lsep = a.lse_in + ((size_t)b * a.nh + hh) * S;
  const float* delp = a.delta + ((size_t)b * a.nh + hh) * S;
  int kr = k0 + r; if (kr > S - 1) kr = S - 1;
  const bool kvalid = (k0 + r) < S;
  const float kbias = (kvalid && a.mask[(size_t)b * S + kr]) ? 0.f : NEG_MASK;
  bf16x8 bk[4], bv[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bk[ks] = ld_frag_global(base + a.H + (size_t)kr * H3 + 16 * ks + 8 * h);
    bv[ks] = ld_frag_global(base + 2 * a.H + (size_t)kr * H3 + 16 * ks + 8 * h);
  }
  f32x16 dk[2], dv[2];
  dk[0] = dk[1] = dv[0] = dv[1] = (f32x16)(0.f);
  const uint32_t S4 = (uint32_t)((S + 3) & ~3);
  const uint32_t rkey = czc_rng_key(a.seed, a.site);
  const uint32_t ebase = (uint32_t)((b * a.nh + hh) * S) * S4 + (uint32_t)(k0 + r);

  TileRegs<64> qr, dr;
  float lreg = 0.f, dreg = 0.f;
  auto fetch = [&](int qb0) {
    tile_fetch<64>(qr, base + (size_t)qb0 * H3, H3, S - qb0, tid);
    tile_fetch<64>(dr, dobase + (size_t)qb0 * a.H, a.H, S - qb0, tid);
    if (tid < 64) {
      const int q = qb0 + tid;
      lreg = (q < S) ? lsep[q] * LOG2E : 1.0e30f;
      dreg = (q < S) ? delp[q] : 0.f;
    }
  };
  auto commit = [&](int bufi) {
    tile_commit<64>(qr, qt[bufi], tid);
    tile_commit<64>(dr, dot[bufi], tid);
    if (tid < 64) { lse_s[bufi][tid] = lreg; del_s[bufi][tid] = dreg; }
  };
  fetch(0);
  commit(0);
  int buf = 0;
  for (int qb0 = 0; qb0 < S; qb0 += 64, buf ^= 1) {
    __syncthreads();
    const bool more = qb0 + 64 < S;
    if (more) fetch(qb0 + 64);
    if (active) {
#pragma unroll 1
      for (int hf = 0; hf < 2; ++hf) {
        if (qb0 + 32 * hf < S) {   // wave-uniform
          f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(qt[buf], 32 * hf, ks, lane), bk[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(dot[buf], 32 * hf, ks, lane), bv[ks], dp, 0, 0, 0);
          }
          // rows (registers) are queries 32hf + 8g + 4h + j, the lane is the key
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *(const f32x4*)(&lse_s[buf][32 * hf + 8 * g + 4 * h]);
            const f32x4 d4 = *(const f32x4*)(&del_s[buf][32 * hf + 8 * g + 4 * h]);
            const f32x4 kl = kbias - l4;
            // The draws of (query, keys 4c .. 4c+3) are one RNG quad and those four keys are four adjacent lanes
            // here: lane (r & 3) = j hashes the quad of query j and the group shares the words with quad-perm DPP
            // moves -- one hash per lane and 4 queries instead of four.  A lane's own 16-bit draw (r & 3) of a
            // word pair is compared in place: high half as is, low half shifted up (draw >= thr <=> t >= thr << 16).
            czc_quad qd = {0u, 0u};
            if (DROP)
              qd = czc_rng_quad(rkey, (ebase + (uint32_t)(qb0 + 32 * hf + 8 * g + 4 * h + (r & 3)) * S4) >> 2);
            const uint32_t qlo[4] = {quad_bcast<0>(qd.lo), quad_bcast<1>(qd.lo), quad_bcast<2>(qd.lo), quad_bcast<3>(qd.lo)};
            const uint32_t qhi[4] = {quad_bcast<0>(qd.hi), quad_bcast<1>(qd.hi), quad_bcast<2>(qd.hi), quad_bcast<3>(qd.hi)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p = __builtin_amdgcn_exp2f(fmaf(s[4 * g + j], c2, kl[j]));
              float z = 1.f;
              if (DROP) {
                const uint32_t t = ((r & 2) ? qhi[j] : qlo[j]) << ((r & 1) ? 0 : 16);
                z = t >= (a.thr << 16) ? a.inv_keep : 0.f;
              }
              const float dpz = z * dp[4 * g + j];
              dp[4 * g + j] = p * (dpz - d4[j]);  // dS
              s[4 * g + j] = p * z;               // P~
            }
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = acc_frag(s, s2), df = acc_frag(dp, s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot[buf], 32 * hf + 16 * s2, 32 * dt, lane), pf, dv[dt], 0, 0, 0);
              dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt[buf], 32 * hf + 16 * s2, 32 * dt, lane), df, dk[dt], 0, 0, 0);
            }
          }
        }
      }
    }
    if (more) commit(buf ^ 1);
  }
  if (a.dbias) {
    float* red = (float*)qt[0];   // [2][64]: K then V
    __syncthreads();
    if (tid < 128) red[tid] = 0.f;
    __syncthreads();
    if (active) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int d = 32 * dt + (i & 3) + 8 * (i >> 2) + 4 * h;
          const float vk = half_sum(kvalid ? dk[dt][i] * 0.125f : 0.f);
          const float vv = half_sum(kvalid ? dv[dt][i] : 0.f);
          if (r == 0) { atomicAdd(&red[d], vk); atomicAdd(&red[64 + d], vv); }
        }
    }
    __syncthreads();
    if (tid < 128) {
      float* row = a.dbias + (size_t)(blockIdx.z * gridDim.x + blockIdx.x) * H3;
      row[(tid < 64 ? a.H : 2 * a.H) + hh * 64 + (tid & 63)] = red[tid];
    }
  }
  if (!active) return;
  if (!kvalid) return;
  bf16_t* dstk = a.dqkv + ((size_t)b * S + k0 + r) * H3 + a.H + hh * 64;
  bf16_t* dstv = dstk + a.H;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = pack2bf(dk[dt][4 * g] * 0.125f, dk[dt][4 * g + 1] * 0.125f);
      w.y = pack2bf(dk[dt][4 * g + 2] * 0.125f, dk[dt][4 * g + 3] * 0.125f);
      *(uint2*)(dstk + 32 * dt + 8 * g + 4 * h) = w;
      w.x = pack2bf(dv[dt][4 * g], dv[dt][4 * g + 1]);
      w.y = pack2bf(dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
      *(uint2*)(dstv + 32 * dt + 8 * g + 4 * h) = w;
    }
}

// ---------------------------------------------------------------------------------------------
// backward, sequences of at most 256 tokens: dQ, dK and dV from ONE pass over the scores.
// The two kernels above each rebuild S, P~ and dS (exp, dropout draws, conversions: they are VALU-bound, the MFMAs
// wait); with the whole (batch, head) in one 512-thread workgroup that work is done once.  Wave w owns keys
// 32w .. 32w+31 (K, V fragments and the dK, dV accumulators in registers, as in the dK/dV kernel); all Q and dO rows
// sit in LDS; dQ is accumulated in LDS as fp32 [256][64].  In step s wave w works on query block (w + s) mod 8, so at
// any time every block has exactly one writer: no atomics, and each dQ element receives its eight contributions in a
// fixed order (bitwise reproducible, like the two-kernel form).  The dQ product contracts over KEYS while the dS tile
// sits in accumulators as [query rows][key = lane]: it goes through a wave-private 2 KB LDS tile (bf16 [key][query],
// one row per lane) and comes back as the transposed A operand; its B operand, this wave's K tile indexed
// [key][d = lane], is loaded once from global memory in the matching k-order.
// ---------------------------------------------------------------------------------------------
// A operand [m = query = lane&31][k = key] from a wave's bf16 [32 keys][32 queries] tile (64-B rows), k-order of frag_tr
__device__ __forceinline__ bf16x8 frag_tr64(const unsigned char* tile, int rbase, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = g >> 1;
  const int c = 2 * (g & 1) + (p >> 1);
  const int row1 = rbase + 4 * h + q, row2 = row1 + 8;
  typedef s16x4 __attribute__((address_space(3)))* trp_t;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(tile + row1 * 64 + (c << 4) + 8 * (p & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(tile + row2 * 64 + (c << 4) + 8 * (p & 1)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(const AttnArgs a) {
  // one buffer: Q image 32 KB | dO image 32 KB | dQ accumulators, TRANSPOSED: 64 d-rows x 260 floats (65 KB; a lane's
  // four consecutive queries of one d are one 16-byte access, rows 4 banks apart) | per-wave dS tiles 16 KB | lse, delta
  // 2 KB.  The epilogue reuses it from the start for the dK and dV tiles (2 x 256 rows x 68 floats = 136 KB) and the
  // 8 KB after them for partial column sums.
  constexpr int PITCH = 68, QP = 260;
  __shared__ __attribute__((aligned(16))) unsigned char smem[153600];
  unsigned char* const qt = smem;
  unsigned char* const dot = smem + 32768;
  float* const dqs = (float*)(smem + 65536);
  float* const lse_s = (float*)(smem + 151552);
  float* const del_s = lse_s + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, hh = blockIdx.y;
  const int S = a.S, H3 = 3 * a.H;
  const int k0 = wave * 32;
  const int r = lane & 31, h = lane >> 5;
  const bool active = k0 < S;   // wave-uniform
  const float c2 = 0.125f * LOG2E;

  const bf16_t* base = a.qkv + (size_t)b * S * H3 + hh * 64;
  const bf16_t* dobase = a.dctx + (size_t)b * S * a.H + hh * 64;
  const bf16_t* obase = a.ctx + (size_t)b * S * a.H + hh * 64;
  const float* lsep = a.lse_in + ((size_t)b * a.nh + hh) * S;

  // Every global load of the prologue is issued before the first use of any of them (one memory round trip): Q / dO
  // rows for the LDS images, the O chunks that pair with this thread's dO chunks (delta), lse, the key's mask byte and
  // this wave's K and V fragments.
  uint4 qv[4], dv4[4], ov[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 512 * i, row = id >> 3, c = id & 7;
    qv[i] = dv4[i] = ov[i] = (uint4){0u, 0u, 0u, 0u};
    if (row < S) {
      qv[i] = *(const uint4*)(base + (size_t)row * H3 + c * 8);
      dv4[i] = *(const uint4*)(dobase + (size_t)row * a.H + c * 8);
      ov[i] = *(const uint4*)(obase + (size_t)row * a.H + c * 8);
    }
  }
  const float lse_v = (tid < S) ? lsep[tid] * LOG2E : 1.0e30f;
  int kr = k0 + r; if (kr > S - 1) kr = S - 1;
  const bool kvalid = (k0 + r) < S;
  const uint8_t kmask = a.mask[(size_t)b * S + kr];
  bf16x8 bk[4], bv[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bk[ks] = ld_frag_global(base + a.H + (size_t)kr * H3 + 16 * ks + 8 * h);
    bv[ks] = ld_frag_global(base + 2 * a.H + (size_t)kr * H3 + 16 * ks + 8 * h);
  }
  __builtin_amdgcn_sched_barrier(0);
  // Q and dO images (rows past S are zero); delta[row] = sum_d dO O: the eight threads that hold a row's chunks add up
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 512 * i, row = id >> 3, c = id & 7;
    *(uint4*)(qt + row * 128 + ((c ^ swz64(row)) << 4)) = qv[i];
    *(uint4*)(dot + row * 128 + ((c ^ swz64(row)) << 4)) = dv4[i];
    const bf16x8 x = __builtin_bit_cast(bf16x8, dv4[i]), y = __builtin_bit_cast(bf16x8, ov[i]);
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)x[j] * (float)y[j];
    dl += dpp_mov<CZC_DPP_XOR1>(dl);
    dl += dpp_mov<CZC_DPP_XOR2>(dl);
    dl += dpp_mov<CZC_DPP_HALF_MIRROR>(dl);
    if (c == 0) {
      del_s[row] = dl;
      if (row < S && a.delta) a.delta[((size_t)b * a.nh + hh) * S + row] = dl;
    }
  }
  if (tid < 256) lse_s[tid] = lse_v;
  // This wave's K tile once more as the B operand [k = key][n = d = 32dt + r] of the dQ product, keys in the
  // accumulator-fed k-order: the fragments just loaded go through the wave's own slice of the (still unused) dQ
  // accumulator area as a [32 keys][64 d] image and come back through transposing reads.  Then the slice is zeroed.
  bf16x8 bkd[2][2];
  {
    unsigned char* kimg = (unsigned char*)(dqs + 8 * QP * wave);     // this wave's slice: d-rows 8w .. 8w+7
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      *(bf16x8*)(kimg + r * 128 + (((2 * ks + h) ^ swz64(r)) << 4)) = bk[ks];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bkd[dt][ks] = frag_tr(kimg, 16 * ks, 32 * dt, lane);
    __builtin_amdgcn_sched_barrier(0);
    for (int i = lane; i < 8 * QP / 4; i += 64) *(f32x4*)(dqs + 8 * QP * wave + 4 * i) = (f32x4)(0.f);
  }
  const float kbias = (kvalid && kmask) ? 0.f : NEG_MASK;
  f32x16 dk[2], dv[2];
  dk[0] = dk[1] = dv[0] = dv[1] = (f32x16)(0.f);
  const uint32_t S4 = (uint32_t)((S + 3) & ~3);
  const uint32_t rkey = czc_rng_key(a.seed, a.site);
  const uint32_t ebase = (uint32_t)((b * a.nh + hh) * S) * S4 + (uint32_t)(k0 + r);
  unsigned char* my = smem + 135168 + wave * 2048;
  __syncthreads();

#pragma unroll 1
  for (int step = 0; step < 8; ++step) {
    const int qb = 32 * ((wave + step) & 7);
    if (active && qb < S) {   // wave-uniform
      f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(qt, qb, ks, lane), bk[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(dot, qb, ks, lane), bv[ks], dp, 0, 0, 0);
      }
      // rows (registers) are queries qb + 8g + 4h + j, the lane is the key
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *(const f32x4*)(&lse_s[qb + 8 * g + 4 * h]);
        const f32x4 d4 = *(const f32x4*)(&del_s[qb + 8 * g + 4 * h]);
        const f32x4 kl = kbias - l4;
        czc_quad qd = {0u, 0u};   // one RNG quad per lane, shared inside each group of four lanes (see the dK/dV kernel)
        if (DROP) qd = czc_rng_quad(rkey, (ebase + (uint32_t)(qb + 8 * g + 4 * h + (r & 3)) * S4) >> 2);
        const uint32_t qlo[4] = {quad_bcast<0>(qd.lo), quad_bcast<1>(qd.lo), quad_bcast<2>(qd.lo), quad_bcast<3>(qd.lo)};
        const uint32_t qhi[4] = {quad_bcast<0>(qd.hi), quad_bcast<1>(qd.hi), quad_bcast<2>(qd.hi), quad_bcast<3>(qd.hi)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[4 * g + j], c2, kl[j]));
          float z = 1.f;
          if (DROP) {
            const uint32_t t = ((r & 2) ? qhi[j] : qlo[j]) << ((r & 1) ? 0 : 16);
            z = t >= (a.thr << 16) ? a.inv_keep : 0.f;
          }
          const float dpz = z * dp[4 * g + j];
          dp[4 * g + j] = p * (dpz - d4[j]);  // dS
          s[4 * g + j] = p * z;               // P~
        }
      }
      f32x16 dq[2];
      dq[0] = dq[1] = (f32x16)(0.f);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_frag(s, s2), df = acc_frag(dp, s2);
        // dS^T tile row `r` (this lane's key): queries 16 s2 + 4h .. +3 and 16 s2 + 8 + 4h .. +3
        const uint4 dw = __builtin_bit_cast(uint4, df);
        *(uint2*)(my + r * 64 + 2 * (16 * s2 + 4 * h)) = make_uint2(dw.x, dw.y);
        *(uint2*)(my + r * 64 + 2 * (16 * s2 + 8 + 4 * h)) = make_uint2(dw.z, dw.w);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot, qb + 16 * s2, 32 * dt, lane), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, qb + 16 * s2, 32 * dt, lane), df, dk[dt], 0, 0, 0);
        }
      }
      // dQ[query][d] += sum_key dS[query][key] K[key][d]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 af = frag_tr64(my, 16 * ks, lane);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bkd[dt][ks], dq[dt], 0, 0, 0);
      }
      // dq[dt][4g + j] = dQ[qb + 8g + 4h + j][32dt + r]: this wave is the only writer of block qb in this step; plain
      // 16-byte read-add-write per (dt, g) (ds_add_f32 measured 9x slower for the whole step)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4* at = (f32x4*)(dqs + (32 * dt + r) * QP + qb + 8 * g + 4 * h);
          f32x4 v = *at;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += dq[dt][4 * g + j];
          *at = v;
        }
    }
    __syncthreads();
  }
  // dQ: LDS fp32 (transposed) -> bf16 rows (x 1/8), two threads per query; its bias gradient = sums of the same numbers
  float* const red = lse_s;                       // [3][64] Q, K, V column sums of dqkv; lse / delta are done with
  float* const psum = (float*)(smem + 139264);    // [16][128] partial column sums of the dK | dV tiles
  {
    const int q = tid & 255, half = tid >> 8;     // consecutive lanes = consecutive queries: conflict-free column reads
    if (q < S) {
      bf16_t* dstq = a.dqkv + ((size_t)b * S + q) * H3 + hh * 64 + 32 * half;
      const float* src = dqs + 32 * half * QP + q;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint4 w;
        w.x = pack2bf(src[(8 * c + 0) * QP] * 0.125f, src[(8 * c + 1) * QP] * 0.125f);
        w.y = pack2bf(src[(8 * c + 2) * QP] * 0.125f, src[(8 * c + 3) * QP] * 0.125f);
        w.z = pack2bf(src[(8 * c + 4) * QP] * 0.125f, src[(8 * c + 5) * QP] * 0.125f);
        w.w = pack2bf(src[(8 * c + 6) * QP] * 0.125f, src[(8 * c + 7) * QP] * 0.125f);
        *(uint4*)(dstq + 8 * c) = w;
      }
    }
  }
  if (a.dbias) {
    // Q part of the bias gradient: row sums of the transposed accumulators, eight threads per d (queries past S are 0)
    const int d = tid >> 3, part = tid & 7;
    f32x4 sq = (f32x4)(0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) sq += *(const f32x4*)(dqs + d * QP + 32 * part + 4 * i);
    float t = (sq[0] + sq[1]) + (sq[2] + sq[3]);
    t += dpp_mov<CZC_DPP_XOR1>(t);
    t += dpp_mov<CZC_DPP_XOR2>(t);
    t += dpp_mov<CZC_DPP_HALF_MIRROR>(t);
    if (part == 0) red[d] = t * 0.125f;
  }
  __syncthreads();
  // dK (x 1/8) and dV tiles -> LDS as fp32 [256 keys][68] each (a lane owns a row; inactive waves and keys past S write
  // zeros), then every thread sums columns for the bias gradient and packs 64-byte pieces of rows for dqkv
  float* const dks = (float*)smem;
  float* const dvs = dks + 256 * PITCH;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 32 * dt + 8 * g + 4 * h;
      f32x4 x, y;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = (active && kvalid) ? dk[dt][4 * g + j] * 0.125f : 0.f;
        y[j] = (active && kvalid) ? dv[dt][4 * g + j] : 0.f;
      }
      *(f32x4*)(dks + (k0 + r) * PITCH + d0) = x;
      *(f32x4*)(dvs + (k0 + r) * PITCH + d0) = y;
    }
  __syncthreads();
  if (a.dbias) {
    const int quad = tid & 31, grp = tid >> 5;      // 32 column quads (dK | dV) x 16 groups of 16 rows
    const float* src = (quad < 16 ? dks : dvs) + 4 * (quad & 15);
    f32x4 sk = (f32x4)(0.f);
#pragma unroll
    for (int i = 0; i < 16; ++i) sk += *(const f32x4*)(src + (16 * grp + i) * PITCH);
    *(f32x4*)(psum + grp * 128 + 4 * quad) = sk;
  }
  {
    const int row = tid >> 1, half = tid & 1;
    if (row < S) {
      bf16_t* dstk = a.dqkv + ((size_t)b * S + row) * H3 + a.H + hh * 64 + 32 * half;
      bf16_t* dstv = dstk + a.H;
      const float* sk = dks + row * PITCH + 32 * half;
      const float* sv = dvs + row * PITCH + 32 * half;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 x = *(const f32x4*)(sk + 8 * c), y = *(const f32x4*)(sk + 8 * c + 4);
        const f32x4 u = *(const f32x4*)(sv + 8 * c), v = *(const f32x4*)(sv + 8 * c + 4);
        uint4 w;
        w.x = pack2bf(x[0], x[1]); w.y = pack2bf(x[2], x[3]); w.z = pack2bf(y[0], y[1]); w.w = pack2bf(y[2], y[3]);
        *(uint4*)(dstk + 8 * c) = w;
        w.x = pack2bf(u[0], u[1]); w.y = pack2bf(u[2], u[3]); w.z = pack2bf(v[0], v[1]); w.w = pack2bf(v[2], v[3]);
        *(uint4*)(dstv + 8 * c) = w;
      }
    }
  }
  if (a.dbias) {
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += psum[g * 128 + tid];
      red[64 + tid] = t;
    }
    __syncthreads();
    if (tid < 192) {
      float* row = a.dbias + (size_t)blockIdx.z * H3;
      row[(tid >> 6) * a.H + hh * 64 + (tid & 63)] = red[tid];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward, sequences of at most 256 tokens, STREAMING form: one persistent workgroup per CU walks over its (batch, head)
// items; nothing of an item is loaded or stored outside the step loop except its K / V fragments and its dK / dV tiles.
//   * Wave w owns keys 32w .. 32w+31 (K, V fragments, dK and dV accumulators in registers) as in the fused kernel above,
//     but in a step ALL waves work on the SAME 32-query block: Q, dO and O blocks (4 KB each) and the block's lse travel
//     by LDS DMA through a 4-stage ring, requested three steps ahead -- across item seams too -- so the step loop never
//     waits for HBM, and delta = rowsum(dO O) of block n+1 is taken from the ring during step n.
//   * dQ of a block is complete after one step: each wave leaves its dS tile ([32 keys][32 queries] bf16) in a
//     double-buffered LDS ring; in the next step wave (qh, dc) multiplies dS^T of ALL keys with its 16 columns of K (held
//     in registers for the whole item as eight 16x16x32 operands): dQ^T[16 d][16 queries], no accumulation across steps,
//     no fp32 dQ image in LDS (65 KB in the fused kernel), each element summed in one MFMA chain in a fixed order.  The
//     tile goes through a 4 KB LDS image and is stored as 8-byte row pieces one step later.
//   * The next item's K and V rows arrive by LDS DMA during this item's steps (one 1 KB piece per wave and step) in a
//     64 KB staging area; at the seam the waves pick their fragments up from there, then the same area transposes the
//     dK / dV tiles for row-contiguous stores and the bias-gradient column sums.
// Counted waits only: per step a wave issues, in this order, its K / V pieces, its ring
// pieces (two for waves 0-5, the lse piece for wave 6, none for wave 7; past the schedule's end they re-fetch the last
// block), then stores; at the end of step n its ring pieces of block n+2 (issued in step n-1) have at least the ring pieces
// of step n behind them: s_waitcnt vmcnt(2) / vmcnt(1).
// LDS: ring 4 x (Q 4096 | dO 4096 | O 4096 | raw lse 256) | lse (log2 units) 4 x 32 | delta 4 x 32 | dS 2 x 16 KB |
//      dQ tiles 2 x 4 KB (the seam's column-sum partials reuse them) | K staging 32 KB | V staging 32 KB | sums 2 KB.
// ---------------------------------------------------------------------------------------------
typedef void __attribute__((address_space(3)))* att_lptr_t;
#ifdef ATT_BWD_NT   // A/B build: streaming policy for the backward's LDS-DMA operands
#define ATT_POL " nt"
#else
#define ATT_POL ""
#endif
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is ours to set here
__device__ __forceinline__ void att_dma16(const void* src, unsigned dst) {   // 16 B per lane -> dst + 16 * lane
  const unsigned l = __builtin_amdgcn_readfirstlane(dst);   // LDS byte address (wave-uniform)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ATT_POL ::"v"(src), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ void att_dma4(const void* src, unsigned dst) {    // 4 B per lane -> dst + 4 * lane
  const unsigned l = __builtin_amdgcn_readfirstlane(dst);   // LDS byte address (wave-uniform)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ATT_POL ::"v"(src), "s"(l) : "memory", "m0");
}
// the same from a wave-uniform 64-bit base (scalar registers) plus a 32-bit per-lane byte offset: the lane part of a piece's
// address never changes, so the step loop spends scalar adds on addresses instead of 64-bit vector arithmetic
__device__ __forceinline__ void att_dma16_s(unsigned off, const void* base, unsigned dst) {
  const unsigned l = __builtin_amdgcn_readfirstlane(dst);   // LDS byte address (wave-uniform)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ATT_POL ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}
// Two pieces to dst and dst + 1024 behind ONE write of M0: the instruction's immediate offset moves both the source and the
// LDS address, so the second lane offset comes in 1024 bytes short (off1m = off1 - 1024).
__device__ __forceinline__ void att_dma16_s2(unsigned off0, unsigned off1m, const void* base, unsigned dst) {
  const unsigned l = __builtin_amdgcn_readfirstlane(dst);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ATT_POL "\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" ATT_POL
               ::"v"(off0), "v"(off1m), "s"(base), "s"(l) : "memory", "m0");
}
__device__ __forceinline__ void att_dma4_s(unsigned off, const void* base, unsigned dst) {
  const unsigned l = __builtin_amdgcn_readfirstlane(dst);   // LDS byte address (wave-uniform)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ATT_POL ::"v"(off), "s"(base), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop
// The mask-byte load of the streaming kernel is inline asm as well: the compiler's wait-count pass knows nothing of the DMA
// pieces, and for a load of its own pending across the step loop it put an s_waitcnt vmcnt(0) in front of the next write
// to that register -- which waits for every piece just requested.  Stores stay the compiler's: it needs no wait for them,
// and it knows their hazards (a 16-byte store wants TWO wait states before a VALU write of its data registers on gfx950:
// with one, an asm store here lost the first dwords of lanes 12-15 of every 16 to the next address computation).
#define ATT_BAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// 8-byte chunk swizzle of a dS tile row (key r of the tile): the 16x16x32 operand reads below and the row writes of the
// producer are both bank-conflict free with it
__device__ __forceinline__ int dsx(int r) { return (((r >> 3) & 1) << 2) | (((r >> 4) & 1) << 1) | ((r >> 2) & 1); }
// 16x16x32 operand, lane (i = lane & 15, g = lane >> 4): element j = tile[8g + j][col0 + i]; tile = bf16 [32][32], 64-B rows
__device__ __forceinline__ bf16x8 frag_ds16(const unsigned char* tile, int col0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int row1 = 8 * g + q, row2 = row1 + 4, ch = (col0 >> 2) + p;
  typedef s16x4 __attribute__((address_space(3)))* trp_t;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(tile + row1 * 64 + ((ch ^ dsx(row1)) << 3)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(tile + row2 * 64 + ((ch ^ dsx(row2)) << 3)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// the same from a [rows][64] image (128-B rows, swz64 chunks): element j = img[row0 + 8g + j][col0 + i]
__device__ __forceinline__ bf16x8 frag_img16(const unsigned char* img, int row0, int col0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int row1 = row0 + 8 * g + q, row2 = row1 + 4, c = (col0 >> 3) + (p >> 1);
  typedef s16x4 __attribute__((address_space(3)))* trp_t;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(img + row1 * 128 + ((c ^ swz64(row1)) << 4) + 8 * (p & 1)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(img + row2 * 128 + ((c ^ swz64(row2)) << 4) + 8 * (p & 1)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// FULL: the sequence is exactly 256 tokens (the headline shape): every wave is active, no row index is clamped, the block
// count is a constant -- the wave-uniform branches that guard those cases split the step into many basic blocks, and the
// compiler schedules inside a block only; without them the dQ chain and the delta chain sit in ONE block with the softmax
// arithmetic and their LDS / MFMA latencies run under it.
// PIPE (with FULL): the S / dP products of block n+1 are issued at the top of step n, before the softmax arithmetic of block n
// (whose scores were made a step earlier): their fragment reads and MFMA latency run under it.
// BITS (with DROP): the keep bits the forward left (AttnArgs::bits) travel through the ring, 1 KB per block (wave 7's
// piece: the words of the block's 32 queries for all 256 keys); a lane reads the word of its key and tests bits instead of
// hashing: the draws were a third of the vector instructions.  All 160 KB of LDS.
template <bool DROP, bool FULL, bool PIPE, bool BITS>
__global__ __launch_bounds__(512, 2) void attn_bwd_stream_kernel(const AttnArgs a) {
  constexpr int ST = BITS ? 13568 : 12544;   // Q 4096 | dO 4096 | O 4096 | raw lse 256 | keep words 1024
  constexpr int O_LSE = 4 * ST, O_DEL = O_LSE + 512, O_DS = O_DEL + 512, O_OUT = O_DS + 32768, O_K = O_OUT + 8192, O_V = O_K + 32768,
                O_RED = O_V + 32768;
  __shared__ __attribute__((aligned(16))) unsigned char smem[O_RED + 2048];
  const unsigned smem_lds = (unsigned)(__UINTPTR_TYPE__)(att_lptr_t)smem;   // DMA destinations are LDS byte addresses
  // Lane-constant addresses must not be hoisted out of the loops (they were: 44 registers spilled to scratch, and scratch
  // reloads are vector-memory operations that wait behind the DMA): every loop body starts from a laundered lane id.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int lane0 = threadIdx.x & 63;
#define ATT_LANE(ln) int ln = lane0; asm volatile("" : "+v"(ln))
  const int S = FULL ? 256 : a.S, H = a.H, H3 = 3 * a.H, nh = a.nh;
  const int nblk = FULL ? 8 : (S + 31) >> 5;
  const int G = gridDim.x;
  const int my_items = (a.B * nh - (int)blockIdx.x + G - 1) / G;
  const int total = my_items * nblk;
  const float c2 = 0.125f * LOG2E;
  const int k0 = wave * 32;
  const bool active = FULL || k0 < S;            // wave-uniform
  const int qh = wave & 1, dc = wave >> 1;       // this wave's dQ^T tile: d 16dc .. +15, queries 16qh .. +15
  const uint32_t S4 = (uint32_t)((S + 3) & ~3);
  const uint32_t rkey = czc_rng_key(a.seed, a.site);
  float* const lse_s = (float*)(smem + O_LSE);
  float* const del_s = (float*)(smem + O_DEL);
  float* const red = (float*)(smem + O_RED);      // (unused slots) | [2][64] the two query halves' dQ column sums
  float* const redq = red + 192;
  float* const psum = (float*)(smem + O_OUT);     // [16][128] partial column sums of the dK | dV tiles (seam only)

  // ---- prefetch cursor: block pf_n of this workgroup's schedule = step pf_st of item pf_item
  int pf_n = 0, pf_st = 0, pf_item = blockIdx.x;
  const bf16_t *pf_q, *pf_do, *pf_o;
  const float* pf_lse;
  const bf16_t* pf_mine;
  const uint32_t* pf_bits = nullptr;
  auto pf_set = [&]() __attribute__((always_inline)) {
    const int b = pf_item / nh, hh = pf_item - b * nh;
    if (BITS) pf_bits = a.bits + (size_t)pf_item * 8 * 256;
    pf_q = a.qkv + (size_t)b * S * H3 + hh * 64;
    pf_do = a.dctx + (size_t)b * S * H + hh * 64;
    pf_o = a.ctx + (size_t)b * S * H + hh * 64;
    pf_lse = a.lse_in + ((size_t)b * nh + hh) * S;
    // the tensor this wave's two ring pieces come from (waves 0-5), picked from values, not from the captured variables
    const bf16_t* const mq = a.qkv + (size_t)b * S * H3 + hh * 64;
    const bf16_t* const md = a.dctx + (size_t)b * S * H + hh * 64;
    const bf16_t* const mo = a.ctx + (size_t)b * S * H + hh * 64;
    pf_mine = wave < 2 ? mq : (wave < 4 ? md : mo);
  };
  pf_set();
  // Lane parts of the DMA source addresses, computed once on purpose (six registers): byte offsets of this lane's 16 bytes
  // inside a 32-row block (ring pieces; slot 2 wave + j) and inside a 64-row group (K / V pieces), for rows that need no
  // clamping to S - 1.  Blocks and groups that reach past S take the general path below.
  unsigned ring_off[2], kv_off[2];
  {
    const int lane = lane0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int slot = 2 * wave + j, tsr = slot >> 2, p = slot & 3;
      const int row = 8 * p + (lane >> 3), c = (lane & 7) ^ swz64(row);
      ring_off[j] = slot < 12 ? (unsigned)((row * (tsr == 0 ? H3 : H) + c * 8) * 2) : (slot == 14 ? (unsigned)(lane * 16) : (unsigned)((lane & 31) * 4));
      const int rk = 16 * (wave & 3) + 8 * j + (lane >> 3), ck = (lane & 7) ^ swz64(rk);
      kv_off[j] = (unsigned)((rk * H3 + ck * 8) * 2);
    }
  }
  unsigned kr_off;   // this lane's key, clamped to S - 1: offset of its mask byte in a batch row
  { int kr = k0 + (lane0 & 31); if (kr > S - 1) kr = S - 1; kr_off = (unsigned)kr; }
  auto issue_ring = [&](int lane) __attribute__((always_inline)) {   // two pieces per wave of block pf_n -> stage pf_n & 3; past the schedule's end: the last block again
    const unsigned stg = smem_lds + (pf_n & 3) * ST;
    const int qb = 32 * pf_st;
    if (FULL || qb + 32 <= S) {   // no row of the block is clamped: scalar bases + the precomputed lane offsets
      if (wave < 6) {   // a wave's two pieces are neighbours in the stage: one write of M0, the second piece by the immediate offset
        const int tsr = wave >> 1, p = 2 * (wave & 1);
        const bf16_t* base = pf_mine + (size_t)qb * (wave < 2 ? H3 : H);
        att_dma16_s2(ring_off[0], ring_off[1] - 1024u, base, stg + tsr * 4096 + p * 1024);
      } else if (wave == 6) {
        att_dma4_s(ring_off[0], pf_lse + qb, stg + 12288);
      } else if (BITS) {   // wave 7: the block's keep words, 64 lanes x 16 bytes
        att_dma16_s(ring_off[0], pf_bits + pf_st * 256, stg + 12544);
      }
    } else
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int slot = 2 * wave + j;             // 0-3 Q, 4-7 dO, 8-11 O (waves 0-5), 12 lse (wave 6); wave 7 has no ring piece
      if (BITS && slot == 14) { att_dma16_s(ring_off[j], pf_bits + pf_st * 256, stg + 12544); continue; }   // keep words: nothing to clamp
      if (slot > 12) continue;
      if (slot < 12) {
        const int tsr = slot >> 2, p = slot & 3;
        const int row = 8 * p + (lane >> 3), c = (lane & 7) ^ swz64(row);
        int gr = qb + row; if (gr > S - 1) gr = S - 1;
        const bf16_t* src = tsr == 0 ? pf_q + (size_t)gr * H3 + c * 8 : (tsr == 1 ? pf_do : pf_o) + (size_t)gr * H + c * 8;
        att_dma16(src, stg + tsr * 4096 + p * 1024);
      } else {
        int gr = qb + (lane & 31); if (gr > S - 1) gr = S - 1;
        att_dma4(pf_lse + gr, stg + 12288);
      }
    }
    ++pf_n;
    if (pf_n < total && ++pf_st == nblk) { pf_st = 0; pf_item += G; pf_set(); }
  };
  auto issue_kv = [&](const bf16_t* qbase, int st, int lane) __attribute__((always_inline)) {   // rows 32 st + 8 (wave & 3) .. +7 of K (waves 0-3) or V (waves 4-7) of the item at qbase
    const int tsr = wave >> 2;
    const int row = 32 * st + 8 * (wave & 3) + (lane >> 3), c = (lane & 7) ^ swz64(row);
    int gr = row; if (gr > S - 1) gr = S - 1;
    att_dma16(qbase + (size_t)gr * H3 + (tsr + 1) * H + c * 8, smem_lds + (tsr ? O_V : O_K) + (32 * st + 8 * (wave & 3)) * 128);
  };
  auto issue_kv64 = [&](const bf16_t* qbase, int grp, int lane) __attribute__((always_inline)) {   // rows 64 grp .. +63 of K (waves 0-3) and V (waves 4-7): two pieces per wave
    const int tsr = wave >> 2;
    if (FULL || 64 * grp + 64 <= S) {
      const bf16_t* base = qbase + (size_t)(64 * grp) * H3 + (tsr + 1) * H;
      att_dma16_s2(kv_off[0], kv_off[1] - 1024u, base, smem_lds + (tsr ? O_V : O_K) + (64 * grp + 16 * (wave & 3)) * 128);
    } else
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r0 = 64 * grp + 16 * (wave & 3) + 8 * j;
      const int row = r0 + (lane >> 3), c = (lane & 7) ^ swz64(row);
      int gr = row; if (gr > S - 1) gr = S - 1;
      att_dma16(qbase + (size_t)gr * H3 + (tsr + 1) * H + c * 8, smem_lds + (tsr ? O_V : O_K) + r0 * 128);
    }
  };
  // delta and lse (log2 units; rows past S: +1e30, their probabilities vanish) of block m, which is step st_m of its item
  auto delta_of = [&](int m, int st_m, int lane) __attribute__((always_inline)) {
    const unsigned char* sg = smem + (m & 3) * ST;
    const int row = 4 * wave + (lane >> 4), q8 = lane & 15;
    const int off = row * 128 + (((q8 >> 1) ^ swz64(row)) << 4) + 8 * (q8 & 1);
    const uint2 x = *(const uint2*)(sg + 4096 + off), y = *(const uint2*)(sg + 8192 + off);
    float dl = bflo(x.x) * bflo(y.x) + bfhi(x.x) * bfhi(y.x) + bflo(x.y) * bflo(y.y) + bfhi(x.y) * bfhi(y.y);
    dl = row_sum16(dl);
    if (FULL) {   // all sixteen lanes of a row hold the same two values: they all write them (no divergent branch)
      del_s[32 * (m & 3) + row] = dl;
      lse_s[32 * (m & 3) + row] = ((const float*)(sg + 12288))[row] * LOG2E;
      return;
    }
    if (q8 == 0) del_s[32 * (m & 3) + row] = dl;
    if (q8 == 1) {
      const float raw = ((const float*)(sg + 12288))[row];
      lse_s[32 * (m & 3) + row] = (32 * st_m + row < S) ? raw * LOG2E : 1.0e30f;
    }
  };

#ifdef ATT_STAMPS   // diagnostic build only (tools/stamp_attn.py): s_memtime of wave 0 / lane 0 at fixed points of the second item
  unsigned long long* const stamp_buf = (unsigned long long*)a.delta + 128 * blockIdx.x;
  int stamp_on = 0;
#define ATT_STAMP(k) do { if (stamp_on && wave == ATT_STAMPS && lane0 == 0) { stamp_buf[k] = __builtin_amdgcn_s_memtime(); if ((k) == 0 || (k) == 65) stamp_buf[66 + ((k) != 0)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define ATT_STAMP(k) do { } while (0)
#endif
  // ---- per-item state
  int item = blockIdx.x, b = item / nh, hh = item - b * nh;
  bf16x8 bk[4], bv[4], kT[8];
  float kbias = NEG_MASK;
  auto take_fragments = [&](uint8_t kmask, int lane) __attribute__((always_inline)) {   // this wave's K / V rows and its 16 columns of K (all keys) from the staging area
    const unsigned char* kimg = smem + O_K;
    const unsigned char* vimg = smem + O_V;
    if (active) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { bk[ks] = frag_row(kimg, k0, ks, lane); bv[ks] = frag_row(vimg, k0, ks, lane); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (FULL || i < nblk) kT[i] = frag_img16(kimg, 32 * i, 16 * dc, lane);
    kbias = ((k0 + (lane & 31)) < S && kmask) ? 0.f : NEG_MASK;
  };
  auto sdp_of = [&](int m, f32x16& so, f32x16& dpo, int lane) __attribute__((always_inline)) {   // S = Q K^T and dP = dO V^T of block m (stage m & 3), this wave's 32 keys
    const unsigned char* sg = smem + (m & 3) * ST;
    so = dpo = (f32x16)(0.f);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      so = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(sg, 0, ks, lane), bk[ks], so, 0, 0, 0);
      dpo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(sg + 4096, 0, ks, lane), bv[ks], dpo, 0, 0, 0);
    }
  };
  f32x16 s_cur = (f32x16)(0.f), dp_cur = (f32x16)(0.f);   // PIPE: the scores of the block the next step works on
  f32x16 dk[2], dv[2];
  f32x4 dqsum = (f32x4)(0.f);
  dk[0] = dk[1] = dv[0] = dv[1] = (f32x16)(0.f);

  // dQ^T tile of block m from the dS ring -> bf16 image m & 1 ([32 queries][64 d], 8-byte chunks XOR (query & 15))
  auto dq_block = [&](int m, int lane, bool live = true) __attribute__((always_inline)) {   // live = false (FULL, first step of an item): nothing is there yet, the tile is zeros
    const unsigned char* dsb = smem + O_DS + (m & 1) * 16384;
    f32x4 acc = (f32x4)(0.f), acc1 = (f32x4)(0.f);   // two chains (even / odd key tiles): half the dependent-MFMA latency
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      if (FULL || i < nblk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[i], frag_ds16(dsb + i * 2048, 16 * qh, lane), acc, 0, 0, 0);
      if (FULL || i + 1 < nblk) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[i + 1], frag_ds16(dsb + (i + 1) * 2048, 16 * qh, lane), acc1, 0, 0, 0);
    }
    acc = (acc + acc1) * 0.125f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = live ? acc[j] : 0.f;
    dqsum += acc;
    const int q = 16 * qh + (lane & 15);
    *(uint2*)(smem + O_OUT + (m & 1) * 4096 + q * 128 + (((4 * dc + (lane >> 4)) ^ (q & 15)) << 3)) =
        make_uint2(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]));
  };
  auto dq_store = [&](int m, int qb, int lane) __attribute__((always_inline)) {   // image m & 1 -> dqkv rows qb .. qb+31 of the current item, 8 bytes per thread
    const int row = 4 * wave + (lane >> 4), piece = lane & 15;
    if (qb + row < S) {
      const uint2 v = *(const uint2*)(smem + O_OUT + (m & 1) * 4096 + row * 128 + ((piece ^ (row & 15)) << 3));
      *(uint2*)(a.dqkv + ((size_t)b * S + qb + row) * H3 + hh * 64 + piece * 4) = v;
    }
  };

  // ---- prologue: the first item's K / V rows, blocks 0 .. 2, mask byte
  {
    ATT_LANE(lane);
    for (int st = 0; st < nblk; ++st) issue_kv(pf_q, st, lane);
    issue_ring(lane); issue_ring(lane); issue_ring(lane);
    int kr = k0 + (lane & 31); if (kr > S - 1) kr = S - 1;
    const uint8_t kmask = a.mask[(size_t)b * S + kr];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_BAR();
    delta_of(0, 0, lane);
    take_fragments(kmask, lane);
    if (PIPE) sdp_of(0, s_cur, dp_cur, lane);
    ATT_BAR();
  }

  // With more than one step per item the dK / dV rows of an item leave during the NEXT item's steps (64 rows of each per
  // step, read from the images the seam left in the staging area), and the K / V rows of the item after that arrive in the
  // 64 rows freed one step earlier: a seam that stored them itself spent 9.7k of an item's 69k cycles issuing stores.
  const bool defer = nblk > 1;
  int pend_b = 0, pend_hh = 0;
  bool pending = false;
  int n = 0;
#pragma unroll 1
  for (int il = 0; il < my_items; ++il) {
    const bool has_next = il + 1 < my_items;
    const int nitem = has_next ? item + G : item;
    const int nb = nitem / nh, nhh = nitem - nb * nh;
    const bf16_t* nq = a.qkv + (size_t)nb * S * H3 + nhh * 64;
    const uint32_t ebase0 = (uint32_t)((b * nh + hh) * S) * S4 + (uint32_t)k0;
#ifdef ATT_STAMPS
    stamp_on = il == 1;
#endif
    ATT_STAMP(0);
#pragma unroll 1
    for (int st = 0; st < nblk; ++st, ++n) {
      ATT_LANE(lane);
      const int r = lane & 31, h = lane >> 5;
      const unsigned char* stg = smem + (n & 3) * ST;
      const unsigned char* qt = stg;
      const unsigned char* dot = stg + 4096;
      const float* lse_b = lse_s + 32 * (n & 3);
      const float* del_b = del_s + 32 * (n & 3);
      const int qb = 32 * st;
      // The step's stores go BEFORE its requests: vmcnt retires in order, so the counted wait at the step's end (all but the
      // newest pieces) finds this step's requests youngest, whatever the step stores
      if (st > 1) dq_store(n - 2, qb - 64, lane);
      if (pending && 2 * st < nblk) {   // rows 64 st .. +63 of the previous item's dK and dV
        const int tid = 64 * wave + lane, row = 64 * st + (tid >> 3), c = tid & 7;
        if (row < S) {
          const int off = row * 128 + ((c ^ swz64(row)) << 4);
          bf16_t* dstk = a.dqkv + ((size_t)pend_b * S + row) * H3 + H + pend_hh * 64 + 8 * c;
          *(uint4*)dstk = *(const uint4*)(smem + O_K + off);
          *(uint4*)(dstk + H) = *(const uint4*)(smem + O_V + off);
        }
      }
      if (!defer) issue_kv(nq, st, lane);
      else if (st >= 1 && 2 * (st - 1) < nblk) issue_kv64(nq, st - 1, lane);
      issue_ring(lane);
      ATT_STAMP(1 + 6 * st);
      if (!FULL && st > 0) dq_block(n - 1, lane);
      ATT_STAMP(2 + 6 * st);
      if (active) {
        f32x16 s, dp;
        if (PIPE) {   // this block's scores were made a step ago; the next block's are requested now (at an item's last step
                      // they are made with the wrong K / V and never used: the seam makes the next item's first ones)
          s = s_cur; dp = dp_cur;
          sdp_of(n + 1, s_cur, dp_cur, lane);
        } else {
          sdp_of(n, s, dp, lane);
        }
        if (FULL) {   // the two short chains in the same basic block as the softmax arithmetic
          dq_block(n - 1, lane, st > 0);
          delta_of(n + 1, st + 1 == nblk ? 0 : st + 1, lane);
        }
        // rows (registers) are queries qb + 8g + 4h + j, the lane is the key
        const uint32_t ebase = ebase0 + (uint32_t)r;
        // BITS: the word of this lane's key, bit i = query qb + i is kept; shifted so that bit 8g + j is query 8g + 4h + j
        uint32_t kw = 0;
        if (BITS) kw = ((const uint32_t*)(stg + 12544))[k0 + 8 * (r >> 3) + 2 * (r & 3) + ((r >> 2) & 1)] >> (4 * h);
        const uint32_t ikb = __builtin_bit_cast(uint32_t, a.inv_keep);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = *(const f32x4*)(&lse_b[8 * g + 4 * h]);
          const f32x4 d4 = *(const f32x4*)(&del_b[8 * g + 4 * h]);
          const f32x4 kl = kbias - l4;
          czc_quad qd = {0u, 0u};   // one RNG quad per lane, shared inside each group of four lanes (see the dK/dV kernel)
          if (DROP && !BITS) qd = czc_rng_quad(rkey, (ebase + (uint32_t)(qb + 8 * g + 4 * h + (r & 3)) * S4) >> 2);
          const uint32_t qlo[4] = {quad_bcast<0>(qd.lo), quad_bcast<1>(qd.lo), quad_bcast<2>(qd.lo), quad_bcast<3>(qd.lo)};
          const uint32_t qhi[4] = {quad_bcast<0>(qd.hi), quad_bcast<1>(qd.hi), quad_bcast<2>(qd.hi), quad_bcast<3>(qd.hi)};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[4 * g + j], c2, kl[j]));
            float z = 1.f;
            if (DROP && BITS) {   // sign-extended one-bit field (0 / all ones) AND the bits of 1 / (1 - p)
              int keep;   // as asm: the compiler rewrites the builtin into and + compare + select
              asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(kw), "n"(8 * g + j));
              z = __builtin_bit_cast(float, (uint32_t)keep & ikb);
            } else if (DROP) {
              const uint32_t t = ((r & 2) ? qhi[j] : qlo[j]) << ((r & 1) ? 0 : 16);
              z = t >= (a.thr << 16) ? a.inv_keep : 0.f;
            }
            const float pz = p * z;                                  // P~
            dp[4 * g + j] = fmaf(pz, dp[4 * g + j], -(p * d4[j]));   // dS = P (z dP - delta)
            s[4 * g + j] = pz;
          }
        }
        unsigned char* my = smem + O_DS + (n & 1) * 16384 + wave * 2048 + r * 64;
        const int xs = dsx(r);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_frag(s, s2), df = acc_frag(dp, s2);
          // dS^T tile row r (this lane's key): queries 16 s2 + 4h .. +3 and 16 s2 + 8 + 4h .. +3
          const uint4 dw = __builtin_bit_cast(uint4, df);
          *(uint2*)(my + (((4 * s2 + h) ^ xs) << 3)) = make_uint2(dw.x, dw.y);
          *(uint2*)(my + (((4 * s2 + 2 + h) ^ xs) << 3)) = make_uint2(dw.z, dw.w);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot, 16 * s2, 32 * dt, lane), pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, 16 * s2, 32 * dt, lane), df, dk[dt], 0, 0, 0);
          }
        }
      }
      ATT_STAMP(3 + 6 * st);
      if (!FULL) delta_of(n + 1, st + 1 == nblk ? 0 : st + 1, lane);
      ATT_STAMP(4 + 6 * st);
      // behind a wave's pieces of block n+2 (requested in step n-1) are at least its pieces of block n+3: two (waves 0-5),
      // one (wave 6: lse); wave 7 has no ring piece
      if (wave < 6) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (wave == 6 || BITS) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // BITS: wave 7 has the keep-word piece
      ATT_STAMP(5 + 6 * st);
      ATT_BAR();
      ATT_STAMP(6 + 6 * st);
    }
    // ---- seam: last dQ block, the next item's fragments, this item's dK / dV
    ATT_LANE(lane);
    const int r = lane & 31, h = lane >> 5, tid = 64 * wave + lane;
    // The next item's mask bytes of this wave's 32 keys: one scalar load of 32 bytes, waited for on the spot (a vector
    // load here would be the compiler's, and its wait-count pass, which knows nothing of the DMA pieces in flight, would
    // wait for all of them; an asm vector load hands the compiler a register it believes valid before the byte arrives).
    // Waves whose keys reach past S or whose bytes are not dword-aligned read their byte the plain way.
    uint32_t kmask_n = 0;
    {
      const uint8_t* mrow = a.mask + (size_t)nb * S;
      if (k0 + 32 <= S && (((__UINTPTR_TYPE__)(mrow + k0)) & 3) == 0) {
        typedef unsigned u32x8_t __attribute__((ext_vector_type(8)));
        u32x8_t mw;
        asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mw) : "s"(mrow + k0) : "memory");
        const uint32_t a0 = (r & 4) ? mw[1] : mw[0], a1 = (r & 4) ? mw[3] : mw[2], a2 = (r & 4) ? mw[5] : mw[4], a3 = (r & 4) ? mw[7] : mw[6];
        const uint32_t b0 = (r & 8) ? a1 : a0, b1 = (r & 8) ? a3 : a2;
        kmask_n = (((r & 16) ? b1 : b0) >> (8 * (r & 3))) & 0xffu;
      } else if (active) {
        kmask_n = mrow[kr_off];
        asm volatile("" : "+v"(kmask_n));   // used here: the compiler waits for its load now, nothing of it stays pending into the loops
      }
    }
    dq_block(n - 1, lane);
    if (nblk > 1) dq_store(n - 2, 32 * (nblk - 2), lane);
    {
      f32x4 t = dqsum;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = row_sum16(t[j]);
      if ((lane & 15) == 0) *(f32x4*)(redq + 64 * qh + 16 * dc + 4 * (lane >> 4)) = t;
      dqsum = (f32x4)(0.f);
    }
    ATT_STAMP(60);
    // the K / V pieces of the next item: only the last step's ring pieces are behind them for sure
    if (wave < 6) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_BAR();
    ATT_STAMP(61);
    dq_store(n - 1, 32 * (nblk - 1), lane);
    // dK (x 1/8) and dV tiles -> bf16 registers; then the next item's fragments may replace K / V
    const bool okvalid = active && (k0 + r) < S;
    uint2 pk[8], pv[8];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float x[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = okvalid ? dk[dt][4 * g + j] * 0.125f : 0.f;
          y[j] = okvalid ? dv[dt][4 * g + j] : 0.f;
        }
        pk[4 * dt + g] = make_uint2(pack2bf(x[0], x[1]), pack2bf(x[2], x[3]));
        pv[4 * dt + g] = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
      }
    dk[0] = dk[1] = dv[0] = dv[1] = (f32x16)(0.f);
    if (has_next) {
      take_fragments((uint8_t)(kmask_n & 0xffu), lane);
      if (PIPE) sdp_of(n, s_cur, dp_cur, lane);
    }
    ATT_BAR();
    ATT_STAMP(62);
    // bf16 [256 keys][64] images in the staging area (a lane owns a row; keys past S are zeros)
    {
      unsigned char* rk = smem + O_K + (k0 + r) * 128;
      unsigned char* rv = smem + O_V + (k0 + r) * 128;
      const int sw = swz64(k0 + r);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int off = ((c ^ sw) << 4) + 8 * h;
        *(uint2*)(rk + off) = pk[c];
        *(uint2*)(rv + off) = pv[c];
      }
    }
    ATT_BAR();
    ATT_STAMP(63);
    pending = defer && has_next;
    pend_b = b; pend_hh = hh;
    {
      const int row = tid >> 1, half = tid & 1;
      if (!pending && row < S) {
        bf16_t* dstk = a.dqkv + ((size_t)b * S + row) * H3 + H + hh * 64 + 32 * half;
        bf16_t* dstv = dstk + H;
        const int sw = swz64(row);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int off = row * 128 + (((4 * half + c) ^ sw) << 4);
          *(uint4*)(dstk + 8 * c) = *(const uint4*)(smem + O_K + off);
          *(uint4*)(dstv + 8 * c) = *(const uint4*)(smem + O_V + off);
        }
      }
      if (a.dbias) {
        const int quad = tid & 31, grp = tid >> 5;      // 32 column quads (dK | dV) x 16 groups of 16 rows
        const unsigned char* img = smem + (quad < 16 ? O_K : O_V);
        const int q4 = quad & 15;
        f32x4 sk = (f32x4)(0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int rw = 16 * grp + i;
          const uint2 w = *(const uint2*)(img + rw * 128 + (((q4 >> 1) ^ swz64(rw)) << 4) + 8 * (q4 & 1));
          sk += (f32x4){bflo(w.x), bfhi(w.x), bflo(w.y), bfhi(w.y)};
        }
        *(f32x4*)(psum + grp * 128 + 4 * quad) = sk;
      }
    }
    ATT_BAR();
    ATT_STAMP(64);
    if (a.dbias && tid < 192) {
      float t;
      if (tid < 64) t = redq[tid] + redq[64 + tid];
      else {
        t = 0.f;
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) t += psum[gq * 128 + tid - 64];
      }
      a.dbias[(size_t)b * H3 + (tid >> 6) * H + hh * 64 + (tid & 63)] = t;
    }
    item = nitem; b = nb; hh = nhh;
    ATT_STAMP(65);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // dummy ring pieces may still be on their way into this workgroup's LDS
#undef ATT_LANE
#undef ATT_STAMP
}

static int attn_check(const char* who, int B, int S, int nh, float p) {
  CZC_REQUIRE(B > 0 && S > 0 && S <= 512 && nh > 0, "%s: batch=%d seq=%d heads=%d (seq <= 512)", who, B, S, nh);
  CZC_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout_p=%f", who, p);
  CZC_REQUIRE((uint64_t)B * nh * S * S < (1ull << 32), "%s: batch*heads*seq*seq must be < 2^32 (dropout counter)", who);
  return CZC_OK;
}

}  // namespace czc

using namespace czc;

static int g_attn_keep = -1;          // CZC_ATTN_KEEP_WORDS=0: the keep words are neither written nor read, the backward hashes again (A/B runs)
static int g_attn_stream_full = -1;   // CZC_ATTN_STREAM_FULL=0: the general streaming kernel at 256 tokens too, 1: specialised without the S / dP pipeline, 2 (default): with it (A/B runs)
static int g_attn_stream_bwd = -1;  // CZC_ATTN_STREAM_BWD=0: the one-workgroup-per-item fused backward instead of the streaming one (A/B runs)
static int g_attn_fused_bwd = -1;   // CZC_ATTN_FUSED_BWD=0 keeps the two-kernel backward for short sequences too (A/B runs)

namespace czc {
// one partial row per batch row from the single-pass kernels, one per (batch row, 128-query block) from the kernel pair
int attn_bwd_partial_rows(int batch, int seq) {
  if (g_attn_fused_bwd < 0) { const char* e = getenv("CZC_ATTN_FUSED_BWD"); g_attn_fused_bwd = (e && e[0] == '0') ? 0 : 1; }
  return (seq <= 256 && g_attn_fused_bwd) ? batch : batch * ceil_div(seq, 128);
}
}  // namespace czc

extern "C" int64_t czc_attention_bias_scratch_floats(int batch, int seq, int heads) {
  return (int64_t)batch * ceil_div(seq, 128) * 3 * heads * 64;
}

extern "C" int64_t czc_attention_keep_words(int batch, int seq, int heads) {
  return seq <= 256 ? (int64_t)batch * heads * 8 * 256 : 0;
}

extern "C" int czc_attention_forward(const uint16_t* qkv, const uint8_t* mask, int batch, int seq, int heads,
                                     float dropout_p, uint32_t seed, uint32_t site, uint16_t* ctx, float* lse,
                                     czc_stream_t stream) {
  return czc_attention_forward_keep(qkv, mask, batch, seq, heads, dropout_p, seed, site, ctx, lse, nullptr, stream);
}

extern "C" int czc_attention_forward_keep(const uint16_t* qkv, const uint8_t* mask, int batch, int seq, int heads,
                                          float dropout_p, uint32_t seed, uint32_t site, uint16_t* ctx, float* lse,
                                          uint32_t* keep_words, czc_stream_t stream) {
  if (int rc = attn_check("czc_attention_forward", batch, seq, heads, dropout_p)) return rc;
  CZC_REQUIRE(!keep_words || seq <= 256, "czc_attention_forward_keep: keep words exist for seq <= 256 only (seq=%d)", seq);
  ProfScope prof((hipStream_t)stream, "attention_fwd", 4.0 * seq * (double)seq * 64.0 * heads * batch,
                 2.0 * batch * (double)seq * heads * 64.0 * 4.0);
  AttnArgs a{};
  if (g_attn_keep < 0) { const char* e = getenv("CZC_ATTN_KEEP_WORDS"); g_attn_keep = (e && e[0] == '0') ? 0 : 1; }
  a.qkv = qkv; a.mask = mask; a.out = ctx; a.lse = lse; a.bits = (dropout_p > 0.f && g_attn_keep) ? keep_words : nullptr;
  a.B = batch; a.S = seq; a.nh = heads; a.H = heads * 64;
  a.thr = dropout_p > 0.f ? czc_drop_thr(dropout_p) : 0u;
  a.inv_keep = dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f;
  a.seed = seed; a.site = site;
  if (seq <= 256) {   // all of K and V fit one workgroup's LDS
    dim3 grid(1, heads, batch);
    if (a.thr && a.bits) hipLaunchKernelGGL((attn_fwd_res_kernel<true, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else if (a.thr) hipLaunchKernelGGL((attn_fwd_res_kernel<true, false>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((attn_fwd_res_kernel<false, false>), grid, dim3(512), 0, (hipStream_t)stream, a);
  } else {
    dim3 grid(ceil_div(seq, 128), heads, batch);
    if (a.thr) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_attention_backward(const uint16_t* qkv, const uint8_t* mask, const uint16_t* ctx,
                                      const uint16_t* dctx, const float* lse, int batch, int seq, int heads,
                                      float dropout_p, uint32_t seed, uint32_t site, float* delta, uint16_t* dqkv,
                                      float* d_bias_qkv, float* partials, czc_stream_t stream) {
  return czc_attention_backward_keep(qkv, mask, ctx, dctx, lse, batch, seq, heads, dropout_p, seed, site, nullptr, delta, dqkv,
                                     d_bias_qkv, partials, stream);
}

extern "C" int czc_attention_backward_keep(const uint16_t* qkv, const uint8_t* mask, const uint16_t* ctx,
                                           const uint16_t* dctx, const float* lse, int batch, int seq, int heads,
                                           float dropout_p, uint32_t seed, uint32_t site, const uint32_t* keep_words,
                                           float* delta, uint16_t* dqkv, float* d_bias_qkv, float* partials,
                                           czc_stream_t stream) {
  if (int rc = attn_check("czc_attention_backward", batch, seq, heads, dropout_p)) return rc;
  ProfScope prof((hipStream_t)stream, "attention_bwd", 10.0 * seq * (double)seq * 64.0 * heads * batch,
                 2.0 * batch * (double)seq * heads * 64.0 * 8.0);
  AttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.ctx = ctx; a.dctx = dctx; a.lse_in = lse; a.delta = delta; a.dqkv = dqkv; a.dbias = d_bias_qkv ? partials : nullptr;
  if (g_attn_keep < 0) { const char* e = getenv("CZC_ATTN_KEEP_WORDS"); g_attn_keep = (e && e[0] == '0') ? 0 : 1; }
  a.bits = (dropout_p > 0.f && seq <= 256 && g_attn_keep) ? const_cast<uint32_t*>(keep_words) : nullptr;
  CZC_REQUIRE(!d_bias_qkv || partials, "czc_attention_backward: d_bias_qkv needs the partials scratch");
  a.B = batch; a.S = seq; a.nh = heads; a.H = heads * 64;
  a.thr = dropout_p > 0.f ? czc_drop_thr(dropout_p) : 0u;
  a.inv_keep = dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f;
  a.seed = seed; a.site = site;
  hipStream_t st = (hipStream_t)stream;
  if (g_attn_fused_bwd < 0) { const char* e = getenv("CZC_ATTN_FUSED_BWD"); g_attn_fused_bwd = (e && e[0] == '0') ? 0 : 1; }
  if (g_attn_stream_bwd < 0) { const char* e = getenv("CZC_ATTN_STREAM_BWD"); g_attn_stream_bwd = (e && e[0] == '0') ? 0 : 1; }
  if (seq <= 256 && g_attn_fused_bwd && g_attn_stream_bwd) {   // persistent workgroups, operands streamed through LDS
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      CZC_HIP(hipGetDevice(&dev));
      CZC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int items = batch * heads;
    dim3 gridp(items < cus ? items : cus);
    if (g_attn_stream_full < 0) { const char* e = getenv("CZC_ATTN_STREAM_FULL"); g_attn_stream_full = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }
    if (seq == 256 && g_attn_stream_full == 2) {
      if (a.thr && a.bits) hipLaunchKernelGGL((attn_bwd_stream_kernel<true, true, true, true>), gridp, dim3(512), 0, st, a);
      else if (a.thr) hipLaunchKernelGGL((attn_bwd_stream_kernel<true, true, true, false>), gridp, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((attn_bwd_stream_kernel<false, true, true, false>), gridp, dim3(512), 0, st, a);
    } else if (seq == 256 && g_attn_stream_full) {
      if (a.thr) hipLaunchKernelGGL((attn_bwd_stream_kernel<true, true, false, false>), gridp, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((attn_bwd_stream_kernel<false, true, false, false>), gridp, dim3(512), 0, st, a);
    } else {
      if (a.thr && a.bits) hipLaunchKernelGGL((attn_bwd_stream_kernel<true, false, false, true>), gridp, dim3(512), 0, st, a);
      else if (a.thr) hipLaunchKernelGGL((attn_bwd_stream_kernel<true, false, false, false>), gridp, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((attn_bwd_stream_kernel<false, false, false, false>), gridp, dim3(512), 0, st, a);
    }
    CZC_LAUNCH_CHECK();
    if (d_bias_qkv && !g_defer_partials) return czc_reduce_partials(partials, batch, 3 * a.H, 3 * a.H, d_bias_qkv, stream);
    return CZC_OK;
  }
  if (seq <= 256 && g_attn_fused_bwd) {   // one pass over the scores for dQ, dK and dV
    dim3 grid1(1, heads, batch);

    if (a.thr) hipLaunchKernelGGL(attn_bwd_fused_kernel<true>, grid1, dim3(512), 0, st, a);
    else hipLaunchKernelGGL(attn_bwd_fused_kernel<false>, grid1, dim3(512), 0, st, a);
    CZC_LAUNCH_CHECK();
    if (d_bias_qkv && !g_defer_partials) return czc_reduce_partials(partials, batch, 3 * a.H, 3 * a.H, d_bias_qkv, stream);
    return CZC_OK;
  }
  dim3 grid(ceil_div(seq, 128), heads, batch);
  if (a.thr) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel<true>, grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel<false>, grid, dim3(256), 0, st, a);
  }
  CZC_LAUNCH_CHECK();
  if (d_bias_qkv && !g_defer_partials) return czc_reduce_partials(partials, batch * (int)grid.x, 3 * a.H, 3 * a.H, d_bias_qkv, stream);
  return CZC_OK;
}
