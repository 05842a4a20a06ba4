// czc_common.h — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/czc.h"

namespace czc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;  // storage type for bf16 in global memory

// ---- error plumbing (host) -----------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define CZC_HIP(expr)                                                     \
  do {                                                                    \
    hipError_t _e = (expr);                                               \
    if (_e != hipSuccess) return czc::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define CZC_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      czc::set_error(__VA_ARGS__);        \
      return CZC_ERR_INVALID;             \
    }                                     \
  } while (0)

#define CZC_LAUNCH_CHECK() CZC_HIP(hipGetLastError())

// ---- per-operation timing (czc_prof_enable / czc_prof_read, include/czc.h) -------------------------
// A scope around an operation's launches records a hipEvent pair on the launch stream while profiling is on and costs
// one predictable branch while it is off.  `flops` / `bytes` are the ALGORITHMIC work of the operation (what
// DESIGN.md prices it at), so the reader can turn the measured time into TFLOP/s or GB/s.
extern bool g_prof_on;
int prof_begin(hipStream_t st, const char* name, double flops, double bytes);
void prof_end(hipStream_t st, int handle);
// One batched reduction of several kernels' per-workgroup partial rows (the encoder's backward: the two LayerNorms, the
// dGELU bias and the attention QKV bias of a layer leave four sets of partial rows; four separate reduce launches of 5 us
// each were latency, not work).  While g_defer_partials is set the row kernels and the attention backward skip their own
// reduction; the caller describes the jobs and calls multi_reduce.
struct ReduceJob {
  const float* src;   // [nrows][row_stride] partial rows
  int nrows, row_stride, ncols, seg;   // column c is added to dst[c / seg][c % seg]
  float* dst[3];      // null entries are skipped
};
extern int g_defer_partials;
int multi_reduce(const ReduceJob* jobs, int njobs, hipStream_t st);   // rowops.hip
int ln_bwd_partial_rows(int rows);                                     // rows of partials czc_residual_ln_backward writes
int attn_bwd_partial_rows(int batch, int seq);                         // rows of partials czc_attention_backward writes (after a call)

struct ProfScope {
  hipStream_t st;
  int h;
  ProfScope(hipStream_t s, const char* name, double flops = 0.0, double bytes = 0.0)
      : st(s), h(g_prof_on ? prof_begin(s, name, flops, bytes) : -1) {}
  ~ProfScope() { if (h >= 0) prof_end(st, h); }
};

// ---- bf16 <-> f32 --------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast keeps NaN a NaN and lowers to v_cvt_pk_bf16_f32 (round-to-nearest-even)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  // vector conversion = one v_cvt_pk_bf16_f32 (RNE, NaN stays NaN); two scalar casts + shift/or cost 3-4 more ops
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- wave (64-lane) reductions ----------------------------------------------------------
// Butterflies on the VALU: DPP lane permutes inside a row of 16 (xor 1, xor 2, then the two mirrors, which add the
// other quad / the other half-row because every lane of a quad already holds the quad's value) and the gfx950
// v_permlane16_swap / v_permlane32_swap across rows.  A __shfl_xor is a ds_bpermute_b32: an LDS round trip (tens of
// cycles, waited for with lgkmcnt) per step, which made row statistics and column sums latency-bound.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#define CZC_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define CZC_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define CZC_DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i within 8
#define CZC_DPP_MIRROR 0x140       // lane i <-> 15 - i within 16
// lanes l and l ^ 16 / l ^ 32: both operands of the swap start as v, so afterwards a and b are the two partner values.
// Inline asm: the __builtin_amdgcn_permlane{16,32}_swap of this toolchain hands back its first result twice.
// (s_nop 1: the VALU write -> permlane-swap read hazard the compiler would otherwise pad itself)
__device__ __forceinline__ void xor16_pair(float v, float& a, float& b) {
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void xor32_pair(float v, float& a, float& b) {
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float row_sum16(float v) {   // sum over each row of 16 lanes, in every lane of the row
  v += dpp_mov<CZC_DPP_XOR1>(v);
  v += dpp_mov<CZC_DPP_XOR2>(v);
  v += dpp_mov<CZC_DPP_HALF_MIRROR>(v);
  v += dpp_mov<CZC_DPP_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_mov<CZC_DPP_XOR1>(v));
  v = fmaxf(v, dpp_mov<CZC_DPP_XOR2>(v));
  v = fmaxf(v, dpp_mov<CZC_DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<CZC_DPP_MIRROR>(v));
  return v;
}
__device__ __forceinline__ float xor16_sum(float v) { float a, b; xor16_pair(v, a, b); return a + b; }
__device__ __forceinline__ float xor32_sum(float v) { float a, b; xor32_pair(v, a, b); return a + b; }
__device__ __forceinline__ float xor16_max(float v) { float a, b; xor16_pair(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_max(float v) { float a, b; xor32_pair(v, a, b); return fmaxf(a, b); }
// sum over the 32 lanes of each half-wave (lanes that share lane>>5), result in every lane
__device__ __forceinline__ float half_sum(float v) { return xor16_sum(row_sum16(v)); }
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row_sum16(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row_max16(v))); }

// ---- counter-based dropout RNG -------------------------------------------------------
// Elements 4i .. 4i+3 of a dropout site share ONE mixed word finished twice (64 bits): 16 bits per element,
// keep <=> draw >= thr, thr = round(p * 65536).  Nothing is stored: the backward regenerates the same bits.
// tests/dropout_ref.py restates this in numpy so that the oracle can be handed identical masks.
__host__ __device__ __forceinline__ uint32_t czc_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t czc_rng_key(uint32_t seed, uint32_t site) {
  return czc_hash32(seed + 0x9E3779B9u * (site + 1u));
}
struct czc_quad { uint32_t lo, hi; };   // draws: lo & 0xffff, lo >> 16, hi & 0xffff, hi >> 16
__host__ __device__ __forceinline__ czc_quad czc_rng_quad(uint32_t key, uint32_t quad_index) {
  uint32_t x = quad_index ^ key;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15;
  czc_quad q;                           // two finalisers of the same mixed word, different odd multipliers
  q.lo = x * 0x846ca68bu; q.lo ^= q.lo >> 16;
  q.hi = x * 0xc2b2ae35u; q.hi ^= q.hi >> 16;
  return q;
}
__host__ __device__ __forceinline__ uint32_t czc_draw16(czc_quad q, uint32_t sel) {
  const uint32_t w = (sel & 2u) ? q.hi : q.lo;
  return (sel & 1u) ? (w >> 16) : (w & 0xffffu);
}
// keep-flag of element e
__host__ __device__ __forceinline__ bool czc_keep(uint32_t seed, uint32_t site, uint32_t e, uint32_t thr) {
  return czc_draw16(czc_rng_quad(czc_rng_key(seed, site), e >> 2), e & 3u) >= thr;
}
__host__ __device__ __forceinline__ uint32_t czc_drop_thr(float p) {
  return (uint32_t)(p * 65536.0f + 0.5f);
}

// erf-GELU (HF "gelu") and its derivative with ONE v_exp_f32 and ONE v_rcp_f32 per element:
//   Phi(x) = 0.5 (1 + erf(x / sqrt 2)),  erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z >= 0
//   (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7: far below the bf16 rounding of the stored result; libm's
//   erff costs ~3x the instructions and made the FFN epilogues VALU-bound).  exp(-z^2) = exp(-x^2 / 2) is also
//   the Gaussian factor of gelu'(x) = Phi(x) + x phi(x), so the derivative costs no second exponential.
struct GeluParts { float cdf, pdf; };
__device__ __forceinline__ GeluParts gelu_parts(float x) {
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2/2)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float erfc_half = 0.5f * poly * t * e;          // 0.5 * erfc(z)
  GeluParts g;
  g.cdf = x >= 0.f ? 1.0f - erfc_half : erfc_half;
  g.pdf = 0.39894228040143267794f * e;
  return g;
}
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_parts(x).cdf; }
// the same for 8 values, written stage by stage (all exponentials, all reciprocals, all polynomials ...): a
// transcendental's result may not be used by the very next instructions, and element-at-a-time code made the compiler
// pad every exp / rcp with s_nops (37 per 8 elements in the GELU epilogues)
// The GEMM epilogues' form, eight values at a time.  MODE 0: out = gelu(x);  1: out = gelu'(x);  2: out = gelu(x) and
// out2 = gelu'(x).  Their results are rounded to bf16 (2^-9 relative), so Phi is an odd polynomial here instead of the
// exp + rcp + quartic above:  Phi(x) - 0.5 = t P(t^2),  t = clamp(x, -4, 4),  P of degree 7 (weighted minimax fit,
// |error of Phi| <= 5.3e-5 in fp32 arithmetic over all x -- beyond +-4 the true tail is 3.2e-5 --, gelu within 1.3e-4
// relative for x > -1 and 4.2e-4 absolute anywhere): 12 VALU operations per value and no transcendental, against ~22
// with two transcendentals (rocprofv3: the bias+GELU epilogue issued 2 000 vector instructions per wave and output tile,
// 49 M per FFN-up launch, with the matrix pipe idle behind them).  The derivative adds x phi(x) with one v_exp_f32.
template <int MODE>
__device__ __forceinline__ void gelu8(const float (&x)[8], float (&out)[8], float (&out2)[8]) {
  // Stage by stage over the eight values, the stages fenced for the scheduler: a dependent v_pk_fma_f32 right behind the
  // one that feeds it costs a wait state (hipcc serialised each pair's Horner chain and put an s_nop between every two
  // steps: 560 per output tile); with four independent pairs per stage the next dependent use is four instructions away.
  float t[8], s[8], p[8], e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    t[i] = __builtin_amdgcn_fmed3f(x[i], -4.0f, 4.0f);
    s[i] = t[i] * t[i];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (MODE != 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(fmaf(s[i], -0.72134752044448170368f, -1.32574806473616f));  // phi(t)
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = fmaf(-1.5807964419e-09f, s[i], 1.2171171661e-07f);
  __builtin_amdgcn_sched_barrier(0);
#define CZC_HORNER(C)                                                \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) p[i] = fmaf(p[i], s[i], C); \
  __builtin_amdgcn_sched_barrier(0);
  CZC_HORNER(-4.1008809444e-06f)
  CZC_HORNER(8.0667571337e-05f)
  CZC_HORNER(-1.0482055429e-03f)
  CZC_HORNER(9.6648779952e-03f)
  CZC_HORNER(-6.6175384782e-02f)
  CZC_HORNER(3.9884751216e-01f)
#undef CZC_HORNER
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = fmaf(t[i], p[i], 0.5f);           // Phi (one rounding of 2^-25 at most: far below bf16)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {                                         // 3 operations for both outputs (were 5)
    if (MODE == 0 || MODE == 2) out[i] = x[i] * v[i];
    if (MODE == 1) out[i] = fmaf(t[i], e[i], v[i]);
    if (MODE == 2) out2[i] = fmaf(t[i], e[i], v[i]);
  }
}
__device__ __forceinline__ float dgelu_f(float x) {
  const GeluParts g = gelu_parts(x);
  return g.cdf + x * g.pdf;
}

// 16 bytes of zeros in global memory: out-of-range 16-B chunks of GEMM / attention operands are
// redirected here (a per-lane source address costs nothing with global_load_lds).
static __device__ uint4 g_zero16[4];

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace czc
