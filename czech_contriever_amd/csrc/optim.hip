// optim.hip — whole-model elementwise kernels over the FLAT parameter arrays (gfx950, HBM-bound):
// momentum EMA of the key encoder, AdamW, gradient norm / clip, fp32 -> bf16 mirror.
// Each runs as ONE launch over all 108.9 M elements (the reference issues ~600 tiny launches for
// the EMA alone, src/moco.py:260-269), 16 B per lane per access, grid-stride, and refreshes the
// bf16 mirror of the matrix region in the same pass so no separate cast pass is needed.

#include "czc_common.h"

namespace czc {

__device__ __forceinline__ void st_bf4g(bf16_t* p, f32x4 v) {
  uint2 w;
  w.x = pack2bf(v[0], v[1]);
  w.y = pack2bf(v[2], v[3]);
  *(uint2*)p = w;
}

// pk = pk*m + pq*(1-m): two rounded products and one rounded add, exactly like
// `param_k.data * m + param_q.data * (1. - m)` in torch (no fused multiply-add).
template <int V>   // V & 1: streaming (nt) loads and stores of the fp32 arrays; V & 2: two vectors per thread and iteration
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pk, const float* __restrict__ pq, int64_t n4,
                                                  int64_t n, float m, float om, bf16_t* __restrict__ pk_bf,
                                                  int64_t n_bf) {
#pragma clang fp contract(off)  // the two products must round separately (no FMA), as torch computes them
  constexpr bool NT = (V & 1) != 0;
  constexpr int U = (V & 2) ? 2 : 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
    f32x4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        a[u] = NT ? __builtin_nontemporal_load((const f32x4*)pk + i) : ((const f32x4*)pk)[i];
        b[u] = NT ? __builtin_nontemporal_load((const f32x4*)pq + i) : ((const f32x4*)pq)[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float p1 = a[u][j] * m, p2 = b[u][j] * om; r[j] = p1 + p2; }
        if (NT) __builtin_nontemporal_store(r, (f32x4*)pk + i); else ((f32x4*)pk)[i] = r;
        if (i * 4 + 3 < n_bf) st_bf4g(pk_bf + i * 4, r);
      }
    }
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float p1 = pk[i] * m, p2 = pq[i] * om;
    const float r = p1 + p2;
    pk[i] = r;
    if (i < n_bf) pk_bf[i] = f2bf(r);
  }
}

struct AdamArgs {
  float lr, beta1, beta2, eps, wd_factor;  // wd_factor = 1 - lr*weight_decay
  float one_minus_b1, one_minus_b2, step_size, inv_bc2_sqrt, grad_scale;
};

// EMA: the key encoder's momentum update of the NEXT forward in the same pass (pk = pk*em + p_new*eom while p_new is in
// registers: the standalone EMA pass would read p and pk again and write pk -- 12 B per parameter saved).  The EMA's two
// products and its add are rounded separately, exactly as ema_kernel does, so the fused pass is bit-identical to AdamW
// followed by the EMA.
struct EmaArgs { float* pk; bf16_t* pk_bf; float em, eom; };

template <int V, bool EMA>   // V as ema_kernel
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n4, int64_t n,
                                                    AdamArgs a, const float* __restrict__ clip_coef,
                                                    bf16_t* __restrict__ p_bf, int64_t n_bf, EmaArgs e) {
  constexpr bool NT = (V & 1) != 0;
  constexpr int U = (V & 2) ? 2 : 1;
  const float gs = a.grad_scale * (clip_coef ? clip_coef[0] : 1.0f);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
    f32x4 pp[U], gg[U], mm[U], vv[U], kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        if (EMA) kk[u] = NT ? __builtin_nontemporal_load((const f32x4*)e.pk + i) : ((const f32x4*)e.pk)[i];
        pp[u] = NT ? __builtin_nontemporal_load((const f32x4*)p + i) : ((const f32x4*)p)[i];
        gg[u] = (NT ? __builtin_nontemporal_load((const f32x4*)g + i) : ((const f32x4*)g)[i]) * gs;
        mm[u] = NT ? __builtin_nontemporal_load((const f32x4*)m + i) : ((const f32x4*)m)[i];
        vv[u] = NT ? __builtin_nontemporal_load((const f32x4*)v + i) : ((const f32x4*)v)[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = pp[u][j] * a.wd_factor;
          const float m1 = mm[u][j] + (gg[u][j] - mm[u][j]) * a.one_minus_b1;  // lerp, as torch
          const float v1 = vv[u][j] * a.beta2 + gg[u][j] * gg[u][j] * a.one_minus_b2;
          const float denom = sqrtf(v1) * a.inv_bc2_sqrt + a.eps;
          x -= a.step_size * (m1 / denom);
          pp[u][j] = x; mm[u][j] = m1; vv[u][j] = v1;
        }
        if (NT) {
          __builtin_nontemporal_store(pp[u], (f32x4*)p + i); __builtin_nontemporal_store(mm[u], (f32x4*)m + i);
          __builtin_nontemporal_store(vv[u], (f32x4*)v + i);
        } else {
          ((f32x4*)p)[i] = pp[u]; ((f32x4*)m)[i] = mm[u]; ((f32x4*)v)[i] = vv[u];
        }
        if (i * 4 + 3 < n_bf) st_bf4g(p_bf + i * 4, pp[u]);
        if (EMA) {
#pragma clang fp contract(off)  // two rounded products and a rounded add, as torch computes param_k * m + param_q * (1 - m)
          f32x4 r;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float p1 = kk[u][j] * e.em, p2 = pp[u][j] * e.eom; r[j] = p1 + p2; }
          if (NT) __builtin_nontemporal_store(r, (f32x4*)e.pk + i); else ((f32x4*)e.pk)[i] = r;
          if (i * 4 + 3 < n_bf) st_bf4g(e.pk_bf + i * 4, r);
        }
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float gj = g[i] * gs;
    float x = p[i] * a.wd_factor;
    const float m1 = m[i] + (gj - m[i]) * a.one_minus_b1;
    const float v1 = v[i] * a.beta2 + gj * gj * a.one_minus_b2;
    x -= a.step_size * (m1 / (sqrtf(v1) * a.inv_bc2_sqrt + a.eps));
    p[i] = x; m[i] = m1; v[i] = v1;
    if (i < n_bf) p_bf[i] = f2bf(x);
    if (EMA) {
#pragma clang fp contract(off)
      const float p1 = e.pk[i] * e.em, p2 = x * e.eom;
      const float r = p1 + p2;
      e.pk[i] = r;
      if (i < n_bf) e.pk_bf[i] = f2bf(r);
    }
  }
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* partials) {
  __shared__ float red[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 a = ((const f32x4*)g)[i];
    s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float a = g[(n4 << 2) + threadIdx.x]; s += a * a; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int np, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}
__global__ void clip_coef_kernel(const float* sumsq, float pre_scale, float max_norm, float* coef) {
  const float c = max_norm / (sqrtf(sumsq[0]) * pre_scale + 1e-6f);
  coef[0] = c < 1.0f ? c : 1.0f;
}
__global__ __launch_bounds__(256) void clamp_kernel(float* g, int64_t n, float lo, float hi) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] = fminf(fmaxf(g[i], lo), hi);
}
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) st_bf4g(dst + i * 4, ((const f32x4*)src)[i]);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const int64_t i = (n4 << 2) + threadIdx.x; dst[i] = f2bf(src[i]); }
}

// weight_decay_from_init (reference src/moco.py:332-336): the penalty wd * sum((p - p0)^2) and its gradient 2 wd (p - p0),
// each ONE pass over the flat arrays (the reference walks 197 tensors with three elementwise ops each)
__global__ __launch_bounds__(256) void l2diff_partial_kernel(const float* __restrict__ p, const float* __restrict__ p0, int64_t n,
                                                             float* partials) {
  __shared__ float red[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 d = __builtin_nontemporal_load((const f32x4*)p + i) - __builtin_nontemporal_load((const f32x4*)p0 + i);
    s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const int64_t i = (n4 << 2) + threadIdx.x; const float d = p[i] - p0[i]; s += d * d; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void l2diff_final_kernel(const float* __restrict__ partials, int np, float wd, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = wd * (red[0] + red[1] + red[2] + red[3]);
}
// g (+)= coef * scale[0] * (p - p0); scale = the upstream gradient of the penalty (a device scalar: no host round trip)
__global__ __launch_bounds__(256) void l2diff_grad_kernel(const float* __restrict__ p, const float* __restrict__ p0, int64_t n,
                                                          float coef, const float* __restrict__ scale, float* __restrict__ g,
                                                          int accumulate) {
  const float c = coef * (scale ? scale[0] : 1.0f);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 d = (__builtin_nontemporal_load((const f32x4*)p + i) - __builtin_nontemporal_load((const f32x4*)p0 + i)) * c;
    ((f32x4*)g)[i] = accumulate ? ((const f32x4*)g)[i] + d : d;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float d = (p[i] - p0[i]) * c;
    g[i] = accumulate ? g[i] + d : d;
  }
}

static int flat_grid(int64_t n4) {
  int64_t b = (n4 + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;   // 8 blocks per CU, grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace czc

using namespace czc;

extern "C" int czc_ema_update(float* pk, const float* pq, int64_t n, float m, float one_minus_m, uint16_t* pk_bf16,
                              int64_t n_bf16, czc_stream_t stream) {
  CZC_REQUIRE(n > 0 && n_bf16 >= 0 && n_bf16 <= n && (n_bf16 % 4) == 0, "czc_ema_update: n=%lld n_bf16=%lld",
              (long long)n, (long long)n_bf16);
  CZC_REQUIRE(((uintptr_t)pk % 16) == 0 && ((uintptr_t)pq % 16) == 0, "czc_ema_update: arrays must be 16-byte aligned");
  ProfScope prof((hipStream_t)stream, "ema", 0.0, 12.0 * n + 2.0 * (pk_bf16 ? n_bf16 : 0));
// streaming loads / stores of the fp32 arrays (read and written once; 254-264 -> 251-256 us), one vector per thread and
  // iteration (two: 300 us)
  hipLaunchKernelGGL(ema_kernel<1>, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, pk, pq, n / 4, n, m,
                     one_minus_m, pk_bf16, pk_bf16 ? n_bf16 : 0);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

static int adamw_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, const float* clip_coef_dev, uint16_t* p_bf16,
                        int64_t n_bf16, float* pk, uint16_t* pk_bf16, float ema_m, float ema_om, czc_stream_t stream) {
  CZC_REQUIRE(n > 0 && step >= 1 && n_bf16 >= 0 && n_bf16 <= n && (n_bf16 % 4) == 0,
              "czc_adamw_step: n=%lld step=%d n_bf16=%lld", (long long)n, step, (long long)n_bf16);
  CZC_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
              "czc_adamw_step: arrays must be 16-byte aligned");
  CZC_REQUIRE(!pk || (((uintptr_t)pk % 16) == 0 && pk != p && (n_bf16 == 0 || (pk_bf16 && p_bf16))),
              "czc_adamw_ema_step: the key array must be 16-byte aligned, distinct from p, with its bf16 mirror");
  ProfScope prof((hipStream_t)stream, pk ? "adamw+ema" : "adamw", 0.0, (pk ? 36.0 : 28.0) * n + (pk ? 4.0 : 2.0) * (p_bf16 ? n_bf16 : 0));
  AdamArgs a;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.wd_factor = (float)(1.0 - (double)lr * (double)weight_decay);
  a.one_minus_b1 = (float)(1.0 - (double)beta1);
  a.one_minus_b2 = (float)(1.0 - (double)beta2);
  a.step_size = (float)((double)lr / bc1);
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a.grad_scale = grad_scale;
// streaming loads / stores and two vectors per thread and iteration (eight loads in flight): 590-628 -> 549-553 us
  // in the step (either alone: 595)
  EmaArgs e{pk, (bf16_t*)pk_bf16, ema_m, ema_om};
  if (pk)
    hipLaunchKernelGGL((adamw_kernel<3, true>), dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, n, a,
                       clip_coef_dev, (bf16_t*)p_bf16, p_bf16 ? n_bf16 : 0, e);
  else
    hipLaunchKernelGGL((adamw_kernel<3, false>), dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, n, a,
                       clip_coef_dev, (bf16_t*)p_bf16, p_bf16 ? n_bf16 : 0, e);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale,
                              const float* clip_coef_dev, uint16_t* p_bf16, int64_t n_bf16, czc_stream_t stream) {
  return adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, clip_coef_dev, p_bf16, n_bf16,
                      nullptr, nullptr, 0.f, 0.f, stream);
}

extern "C" int czc_adamw_ema_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, int step, float grad_scale,
                                  const float* clip_coef_dev, uint16_t* p_bf16, int64_t n_bf16, float* pk,
                                  uint16_t* pk_bf16, float ema_m, float ema_one_minus_m, czc_stream_t stream) {
  CZC_REQUIRE(pk, "czc_adamw_ema_step: null key array");
  return adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, clip_coef_dev, p_bf16, n_bf16, pk,
                      pk_bf16, ema_m, ema_one_minus_m, stream);
}

extern "C" int czc_sumsq(const float* g, int64_t n, float* partials, float* out, czc_stream_t stream) {
  CZC_REQUIRE(n > 0 && partials && out, "czc_sumsq: bad arguments");
  const int nb = flat_grid(n / 4);  // <= 2048 <= 4096 partials
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, n, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, out);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_l2_from_init_forward(const float* p, const float* p0, int64_t n, float weight_decay, float* partials,
                                        float* out, czc_stream_t stream) {
  CZC_REQUIRE(n > 0 && p && p0 && partials && out && ((uintptr_t)p % 16) == 0 && ((uintptr_t)p0 % 16) == 0,
              "czc_l2_from_init_forward: bad arguments");
  ProfScope prof((hipStream_t)stream, "l2_from_init_fwd", 0.0, 8.0 * n);
  const int nb = flat_grid(n / 4);  // <= 2048 partials
  hipLaunchKernelGGL(l2diff_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, p0, n, partials);
  hipLaunchKernelGGL(l2diff_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, weight_decay, out);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_l2_from_init_backward(const float* p, const float* p0, int64_t n, float weight_decay, const float* upstream,
                                         float* grads, int accumulate, czc_stream_t stream) {
  CZC_REQUIRE(n > 0 && p && p0 && grads && ((uintptr_t)p % 16) == 0 && ((uintptr_t)p0 % 16) == 0 && ((uintptr_t)grads % 16) == 0,
              "czc_l2_from_init_backward: bad arguments");
  ProfScope prof((hipStream_t)stream, "l2_from_init_bwd", 0.0, (accumulate ? 16.0 : 12.0) * n);
  hipLaunchKernelGGL(l2diff_grad_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, p0, n, 2.0f * weight_decay,
                     upstream, grads, accumulate);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_clip_coef(const float* sumsq, float pre_scale, float max_norm, float* coef, czc_stream_t stream) {
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, pre_scale, max_norm, coef);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_clamp(float* g, int64_t n, float lo, float hi, czc_stream_t stream) {
  CZC_REQUIRE(n > 0, "czc_clamp: n=%lld", (long long)n);
  hipLaunchKernelGGL(clamp_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, g, n, lo, hi);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int czc_cast_bf16(const float* src, uint16_t* dst, int64_t n, czc_stream_t stream) {
  CZC_REQUIRE(n > 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "czc_cast_bf16: n=%lld / alignment", (long long)n);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}
