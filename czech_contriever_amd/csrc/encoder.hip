// encoder.hip — host-side orchestration of the BERT encoder forward / backward on one stream.
//
// Replaces Contriever.forward (src/contriever.py:17-55, i.e. HF BertModel.forward + pooling) and
// the autograd backward that train.py:264 triggers, as two C calls.  No allocation, no
// synchronisation: every kernel below is enqueued on the caller's stream, activations live in a
// caller-provided workspace laid out by enc_layout().
//
// Per layer, forward:   qkv = x Wqkv^T + b          (one fused [3H,H] GEMM, bias epilogue)
//                        ctx = attention(qkv)        (flash-style, attention.hip)
//                        y1  = ctx Wo^T + b ; x1 = LN(drop(y1) + x)
//                        g,h = gelu'(u), gelu(u), u = x1 W1^T + b  (GELU and its derivative fused in the GEMM epilogue)
//                        y2  = h W2^T + b  ; x'  = LN(drop(y2) + x1)
// Backward mirrors it; weight gradients are [N',K'] = dY^T X GEMMs whose operands are both read
// through transposing LDS reads, split over the token dimension into fp32 slabs.
#include "czc_common.h"
#include "gemm_args.h"

namespace czc {

static inline int64_t al256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct ParamLayout {
  int64_t LA, LB, regionA, regionB, total;
  int64_t H, I, L, V, P, T;
};
static ParamLayout param_layout(const czc_bert_shape* s) {
  ParamLayout p;
  p.H = s->hidden_size; p.I = s->intermediate_size; p.L = s->num_layers;
  p.V = s->vocab_size; p.P = s->max_position; p.T = s->type_vocab_size;
  p.LA = 4 * p.H * p.H + 2 * p.I * p.H;
  p.LB = 9 * p.H + p.I;
  p.regionA = p.L * p.LA;
  p.regionB = p.L * p.LB;
  p.total = p.regionA + p.regionB + (p.V + p.P + p.T + 2) * p.H;
  return p;
}
static int64_t param_off(const ParamLayout& p, int kind, int l) {
  const int64_t H = p.H, I = p.I;
  const int64_t a = l * p.LA, b = p.regionA + l * p.LB, c = p.regionA + p.regionB;
  switch (kind) {
    case CZC_P_WQKV: return a;
    case CZC_P_WO: return a + 3 * H * H;
    case CZC_P_W1: return a + 4 * H * H;
    case CZC_P_W2: return a + 4 * H * H + I * H;
    case CZC_P_BQKV: return b;
    case CZC_P_BO: return b + 3 * H;
    case CZC_P_LN1W: return b + 4 * H;
    case CZC_P_LN1B: return b + 5 * H;
    case CZC_P_B1: return b + 6 * H;
    case CZC_P_B2: return b + 6 * H + I;
    case CZC_P_LN2W: return b + 7 * H + I;
    case CZC_P_LN2B: return b + 8 * H + I;
    case CZC_P_WORD: return c;
    case CZC_P_POS: return c + p.V * H;
    case CZC_P_TYPE: return c + (p.V + p.P) * H;
    case CZC_P_ELNW: return c + (p.V + p.P + p.T) * H;
    case CZC_P_ELNB: return c + (p.V + p.P + p.T + 1) * H;
  }
  return -1;
}

static int shape_check(const char* who, const czc_bert_shape* s) {
  CZC_REQUIRE(s, "%s: null shape", who);
  CZC_REQUIRE(s->hidden_size > 0 && s->hidden_size % 64 == 0 && s->hidden_size <= 1024,
              "%s: hidden_size=%d must be a multiple of 64 and <= 1024", who, s->hidden_size);
  CZC_REQUIRE(s->num_heads > 0 && s->hidden_size == s->num_heads * 64,
              "%s: only head_dim 64 is built (hidden_size=%d, num_heads=%d)", who, s->hidden_size, s->num_heads);
  CZC_REQUIRE(s->intermediate_size > 0 && s->intermediate_size % 64 == 0, "%s: intermediate_size=%d must be a multiple of 64", who, s->intermediate_size);
  CZC_REQUIRE(s->num_layers > 0 && s->vocab_size > 0 && s->max_position > 0 && s->type_vocab_size > 0, "%s: bad shape", who);
  return CZC_OK;
}

struct LayerWs { int64_t qkv, ctx, lse, keep, y1, st1, x1, u, hact, y2, st2; };   // keep: attention dropout keep words (saved passes, seq <= 256), else -1
struct EncLayout {
  int64_t x[64 + 1];
  LayerWs layer[64];
  int64_t st0, raw, total;
};
static EncLayout enc_layout(const czc_bert_shape* s, int B, int S, int save) {
  EncLayout e;
  const int64_t M = (int64_t)B * S, H = s->hidden_size, I = s->intermediate_size, L = s->num_layers;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { int64_t r = o; o += al256(bytes); return r; };
  e.st0 = take(M * 2 * 4);
  e.raw = take((int64_t)B * H * 4);
  if (save) {
    for (int l = 0; l <= L; ++l) e.x[l] = take(M * H * 2);
  } else {
    const int64_t x0 = take(M * H * 2), x1 = take(M * H * 2);
    for (int l = 0; l <= L; ++l) e.x[l] = (l & 1) ? x1 : x0;
  }
  for (int l = 0; l < L; ++l) {
    if (save || l == 0) {
      LayerWs& w = e.layer[l];
      w.qkv = take(M * 3 * H * 2); w.ctx = take(M * H * 2); w.lse = take((int64_t)B * s->num_heads * S * 4);
      // keep words only where a saved pass can draw attention dropout at all (a shape with attn_dropout 0 never writes them)
      const int64_t kw = (save && s->attn_dropout > 0.f) ? czc_attention_keep_words(B, S, s->num_heads) : 0;
      w.keep = kw > 0 ? take(kw * 4) : -1;
      w.y1 = take(M * H * 2); w.st1 = take(M * 2 * 4); w.x1 = take(M * H * 2);
      w.u = take(M * I * 2); w.hact = take(M * I * 2); w.y2 = take(M * H * 2); w.st2 = take(M * 2 * 4);
    } else {
      e.layer[l] = e.layer[0];
    }
  }
  e.total = o;
  return e;
}

struct BwdLayout { int64_t g0, z, y, x1, c, q3, u, delta, part, part_ln1, part_gelu, part_attn, slabs, total; int64_t slab_bytes; };
static int dw_splits(int Mo, int No, int64_t Kred) {
  const int tiles = ceil_div(Mo, 256) * ceil_div(No, 256);
  int sp = 256 / tiles;
  if (sp < 1) sp = 1;
  const int kt = (int)ceil_div64(Kred, 64);
  if (sp > kt) sp = kt;
  return sp;
}
static BwdLayout bwd_layout(const czc_bert_shape* s, int B, int S) {
  BwdLayout b;
  const int64_t M = (int64_t)B * S, H = s->hidden_size, I = s->intermediate_size;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { int64_t r = o; o += al256(bytes); return r; };
  b.g0 = take(M * H * 2); b.z = take(M * H * 2); b.y = take(M * H * 2); b.x1 = take(M * H * 2);
  b.c = take(M * H * 2); b.q3 = take(M * 3 * H * 2); b.u = take(M * I * 2);
  b.delta = take((int64_t)B * s->num_heads * S * 4);
  {
    // per-workgroup partial column sums: LN kernels (2048 x 3H), attention QKV bias, dGELU bias (tiles_m*2 x I)
    int64_t pf = czc_colsum_scratch_floats((int)H);
    const int64_t af = czc_attention_bias_scratch_floats(B, S, s->num_heads);
    const int64_t gf = (int64_t)ceil_div64(M, 256) * 2 * I;
    if (af > pf) pf = af;
    if (gf > pf) pf = gf;
    b.part = take(pf * 4);
    // a layer's four sets of partial rows live side by side: they are reduced by ONE launch at the end of the layer
    b.part_ln1 = take(czc_colsum_scratch_floats((int)H) * 4);
    b.part_gelu = take(gf * 4);
    b.part_attn = take(af * 4);
  }
  int64_t sb = 0;
  auto need = [&](int Mo, int No) { int64_t v = (int64_t)dw_splits(Mo, No, M) * Mo * No * 4; if (v > sb) sb = v; };
  need(3 * H, H); need(H, H); need(I, H); need(H, I); need(4 * H, H);
  b.slab_bytes = sb;
  b.slabs = take(sb);
  b.total = o;
  return b;
}

#define SITE_EMB 1u
#define SITE_ATTN(l) (16u * ((l) + 1) + 0u)
#define SITE_H1(l) (16u * ((l) + 1) + 1u)
#define SITE_H2(l) (16u * ((l) + 1) + 2u)

static int gemm(int al, int bl, int epi, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                int ldc, const float* bias, const void* aux, void* C2, int splitk, hipStream_t st, float* colsum = nullptr) {
  GemmArgs a{};
  a.colsum = colsum;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.C2 = C2; a.bias = bias; a.aux = (const bf16_t*)aux;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  return gemm_dispatch(al, bl, epi, a, splitk, st);
}
#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// one 64 x 64 tile of one weight matrix per workgroup: rows of `src` in (16 B per lane), transposed through LDS,
// rows of `dst` out; block id -> (layer, matrix, tile) from the shape alone
__global__ __launch_bounds__(256) void transpose_matrices_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                                 int H, int I, int64_t layer_stride, int64_t o_qkv,
                                                                 int64_t o_o, int64_t o_1, int64_t o_2) {
  __shared__ bf16_t t[64][66];
  const int th = H / 64, ti = I / 64;
  const int per_layer = 3 * th * th + th * th + 2 * ti * th;
  int bid = blockIdx.x;
  const int layer = bid / per_layer;
  bid -= layer * per_layer;
  int rows, cols; int64_t off;
  if (bid < 3 * th * th)          { rows = 3 * H; cols = H; off = o_qkv; }
  else if ((bid -= 3 * th * th) < th * th) { rows = H; cols = H; off = o_o; }
  else if ((bid -= th * th) < ti * th)     { rows = I; cols = H; off = o_1; }
  else { bid -= ti * th;            rows = H; cols = I; off = o_2; }
  const int tc = cols / 64, tr = bid / tc, tcn = bid % tc;
  const bf16_t* s0 = src + layer * layer_stride + off + (size_t)(tr * 64) * cols + tcn * 64;
  bf16_t* d0 = dst + layer * layer_stride + off + (size_t)(tcn * 64) * rows + tr * 64;
  const int r = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint4 v = *(const uint4*)(s0 + (size_t)(r + 32 * h) * cols + c8);
    const bf16_t* e = (const bf16_t*)&v;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[r + 32 * h][c8 + j] = e[j];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = t[c8 + j][r + 32 * h];
    *(uint4*)(d0 + (size_t)(r + 32 * h) * rows + c8) = *(const uint4*)o;
  }
}

}  // namespace czc

using namespace czc;

extern "C" int64_t czc_param_count(const czc_bert_shape* s) { return s ? param_layout(s).total : -1; }
extern "C" int64_t czc_matrix_count(const czc_bert_shape* s) { return s ? param_layout(s).regionA : -1; }
extern "C" int64_t czc_param_offset(const czc_bert_shape* s, int kind, int layer) {
  if (!s || kind < 0 || kind >= CZC_P_KINDS || layer < 0 || layer >= s->num_layers) return -1;
  return param_off(param_layout(s), kind, kind >= CZC_P_WORD ? 0 : layer);
}
extern "C" int czc_transpose_matrices(const czc_bert_shape* s, const uint16_t* src, uint16_t* dst, czc_stream_t stream) {
  RC(shape_check("czc_transpose_matrices", s));
  CZC_REQUIRE(src && dst && src != dst, "czc_transpose_matrices: null or aliased arguments");
  CZC_REQUIRE(s->hidden_size % 64 == 0 && s->intermediate_size % 64 == 0,
              "czc_transpose_matrices: hidden %d / intermediate %d must be multiples of 64", s->hidden_size, s->intermediate_size);
  const ParamLayout pl = param_layout(s);
  const int H = s->hidden_size, I = s->intermediate_size, th = H / 64, ti = I / 64;
  const int per_layer = 3 * th * th + th * th + 2 * ti * th;
  const int64_t stride = s->num_layers > 1 ? param_off(pl, CZC_P_WQKV, 1) - param_off(pl, CZC_P_WQKV, 0) : 0;
  const int64_t base = param_off(pl, CZC_P_WQKV, 0);
  hipLaunchKernelGGL(transpose_matrices_kernel, dim3(per_layer * s->num_layers), dim3(256), 0, (hipStream_t)stream, src, dst, H, I,
                     stride, base, param_off(pl, CZC_P_WO, 0), param_off(pl, CZC_P_W1, 0), param_off(pl, CZC_P_W2, 0));
  CZC_LAUNCH_CHECK();
  return CZC_OK;
}

extern "C" int64_t czc_encoder_workspace_bytes(const czc_bert_shape* s, int batch, int seq, int save_for_backward) {
  if (!s || batch <= 0 || seq <= 0 || s->num_layers > 64) return -1;
  return enc_layout(s, batch, seq, save_for_backward).total;
}
extern "C" int64_t czc_encoder_backward_scratch_bytes(const czc_bert_shape* s, int batch, int seq) {
  if (!s || batch <= 0 || seq <= 0) return -1;
  return bwd_layout(s, batch, seq).total;
}

extern "C" int czc_encoder_forward(const czc_bert_shape* s, const float* params_f32, const uint16_t* params_bf16,
                                   const int64_t* ids, const uint8_t* mask, int batch, int seq, int pooling,
                                   int normalize, int training, uint32_t seed, int save_for_backward, void* workspace,
                                   int64_t workspace_bytes, float* emb_out, int split_batch, void* split_event,
                                   czc_stream_t stream) {
  RC(shape_check("czc_encoder_forward", s));
  CZC_REQUIRE(batch > 0 && seq > 0 && seq <= s->max_position && seq <= 512, "czc_encoder_forward: batch=%d seq=%d (max_position=%d, <= 512)", batch, seq, s->max_position);
  CZC_REQUIRE(s->num_layers <= 64, "czc_encoder_forward: at most 64 layers");
  CZC_REQUIRE(pooling == CZC_POOL_AVERAGE || pooling == CZC_POOL_CLS, "czc_encoder_forward: pooling=%d", pooling);
  CZC_REQUIRE(params_f32 && params_bf16 && ids && mask && emb_out, "czc_encoder_forward: null argument");
  const EncLayout e = enc_layout(s, batch, seq, save_for_backward);
  CZC_REQUIRE(workspace && ((uintptr_t)workspace % 256) == 0 && workspace_bytes >= e.total,
              "czc_encoder_forward: workspace %lld bytes < %lld needed (or not 256-B aligned)", (long long)workspace_bytes, (long long)e.total);
  hipStream_t st = (hipStream_t)stream;
  const ParamLayout pl = param_layout(s);
  const int H = s->hidden_size, I = s->intermediate_size, L = s->num_layers, nh = s->num_heads;
  const int M = batch * seq;
  char* ws = (char*)workspace;
  const float ph = training ? s->hidden_dropout : 0.f, pa = training ? s->attn_dropout : 0.f;
  const float* P = params_f32;
  const uint16_t* W = params_bf16;
#define PF(kind, l) (P + param_off(pl, kind, l))
#define WB(kind, l) (W + param_off(pl, kind, l))
#define BUF(off) ((uint16_t*)(ws + (off)))
#define FBUF(off) ((float*)(ws + (off)))

  CZC_REQUIRE(split_batch >= 0 && split_batch < batch, "czc_encoder_forward: split_batch=%d outside [0, %d)", split_batch, batch);
  CZC_REQUIRE(split_batch == 0 || !save_for_backward,
              "czc_encoder_forward: the split last layer is for passes nothing differentiates through (the key encoder)");
  RC(czc_embed_ln_forward(ids, batch, seq, H, s->vocab_size, PF(CZC_P_WORD, 0), PF(CZC_P_POS, 0), PF(CZC_P_TYPE, 0),
                          PF(CZC_P_ELNW, 0), PF(CZC_P_ELNB, 0), s->layer_norm_eps, ph, seed, SITE_EMB, BUF(e.x[0]),
                          FBUF(e.st0), stream));
  // layer l on sequences [b0, b0 + nb): every buffer is token-major, so a batch range is a row offset.  `sh` shifts the
  // dropout sites of a second range so that its masks are not those of the first (the draws are indexed from the
  // start of each launch's rows)
  auto layer = [&](int l, int b0, int nb, unsigned sh) -> int {
    const LayerWs& w = e.layer[l];
    const int Mr = nb * seq;
    const int64_t r0 = (int64_t)b0 * seq;
    uint16_t* x = BUF(e.x[l]) + r0 * H;
    uint16_t* qkv = BUF(w.qkv) + r0 * 3 * H; uint16_t* ctx = BUF(w.ctx) + r0 * H;
    uint16_t* y1 = BUF(w.y1) + r0 * H; uint16_t* x1 = BUF(w.x1) + r0 * H;
    uint16_t* u = BUF(w.u) + r0 * I; uint16_t* hact = BUF(w.hact) + r0 * I; uint16_t* y2 = BUF(w.y2) + r0 * H;
    RC(gemm(0, 0, EPI_BIAS, Mr, 3 * H, H, x, H, WB(CZC_P_WQKV, l), H, qkv, 3 * H, PF(CZC_P_BQKV, l), nullptr, nullptr, 1, st));
    // saved passes keep the dropout decisions of the probabilities as bit words for the backward (b0 == 0 there: the split
    // last layer is for passes nothing differentiates through)
    uint32_t* keep = (save_for_backward && w.keep >= 0 && pa > 0.f) ? (uint32_t*)(ws + w.keep) : nullptr;
    RC(czc_attention_forward_keep(qkv, mask + r0, nb, seq, nh, pa, seed, SITE_ATTN(l) + sh, ctx, FBUF(w.lse) + (int64_t)b0 * nh * seq,
                                  keep, stream));
    RC(gemm(0, 0, EPI_BIAS, Mr, H, H, ctx, H, WB(CZC_P_WO, l), H, y1, H, PF(CZC_P_BO, l), nullptr, nullptr, 1, st));
    RC(czc_residual_ln_forward(y1, x, Mr, H, PF(CZC_P_LN1W, l), PF(CZC_P_LN1B, l), s->layer_norm_eps, ph,
                               seed, SITE_H1(l) + sh, x1, FBUF(w.st1) + r0 * 2, stream));
    if (save_for_backward)
      RC(gemm(0, 0, EPI_BIAS_GELU, Mr, I, H, x1, H, WB(CZC_P_W1, l), H, u, I, PF(CZC_P_B1, l), nullptr, hact, 1, st));
    else   // nothing differentiates through this pass (key encoder, inference): keep gelu(u) only
      RC(gemm(0, 0, EPI_BIAS_GELU1, Mr, I, H, x1, H, WB(CZC_P_W1, l), H, hact, I, PF(CZC_P_B1, l), nullptr, nullptr, 1, st));
    RC(gemm(0, 0, EPI_BIAS, Mr, H, I, hact, I, WB(CZC_P_W2, l), I, y2, H, PF(CZC_P_B2, l), nullptr, nullptr, 1, st));
    RC(czc_residual_ln_forward(y2, x1, Mr, H, PF(CZC_P_LN2W, l), PF(CZC_P_LN2B, l), s->layer_norm_eps, ph,
                               seed, SITE_H2(l) + sh, BUF(e.x[l + 1]) + r0 * H, FBUF(w.st2) + r0 * 2, stream));
    return CZC_OK;
  };
  auto pool = [&](int b0, int nb) -> int {
    return czc_pool_forward(BUF(e.x[L]) + (int64_t)b0 * seq * H, mask + (int64_t)b0 * seq, nb, seq, H, pooling, normalize,
                            emb_out + (int64_t)b0 * H, FBUF(e.raw) + (int64_t)b0 * H, stream);
  };
  for (int l = 0; l < L - (split_batch ? 1 : 0); ++l) RC(layer(l, 0, batch, 0u));
  if (split_batch) {
    // The last layer and the pooling in two batch ranges: the embeddings of sequences [0, split_batch) are complete
    // -- `split_event` is recorded behind them -- while the rest of the batch is still being encoded, so the caller's
    // side stream can start their all-gather under this call's tail (src/moco.py, the key encoder under data parallelism)
    RC(layer(L - 1, 0, split_batch, 0u));
    RC(pool(0, split_batch));
    if (split_event) CZC_HIP(hipEventRecord((hipEvent_t)split_event, st));
    RC(layer(L - 1, split_batch, batch - split_batch, 4u));
    RC(pool(split_batch, batch - split_batch));
  } else {
    RC(pool(0, batch));
  }
  return CZC_OK;
}

extern "C" int czc_encoder_backward(const czc_bert_shape* s, const float* params_f32, const uint16_t* params_bf16, const uint16_t* params_bf16_t,
                                    const int64_t* ids, const uint8_t* mask, int batch, int seq, int pooling,
                                    int normalize, int training, uint32_t seed, void* workspace,
                                    int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, const float* d_emb,
                                    float* grads_f32, int accumulate, int layer_hi, int layer_lo, czc_stream_t stream) {
  RC(shape_check("czc_encoder_backward", s));
  CZC_REQUIRE(0 <= layer_lo && layer_lo < layer_hi && layer_hi <= s->num_layers,
              "czc_encoder_backward: layer range [%d, %d) outside [0, %d)", layer_lo, layer_hi, s->num_layers);
  CZC_REQUIRE(batch > 0 && seq > 0 && seq <= 512 && s->num_layers <= 64, "czc_encoder_backward: batch=%d seq=%d", batch, seq);
  CZC_REQUIRE(params_f32 && params_bf16 && ids && mask && d_emb && grads_f32, "czc_encoder_backward: null argument");
  const EncLayout e = enc_layout(s, batch, seq, 1);
  const BwdLayout bl = bwd_layout(s, batch, seq);
  CZC_REQUIRE(workspace && ((uintptr_t)workspace % 256) == 0 && workspace_bytes >= e.total,
              "czc_encoder_backward: workspace %lld bytes < %lld (it must be the save_for_backward=1 workspace)", (long long)workspace_bytes, (long long)e.total);
  CZC_REQUIRE(scratch && ((uintptr_t)scratch % 256) == 0 && scratch_bytes >= bl.total,
              "czc_encoder_backward: scratch %lld bytes < %lld", (long long)scratch_bytes, (long long)bl.total);
  hipStream_t st = (hipStream_t)stream;
  const ParamLayout pl = param_layout(s);
  const int H = s->hidden_size, I = s->intermediate_size, L = s->num_layers, nh = s->num_heads;
  const int M = batch * seq;
  char* ws = (char*)workspace;
  char* sc = (char*)scratch;
  const float ph = training ? s->hidden_dropout : 0.f, pa = training ? s->attn_dropout : 0.f;
  const float* P = params_f32;
  const uint16_t* W = params_bf16;
  const uint16_t* WT = params_bf16_t;   // transposed weight matrices at the same offsets, or null
  float* G = grads_f32;
#define WTB(kind, l) (WT + param_off(pl, kind, l))
  // dX = dY . W: with the transposed mirror W^T is [in][out] row-major, a K-major operand (b_layout 0, ld = out)
  auto dinput = [&](int epi, int Mx, int Nin, int Kout, const void* dY, int ldy, int kind, int l, void* dst, int ldd,
                    const void* aux, float* colsum) -> int {
    if (WT) return gemm(0, 0, epi, Mx, Nin, Kout, dY, ldy, WTB(kind, l), Kout, dst, ldd, nullptr, aux, nullptr, 1, st, colsum);
    return gemm(0, 1, epi, Mx, Nin, Kout, dY, ldy, WB(kind, l), Nin, dst, ldd, nullptr, aux, nullptr, 1, st, colsum);
  };
#define GF(kind, l) (G + param_off(pl, kind, l))
#define SBUF(off) ((uint16_t*)(sc + (off)))

  const bool first = layer_hi == L, last = layer_lo == 0;
  if (!accumulate && first) {
    // matrices are overwritten by the slab reduction; vectors and embeddings are accumulated by atomics
    CZC_HIP(hipMemsetAsync(G + pl.regionA, 0, (size_t)(pl.total - pl.regionA) * 4, st));
  }
  uint16_t* g0 = SBUF(bl.g0); uint16_t* Z = SBUF(bl.z); uint16_t* Y = ph > 0.f ? SBUF(bl.y) : Z;
  uint16_t* X1 = SBUF(bl.x1); uint16_t* C = SBUF(bl.c); uint16_t* Q3 = SBUF(bl.q3); uint16_t* U = SBUF(bl.u);
  float* delta = (float*)(sc + bl.delta);
  float* part = (float*)(sc + bl.part);
  float* part_ln1 = (float*)(sc + bl.part_ln1);
  float* part_gelu = (float*)(sc + bl.part_gelu);
  float* part_attn = (float*)(sc + bl.part_attn);
  float* slabs = (float*)(sc + bl.slabs);
  struct DeferGuard { DeferGuard() { g_defer_partials = 1; } ~DeferGuard() { g_defer_partials = 0; } };

  auto dweight = [&](int Mo, int No, const uint16_t* dY, int ldy, const uint16_t* X, int ldx, float* dst) -> int {
    const int sp = dw_splits(Mo, No, M);
    RC(gemm(1, 1, EPI_F32, Mo, No, M, dY, ldy, X, ldx, slabs, No, nullptr, nullptr, nullptr, sp, st));
    return czc_reduce_slabs(slabs, sp, (int64_t)Mo * No, dst, accumulate, stream);
  };
  // Two weight gradients that are neighbours in the flat array, one launch: dW[a | b] = [dYa | dYb]^T . [Xa | Xb] with
  // rows [0, Ma) from (dYa, Xa) and rows [Ma, Ma + Mb) from (dYb, Xb) (GemmArgs::A2).  Used for [Wqkv | Wo]: the 768 x 768
  // gradient alone has 9 output tiles -- 28-way split-K, 37 K-tiles per workgroup, 28 slabs: 745 TFLOP/s + 15 us of slab
  // sums against 1 100-1 230 for its neighbours; as rows 2304 .. 3071 of one 3072 x 768 product it runs at their rate.
  auto dweight2 = [&](int Ma, int Mb, int No, const uint16_t* dYa, int lda_, const uint16_t* Xa, int ldxa,
                      const uint16_t* dYb, int ldb_, const uint16_t* Xb, int ldxb, float* dst) -> int {
    const int Mo = Ma + Mb, sp = dw_splits(Mo, No, M);
    GemmArgs g{};
    g.A = (const bf16_t*)dYa; g.B = (const bf16_t*)Xa; g.lda = lda_; g.ldb = ldxa;
    g.A2 = (const bf16_t*)dYb; g.B2 = (const bf16_t*)Xb; g.lda2 = ldb_; g.ldb2 = ldxb; g.g_rows = Ma;
    g.C = slabs; g.M = Mo; g.N = No; g.K = M; g.ldc = No;
    RC(gemm_dispatch(1, 1, EPI_F32, g, sp, st));
    return czc_reduce_slabs(slabs, sp, (int64_t)Mo * No, dst, accumulate, stream);
  };
  // (the row tile is 256: the first product's rows must fill whole tiles; GF(WO) must follow GF(WQKV) -- it does, param_off)
  const bool qkvo_together = (3 * H) % 256 == 0 && H % 8 == 0;

  // the gradient w.r.t. the input of layer `layer_hi` stays in scratch (g0) between ranged calls
  if (first) RC(czc_pool_backward(d_emb, FBUF(e.raw), mask, batch, seq, H, pooling, normalize, g0, stream));
  for (int l = layer_hi - 1; l >= layer_lo; --l) {
    const LayerWs& w = e.layer[l];
    {
      DeferGuard defer;   // the four kernels below leave their partial rows; one reduction at the end of the layer
      // x[l+1] = LN(drop(y2) + x1)
      RC(czc_residual_ln_backward(g0, nullptr, BUF(w.y2), BUF(w.x1), FBUF(w.st2), M, H, PF(CZC_P_LN2W, l), ph, seed,
                                  SITE_H2(l), Z, Y, GF(CZC_P_LN2W, l), GF(CZC_P_LN2B, l), GF(CZC_P_B2, l), part, stream));
      RC(dweight(H, I, Y, H, BUF(w.hact), I, GF(CZC_P_W2, l)));
      RC(dinput(EPI_DGELU, M, I, H, Y, H, CZC_P_W2, l, U, I, BUF(w.u), part_gelu));
      RC(dweight(I, H, U, I, BUF(w.x1), H, GF(CZC_P_W1, l)));
      RC(dinput(EPI_ADD, M, H, I, U, I, CZC_P_W1, l, X1, H, Z, nullptr));
      // x1 = LN(drop(y1) + x[l])
      RC(czc_residual_ln_backward(X1, nullptr, BUF(w.y1), BUF(e.x[l]), FBUF(w.st1), M, H, PF(CZC_P_LN1W, l), ph, seed,
                                  SITE_H1(l), Z, Y, GF(CZC_P_LN1W, l), GF(CZC_P_LN1B, l), GF(CZC_P_BO, l), part_ln1, stream));
      if (!qkvo_together) RC(dweight(H, H, Y, H, BUF(w.ctx), H, GF(CZC_P_WO, l)));
      RC(dinput(EPI_STORE, M, H, H, Y, H, CZC_P_WO, l, C, H, nullptr, nullptr));
      RC(czc_attention_backward_keep(BUF(w.qkv), mask, BUF(w.ctx), C, FBUF(w.lse), batch, seq, nh, pa, seed, SITE_ATTN(l),
                                     (w.keep >= 0 && pa > 0.f) ? (const uint32_t*)(ws + w.keep) : nullptr, delta, Q3,
                                     GF(CZC_P_BQKV, l), part_attn, stream));
    }
    {
      const int lnr = ln_bwd_partial_rows(M);
      const int ar = attn_bwd_partial_rows(batch, seq);
      ReduceJob jobs[4] = {
          {part, lnr, 3 * H, 3 * H, H, {GF(CZC_P_LN2W, l), GF(CZC_P_LN2B, l), GF(CZC_P_B2, l)}},
          {part_gelu, ceil_div(M, 256) * 2, I, I, I, {GF(CZC_P_B1, l), nullptr, nullptr}},
          {part_ln1, lnr, 3 * H, 3 * H, H, {GF(CZC_P_LN1W, l), GF(CZC_P_LN1B, l), GF(CZC_P_BO, l)}},
          {part_attn, ar, 3 * H, 3 * H, 3 * H, {GF(CZC_P_BQKV, l), nullptr, nullptr}}};
      RC(multi_reduce(jobs, 4, st));
    }
    if (qkvo_together)   // Y (the gradient at the attention output projection's output) is untouched since the LayerNorm backward
      RC(dweight2(3 * H, H, H, Q3, 3 * H, BUF(e.x[l]), H, Y, H, BUF(w.ctx), H, GF(CZC_P_WQKV, l)));
    else
      RC(dweight(3 * H, H, Q3, 3 * H, BUF(e.x[l]), H, GF(CZC_P_WQKV, l)));
    RC(dinput(EPI_ADD, M, H, 3 * H, Q3, 3 * H, CZC_P_WQKV, l, g0, H, Z, nullptr));
  }
  if (last)
  RC(czc_embed_ln_backward(ids, batch, seq, H, s->vocab_size, PF(CZC_P_WORD, 0), PF(CZC_P_POS, 0), PF(CZC_P_TYPE, 0),
                           PF(CZC_P_ELNW, 0), FBUF(e.st0), ph, seed, SITE_EMB, g0, GF(CZC_P_WORD, 0), GF(CZC_P_POS, 0),
                           GF(CZC_P_TYPE, 0), GF(CZC_P_ELNW, 0), GF(CZC_P_ELNB, 0), part, stream));
  return CZC_OK;
}
