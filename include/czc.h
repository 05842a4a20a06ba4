/* czc.h — C ABI of libczc.so: the MI355X (gfx950) MoCo contrastive-training hot path.
 *
 * The reference (Ajchler/czech-contriever) is pure Python and has no FFI of its own; its hot path
 * is the chain train.py:263-264,324 -> src/moco.py:296-353 -> src/contriever.py:17-55 (HF BertModel)
 * plus src/dist_utils.py:29-37 and src/utils.py:137-150.  Each entry point below names the reference
 * lines whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stubs a maintainer of the
 * reference would add to bind them.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch-allocated); the library allocates
 *     nothing persistent and never synchronises: all work is enqueued on `stream` (a hipStream_t
 *     passed as void*), so calls may be captured into a hipGraph.
 *   - bf16 tensors are raw uint16 storage; "f32" is float; ids are int64; masks are uint8 (0/1),
 *     which is the memory layout of torch.bool.
 *   - return value: 0 = CZC_OK, otherwise an error code; czc_last_error() gives the message
 *     (thread-local).  Nothing is launched when an argument check fails.
 *   - one process per GPU; calls are made from one host thread per process.
 */
#ifndef CZC_H_
#define CZC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CZC_OK 0
#define CZC_ERR_INVALID 1
#define CZC_ERR_HIP 2

#define CZC_ABI_VERSION 6

typedef void* czc_stream_t; /* hipStream_t */
typedef void* czc_event_t;  /* hipEvent_t (timing disabled) */

const char* czc_last_error(void);
int czc_abi_version(void);
/* sha1 (hex) over the library's sources, headers and compiler flags, fixed at build time (czech_contriever_amd/build.py:
 * source_hash): the loader compares it with the sources beside the binary and refuses -- or rebuilds -- a stale one. */
const char* czc_source_hash(void);

/* Stream ordering without torch types: events the caller creates once and reuses (hipEventDisableTiming).
 * czc_event_record(ev, s): ev marks the work enqueued on s so far; czc_stream_wait_event(s, ev): work enqueued on s
 * afterwards starts only once ev has fired.  Neither blocks the host.  Used for the key all-gather / enqueue on a side
 * stream (the queue columns an enqueue overwrites must have been read by this step's logits: write-after-read). */
int czc_event_create(czc_event_t* ev);
int czc_event_destroy(czc_event_t ev);
int czc_event_record(czc_event_t ev, czc_stream_t stream);
int czc_stream_wait_event(czc_stream_t stream, czc_event_t ev);

/* ---------------------------------------------------------------------------------------------
 * Encoder shape (HF BertConfig as read by src/moco.py:226-229 / src/contriever.py:11-13)
 * ------------------------------------------------------------------------------------------- */
typedef struct czc_bert_shape {
  int32_t vocab_size;
  int32_t hidden_size;        /* multiple of 64, <= 1024 */
  int32_t num_layers;
  int32_t num_heads;          /* hidden_size / num_heads must be 64 */
  int32_t intermediate_size;  /* multiple of 64 */
  int32_t max_position;
  int32_t type_vocab_size;
  float layer_norm_eps;
  float hidden_dropout;       /* HF hidden_dropout_prob (0.1) */
  float attn_dropout;         /* HF attention_probs_dropout_prob (0.1) */
} czc_bert_shape;

/* Flat parameter layout.  One encoder = one fp32 array of czc_param_count() elements:
 *   region A (matrices, also mirrored in bf16): per layer [Wqkv 3H*H | Wo H*H | W1 I*H | W2 H*I]
 *   region B (vectors): per layer [bqkv 3H | bo H | ln1.w H | ln1.b H | b1 I | b2 H | ln2.w H | ln2.b H]
 *   region C (embeddings): word V*H | pos P*H | type T*H | emb_ln.w H | emb_ln.b H
 * All Linear weights keep the HF [out,in] row-major orientation.
 * czc_param_offset(kind, layer) returns the element offset of one tensor. */
enum czc_param_kind {
  CZC_P_WQKV = 0, CZC_P_WO, CZC_P_W1, CZC_P_W2,                         /* region A */
  CZC_P_BQKV, CZC_P_BO, CZC_P_LN1W, CZC_P_LN1B, CZC_P_B1, CZC_P_B2, CZC_P_LN2W, CZC_P_LN2B, /* B */
  CZC_P_WORD, CZC_P_POS, CZC_P_TYPE, CZC_P_ELNW, CZC_P_ELNB,            /* region C */
  CZC_P_KINDS
};
int64_t czc_param_count(const czc_bert_shape* s);
int64_t czc_matrix_count(const czc_bert_shape* s); /* size of region A = length of the bf16 mirror */
int64_t czc_param_offset(const czc_bert_shape* s, int kind, int layer);

/* ---------------------------------------------------------------------------------------------
 * Encoder forward / backward  — replaces Contriever.forward (src/contriever.py:17-55) and its
 * autograd backward (train.py:264); arithmetic of HF BertEmbeddings/BertSelfAttention/
 * BertSelfOutput/BertIntermediate/BertOutput.
 * ------------------------------------------------------------------------------------------- */
#define CZC_POOL_AVERAGE 0
#define CZC_POOL_CLS 1

/* bytes of workspace czc_encoder_forward needs.  save_for_backward=1 keeps every activation the
 * backward re-reads (about (8H+2I)*2 bytes per token per layer); 0 keeps one layer's scratch. */
int64_t czc_encoder_workspace_bytes(const czc_bert_shape* s, int batch, int seq, int save_for_backward);
/* dst = the bf16 matrix region `src` (czc_matrix_count elements) with each layer's four weight matrices transposed in
 * place of themselves: Wqkv [3H,H] -> [H,3H], Wo [H,H], W1 [I,H] -> [H,I], W2 [H,I] -> [I,H].  H and I must be
 * multiples of 64.  One launch. */
int czc_transpose_matrices(const czc_bert_shape* s, const uint16_t* src, uint16_t* dst, czc_stream_t stream);

/* extra scratch (gradient temporaries, split-K slabs) czc_encoder_backward needs */
int64_t czc_encoder_backward_scratch_bytes(const czc_bert_shape* s, int batch, int seq);

/* emb[b, :] = pool(BERT(ids, mask)) ; optional L2 normalisation (F.normalize eps 1e-12).
 *   params_f32 : flat fp32 parameters (layout above);  params_bf16: bf16 mirror of region A
 *   ids  int64 [batch, seq];  mask uint8 [batch, seq] (1 = token, 0 = padding; key padding mask
 *   for attention, and pooling weights)
 *   training != 0 applies dropout with the counter RNG keyed by `seed` (pass the same seed to
 *   czc_encoder_backward).  workspace: czc_encoder_workspace_bytes(..) bytes, 256-B aligned.
 *   emb_out f32 [batch, hidden].
 *   split_batch > 0 (only with save_for_backward == 0, i.e. the key encoder): the LAST layer and the pooling run
 *   in two batch ranges, [0, split_batch) first; `split_event` (may be NULL) is recorded on `stream` as soon as
 *   emb_out[0 .. split_batch) is complete, while the rest of the batch is still being encoded -- the caller's side
 *   stream waits on it and starts the cross-GPU all-gather of those keys under this call's tail
 *   (src/moco.py:272-287 via src/dist_utils.py:29-37 in the reference).  Embeddings do not depend on the split
 *   (dropout draws of the second range use their own sites). */
int czc_encoder_forward(const czc_bert_shape* s, const float* params_f32, const uint16_t* params_bf16,
                        const int64_t* ids, const uint8_t* mask, int batch, int seq,
                        int pooling, int normalize, int training, uint32_t seed,
                        int save_for_backward, void* workspace, int64_t workspace_bytes,
                        float* emb_out, int split_batch, czc_event_t split_event, czc_stream_t stream);

/* grads_f32 (flat, same layout as params) += d(sum(emb * d_emb))/d(params)  when accumulate != 0,
 * or is overwritten when accumulate == 0.  `workspace` is the one the matching forward filled.
 * The backward may be issued in layer ranges [layer_lo, layer_hi), top range first (layer_hi = num_layers runs
 * the pooling backward, layer_lo = 0 the embedding backward; the inter-range gradient stays in `scratch`), so the
 * caller can all-reduce each finished range's gradients on another stream while the next range computes.
 * One call with (num_layers, 0) is the whole backward.
 * params_bf16_t (may be null): the matrix region with every weight matrix stored TRANSPOSED at its usual offset
 * (czc_transpose_matrices); with it the input-gradient products dX = dY.W read K-major fragments like the forward
 * ones (6 % faster at 65536 x 768 x 3072) instead of transposing W on the LDS read. */
int czc_encoder_backward(const czc_bert_shape* s, const float* params_f32, const uint16_t* params_bf16,
                         const uint16_t* params_bf16_t,
                         const int64_t* ids, const uint8_t* mask, int batch, int seq,
                         int pooling, int normalize, int training, uint32_t seed,
                         void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                         const float* d_emb, float* grads_f32, int accumulate, int layer_hi, int layer_lo,
                         czc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * MoCo head — replaces MoCo._compute_logits + "/ temperature" + F.cross_entropy + stats
 * (src/moco.py:289-294, 323-330, 340-351) and their backward w.r.t. q.
 * The negative queue is held slot-major in bf16: queue_bf16 [K, C] (row j = key j), so the
 * forward reads rows as the MFMA B operand and the enqueue writes whole rows.
 * ------------------------------------------------------------------------------------------- */
int64_t czc_infonce_workspace_bytes(int batch, int dim, int queue_size);

/* stats_out f32[8]: 0 loss (mean), 1 accuracy in percent, 2 stdq, 3 stdk (src/moco.py:343-351),
 * 4..7 reserved.  lse_out f32 [batch] is kept for the backward. */
int czc_infonce_forward(const float* q, const float* k, const uint16_t* queue_bf16,
                        int batch, int dim, int queue_size, float temperature, float label_smoothing,
                        void* workspace, int64_t workspace_bytes,
                        float* lse_out, float* stats_out, czc_stream_t stream);

/* dq f32 [batch, dim] = loss_scale * d(loss)/d(q) */
int czc_infonce_backward(const float* q, const float* k, const uint16_t* queue_bf16,
                         int batch, int dim, int queue_size, float temperature, float label_smoothing,
                         const float* lse, float loss_scale,
                         void* workspace, int64_t workspace_bytes, float* dq, czc_stream_t stream);

/* In-batch head — replaces src/inbatch.py:71-75 and its autograd: scores = (q / T) . keys^T over the keys
 * of ALL ranks, keys f32 [n_keys, dim] in rank order; the positive of local query i is column label_offset + i
 * (= rank * batch + i); F.cross_entropy with label smoothing, mean over the batch.  fp32 throughout, fixed summation
 * order.  stats_out f32[8]: 0 loss, 1 accuracy %, 2 stdq, 3 stdk (std over THIS rank's keys, rows label_offset ..).
 * dq f32 [batch, dim] and dkeys f32 [n_keys, dim] (either may be NULL) receive d loss / d q and d loss / d keys.
 * workspace: czc_inbatch_workspace_floats(batch, n_keys) floats. */
int64_t czc_inbatch_workspace_floats(int batch, int n_keys);
int czc_inbatch_head(const float* q, const float* keys, int batch, int n_keys, int dim, int label_offset,
                     float temperature, float label_smoothing, float* workspace, int64_t workspace_floats,
                     float* stats_out, float* dq, float* dkeys, czc_stream_t stream);

/* Ring-buffer enqueue — replaces MoCo._dequeue_and_enqueue after the gather (src/moco.py:277-287).
 * keys f32 [n, dim] (all ranks, rank order).  Writes rows [ptr, ptr+n) of queue_bf16 [K, dim] and,
 * when queue_f32 != NULL, the same keys into the reference-layout fp32 buffer queue_f32 [dim, K]
 * (columns ptr..ptr+n) that the checkpoint stores.  queue_ptr (int64[1], device) is read on the
 * device and advanced to (ptr + n) % K: no host synchronisation. */
int czc_enqueue(const float* keys, int n, int dim, int queue_size, uint16_t* queue_bf16,
                float* queue_f32, int64_t* queue_ptr, czc_stream_t stream);

/* queue_bf16 [K, dim] <- transpose(queue_f32 [dim, K])  (after load_state_dict / init) */
int czc_queue_import(const float* queue_f32, int dim, int queue_size, uint16_t* queue_bf16,
                     czc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Momentum EMA — replaces MoCo._momentum_update_key_encoder (src/moco.py:260-269) for the whole
 * flat parameter array in one launch:  pk = pk*m + pq*(1-m)  (two rounded products, one add, as
 * torch computes it), and refreshes the first n_bf16 elements of the bf16 mirror of pk.
 * ------------------------------------------------------------------------------------------- */
int czc_ema_update(float* pk, const float* pq, int64_t n, float m, float one_minus_m,
                   uint16_t* pk_bf16, int64_t n_bf16, czc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * AdamW — replaces torch.optim.AdamW.step as configured by utils.set_optim (src/utils.py:137-150):
 * decoupled weight decay, bias-corrected moments.  `step` counts from 1.  grad_scale multiplies
 * the gradient first (1/world_size for the all-reduced mean, times the clip coefficient read from
 * clip_coef_dev[0] when that pointer is not NULL).  Also refreshes the bf16 mirror.
 * ------------------------------------------------------------------------------------------- */
int czc_adamw_step(float* p, const float* g, float* m, float* v, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   float grad_scale, const float* clip_coef_dev,
                   uint16_t* p_bf16, int64_t n_bf16, czc_stream_t stream);

/* AdamW and, in the same pass, the momentum EMA the NEXT forward would start with (src/moco.py:314 applies it before every
 * key forward): pk = pk*ema_m + p_new*ema_one_minus_m while p_new is in registers, plus the key mirror pk_bf16 over the
 * same n_bf16 elements.  Bit-identical to czc_adamw_step followed by czc_ema_update; 12 bytes per parameter less traffic
 * and one launch fewer.  The caller then skips that forward's czc_ema_update (src/moco.MoCo: `_ema_prefolded`). */
int czc_adamw_ema_step(float* p, const float* g, float* m, float* v, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                       float grad_scale, const float* clip_coef_dev,
                       uint16_t* p_bf16, int64_t n_bf16, float* pk, uint16_t* pk_bf16, float ema_m,
                       float ema_one_minus_m, czc_stream_t stream);

/* sum of squares of g[0..n) -> out[0] (f32, device); out2 partial workspace of >= 4096 floats */
int czc_sumsq(const float* g, int64_t n, float* partials, float* out, czc_stream_t stream);
/* coef[0] = min(1, max_norm / (sqrt(sumsq[0]) * pre_scale + 1e-6))  (torch clip_grad_norm_) */
int czc_clip_coef(const float* sumsq, float pre_scale, float max_norm, float* coef, czc_stream_t stream);
int czc_clamp(float* g, int64_t n, float lo, float hi, czc_stream_t stream);

/* --weight_decay_from_init (src/moco.py:332-336, 355-356): out[0] = weight_decay * sum((p - p0)^2) over the flat arrays
 * (partials: >= 4096 floats of scratch; fixed summation order), and its gradient
 * grads (+)= 2 * weight_decay * upstream[0] * (p - p0)   (upstream: device scalar, NULL = 1; accumulate 0 overwrites). */
int czc_l2_from_init_forward(const float* p, const float* p0, int64_t n, float weight_decay, float* partials, float* out,
                             czc_stream_t stream);
int czc_l2_from_init_backward(const float* p, const float* p0, int64_t n, float weight_decay, const float* upstream,
                              float* grads, int accumulate, czc_stream_t stream);

/* dst_bf16[i] = bf16(src[i]) */
int czc_cast_bf16(const float* src, uint16_t* dst, int64_t n, czc_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Building blocks, exported so that tests can check each kernel against the oracle.
 * ------------------------------------------------------------------------------------------- */
#define CZC_LAYOUT_KMAJOR 0  /* reduction index contiguous:  A[M][K] / B[N][K] */
#define CZC_LAYOUT_MNMAJOR 1 /* free index contiguous:       A[K][M] / B[K][N] */

#define CZC_EPI_STORE 0      /* C bf16 = acc */
#define CZC_EPI_BIAS 1       /* C bf16 = acc + bias[n] */
#define CZC_EPI_BIAS_GELU 2  /* u = acc + bias ; C = gelu'(u) (what the backward needs of u) ; C2 = gelu(u) */
#define CZC_EPI_DGELU 3      /* C = acc * aux[m][n]  (aux = the gelu'(u) CZC_EPI_BIAS_GELU stored) */
#define CZC_EPI_ADD 4        /* C = acc + aux[m][n] */
#define CZC_EPI_F32 5        /* C f32 [splitk][M][ldc] = partial acc (no reduction) */
#define CZC_EPI_BIAS_GELU_ONLY 8 /* C = gelu(acc + bias), evaluated on the fp32 sum (passes nothing differentiates through:
                                  * the key encoder, inference); K-major operands only.  6, 7, 9 are internal (head, index) */

/* C[M,N] = A . B over K.  bf16 in, fp32 accumulate.  ld* in elements.  aux/C2 share ldc.
 * splitk > 1 only with CZC_EPI_F32 (slab z at C + z*M*ldc). */
int czc_gemm_bf16(int a_layout, int b_layout, int epilogue, int M, int N, int K,
                  const uint16_t* A, int lda, const uint16_t* B, int ldb,
                  void* C, int ldc, const float* bias, const uint16_t* aux, void* C2,
                  int splitk, czc_stream_t stream);

/* Two weight-gradient products in ONE launch: C f32 [splitk][M1 + M2][N], rows [0, M1) = A1^T . B1, rows [M1, M1 + M2) =
 * A2^T . B2, all operands MN-major ([K][M] / [K][N]: the reduction runs over tokens), M1 % 256 == 0.  The encoder's
 * backward computes dW[qkv | o] this way (they are neighbours in the flat gradient array). */
int czc_gemm_wgrad_pair(int M1, int M2, int N, int K, const uint16_t* A1, int lda1, const uint16_t* B1, int ldb1,
                        const uint16_t* A2, int lda2, const uint16_t* B2, int ldb2, float* C, int splitk,
                        czc_stream_t stream);

/* Tile configuration used by every later GEMM launch that does not choose one itself:
 * 0 = built-in choice, 1 = 256x256 tile (8 waves, 1 workgroup/CU), 2 = 128x128 (4 waves, 2 workgroups/CU);
 * + 16 = one workgroup per output tile instead of workgroups that walk over several tiles (tuning A/B; launches beside
 *        a collective);
 * + 32 / + 64 = the ping-pong K-loop always / never (default: chosen per shape);
 * + 512 / + 1024 / + 2048 = sc1 / nt / plain bf16 output stores (default: nt, streaming, for N >= 2048: a wide tile's
 *        128 KiB per CU otherwise competes with the operand panels for the XCD's L2) */
int czc_gemm_set_tile(int cfg);
/* Timing probe (bench only): records hipEvents on the launch stream around every czc GEMM launch whose
 * (layouts, epilogue, M, N, K) match, up to max_samples launches; max_samples <= 0 disarms.
 * czc_probe_read synchronises on the recorded events and returns their summed duration. */
int czc_probe_arm(int a_layout, int b_layout, int epilogue, int M, int N, int K, int max_samples);
int czc_probe_read(float* total_ms, int* count);
/* Per-operation timing (bench / profiling only, off by default): while enabled every data-path operation of this
 * library records a hipEvent pair on its launch stream under a name ("gemm a0 b0 epi2 65536x3072x768", "attention_bwd",
 * "residual_ln_fwd", "adamw", ...).  czc_prof_enable(on) clears what was collected.  czc_prof_read waits for the events
 * and writes one line per name:  name \t calls \t total_ms \t algorithmic FLOPs per call \t algorithmic bytes per call;
 * it returns the size the whole report needs (call with len = 0 to size the buffer). */
int czc_prof_enable(int on);
int64_t czc_prof_read(char* buf, int64_t len);

/* dst (+)= sum_z slabs[z]  (n elements each) */
int czc_reduce_slabs(const float* slabs, int nslabs, int64_t n, float* dst, int accumulate,
                     czc_stream_t stream);

/* qkv bf16 [batch*seq, 3H] (Q | K | V, head h at columns h*64) -> ctx bf16 [batch*seq, H];
 * lse f32 [batch, heads, seq] (natural log).  dropout_p on the probabilities when > 0. */
int czc_attention_forward(const uint16_t* qkv, const uint8_t* mask, int batch, int seq, int heads,
                          float dropout_p, uint32_t seed, uint32_t site,
                          uint16_t* ctx, float* lse, czc_stream_t stream);
/* dqkv bf16 [batch*seq, 3H];  delta f32 [batch, heads, seq] is scratch;  d_bias_qkv f32 [3H] (may be NULL)
 * += column sums of dqkv = the bias gradient of the fused QKV projection, accumulated through `partials`
 * (czc_attention_bias_scratch_floats floats of scratch: one row per workgroup, no contended atomics) */
int64_t czc_attention_bias_scratch_floats(int batch, int seq, int heads);
int czc_attention_backward(const uint16_t* qkv, const uint8_t* mask, const uint16_t* ctx,
                           const uint16_t* dctx, const float* lse, int batch, int seq, int heads,
                           float dropout_p, uint32_t seed, uint32_t site,
                           float* delta, uint16_t* dqkv, float* d_bias_qkv, float* partials, czc_stream_t stream);

/* The same with the dropout keep decisions kept between the two (seq <= 256): the forward writes
 * czc_attention_keep_words(batch, seq, heads) 32-bit words -- word [(b * heads + head) * 8 + q / 32][pos(key)] with
 * pos(k) = 32 (k / 32) + 8 ((k % 32) / 8) + 2 (k % 4) + (k / 4) % 2, bit q % 32 set iff probability (q, key) is kept:
 * the compare masks it forms anyway, in the order they form -- and the backward tests bits instead of regenerating the
 * draws (a third of its vector instructions).  keep_words NULL = the plain functions above. */
int64_t czc_attention_keep_words(int batch, int seq, int heads);
int czc_attention_forward_keep(const uint16_t* qkv, const uint8_t* mask, int batch, int seq, int heads,
                               float dropout_p, uint32_t seed, uint32_t site,
                               uint16_t* ctx, float* lse, uint32_t* keep_words, czc_stream_t stream);
int czc_attention_backward_keep(const uint16_t* qkv, const uint8_t* mask, const uint16_t* ctx,
                                const uint16_t* dctx, const float* lse, int batch, int seq, int heads,
                                float dropout_p, uint32_t seed, uint32_t site, const uint32_t* keep_words,
                                float* delta, uint16_t* dqkv, float* d_bias_qkv, float* partials, czc_stream_t stream);

/* x = dropout(LN(word[ids] + pos[s] + type[0]))  bf16 [M,H];  stats f32 [M,2] = (mean, rstd) */
int czc_embed_ln_forward(const int64_t* ids, int batch, int seq, int hidden, int vocab,
                         const float* word, const float* pos, const float* type0,
                         const float* gamma, const float* beta, float eps,
                         float dropout_p, uint32_t seed, uint32_t site,
                         uint16_t* x, float* stats, czc_stream_t stream);
int czc_embed_ln_backward(const int64_t* ids, int batch, int seq, int hidden, int vocab,
                          const float* word, const float* pos, const float* type0,
                          const float* gamma, const float* stats,
                          float dropout_p, uint32_t seed, uint32_t site, const uint16_t* dx,
                          float* d_word, float* d_pos, float* d_type0, float* d_gamma, float* d_beta,
                          float* partials, czc_stream_t stream);

/* x = LN(dropout(y) + resid) ; y already holds the dense bias */
int czc_residual_ln_forward(const uint16_t* y, const uint16_t* resid, int rows, int hidden,
                            const float* gamma, const float* beta, float eps,
                            float dropout_p, uint32_t seed, uint32_t site,
                            uint16_t* x, float* stats, czc_stream_t stream);
/* dz = dLN(dx [+ dx2]) ;  dz_out = dz (residual branch), dy_out = dropout-masked dz (dense branch);
 * d_gamma/d_beta/d_bias (+)= column sums.  dy_out may equal dz_out when p == 0.
 * partials: czc_colsum_scratch_floats(hidden) floats of scratch for per-workgroup partial sums (NULL = contended
 * fp32 atomics, several times slower on large inputs). */
int64_t czc_colsum_scratch_floats(int hidden);
int czc_reduce_partials(const float* partials, int nrows, int row_stride, int ncols, float* dst, czc_stream_t stream);
int czc_residual_ln_backward(const uint16_t* dx, const uint16_t* dx2, const uint16_t* y,
                             const uint16_t* resid, const float* stats, int rows, int hidden,
                             const float* gamma, float dropout_p, uint32_t seed, uint32_t site,
                             uint16_t* dz_out, uint16_t* dy_out,
                             float* d_gamma, float* d_beta, float* d_bias, float* partials, czc_stream_t stream);

/* col[n] += sum_m x[m][n]  (bias gradients) */
int czc_colsum_bf16(const uint16_t* x, int rows, int cols, int ld, float* col, czc_stream_t stream);

/* masked mean / CLS pooling (+ optional normalise); raw_out f32 [batch,H] keeps the
 * un-normalised embedding for the backward (may be NULL when normalize == 0) */
int czc_pool_forward(const uint16_t* x, const uint8_t* mask, int batch, int seq, int hidden,
                     int pooling, int normalize, float* emb, float* raw_out, czc_stream_t stream);
int czc_pool_backward(const float* d_emb, const float* raw, const uint8_t* mask, int batch, int seq,
                      int hidden, int pooling, int normalize, uint16_t* dx, czc_stream_t stream);

/* fill keep[i] (uint8) for element i of a dropout site: lets tests hand the oracle the same mask */
int czc_dropout_mask(uint32_t seed, uint32_t site, float p, int64_t n, uint8_t* keep, czc_stream_t stream);

/* ---- flat inner-product passage index: exact top-k search ---------------------------------------------------
 * Replaces faiss.IndexFlatIP.search as called by Indexer.search_knn, reference src/index.py:35-47 (driven from
 * passage_retrieval.py:184-186).  The index is whatever fp32 [rows, dim] blocks the caller keeps in HBM
 * (Indexer.index_data, src/index.py:25-31, one block per call); a search walks the blocks in order:
 *     czc_index_begin(state)  ->  czc_index_scan(block 0, fresh=1) -> czc_index_scan(block 1, fresh=0) ... ->
 *     czc_index_finish(state -> scores fp32 [nq,k] descending, ids int64 [nq,k])
 * State (device memory owned by the caller): cand_score fp32 [nq, cap], cand_id int32 [nq, cap],
 * cand_count int32 [nq], tau fp32 [nq], with cap >= czc_index_candidate_capacity(k).
 * Scores are fp32 products summed in fp32 (matrix cores, v_mfma_f32_32x32x2_f32); equal scores are ordered by
 * ascending id; a result shorter than k is padded with id -1 and score -FLT_MAX as faiss does.
 * dim % 16 == 0, 1 <= k <= 2048, id_base + rows < 2^31. */
#define CZC_INDEX_CHUNK 65536
int64_t czc_index_candidate_capacity(int32_t k);
int czc_index_begin(int32_t* cand_count, float* tau, int32_t nq, czc_stream_t stream);
int czc_index_scan(const float* db, int64_t rows, int32_t dim, int64_t id_base, const float* queries, int32_t nq,
                   int32_t k, int32_t fresh, float* cand_score, int32_t* cand_id, int32_t* cand_count, float* tau,
                   int64_t cap, czc_stream_t stream);
int czc_index_finish(const float* cand_score, const int32_t* cand_id, const int32_t* cand_count, int64_t cap,
                     int32_t nq, int32_t k, float* out_score, int64_t* out_id, czc_stream_t stream);

/* The same search with bf16 scoring and exact rescoring (same results, ~6x fewer matrix-core cycles).
 * Per index block the caller keeps, next to the fp32 rows, czc_index_prepare's output: a bf16 image [ceil8(rows), dim]
 * and the row norms [ceil8(rows)] (scale 1); per search, the same for the queries with scale =
 * czc_index_coarse_epsilon() (the norms then are the queries' error margins).  Order of calls:
 *   czc_index_coarse_begin -> czc_index_coarse_scan per block (fresh=1 on the first) -> czc_index_rescore (exact fp32
 *   scores of the survivors + the exact top k at the head of the lists) -> czc_index_finish.
 * State: cand_score / cand_err fp32 [nq, cap], cand_id int32 [nq, cap], cand_count / flags / kept int32 [nq], lower_bound / tau
 * fp32 [nq], cap = czc_index_coarse_capacity(k).  flags[q] != 0 afterwards: query q's survivor list overflowed (masses
 * of near-identical scores) and its result is not valid — search it with czc_index_scan instead.
 * blocks / block_starts: DEVICE arrays of the fp32 block pointers and of their first row ids (nblocks + 1 entries). */
int64_t czc_index_coarse_capacity(int32_t k);
float czc_index_coarse_epsilon(void);
int czc_index_prepare(const float* x, int64_t rows, int32_t dim, float scale, uint16_t* x_bf16, float* norms, czc_stream_t stream);
int czc_index_coarse_begin(int32_t* cand_count, float* lower_bound, int32_t* flags, int32_t* kept, int32_t nq, czc_stream_t stream);
int czc_index_coarse_scan(const uint16_t* db_bf16, const float* db_norms, int64_t rows, int32_t dim, int64_t id_base,
                          const uint16_t* q_bf16, const float* q_margin, int32_t nq, int32_t k, int32_t fresh,
                          float* cand_score, float* cand_err, int32_t* cand_id, int32_t* cand_count, float* lower_bound,
                          int32_t* flags, int32_t* kept, int64_t cap, czc_stream_t stream);
int czc_index_rescore(const float* const* blocks, const int64_t* block_starts, int32_t nblocks, int32_t dim,
                      const float* queries, int32_t nq, int32_t k, float* cand_score, int32_t* cand_id, int32_t* cand_count,
                      float* tau, int64_t cap, czc_stream_t stream);

/* ---- similarity-matrix distillation loss -----------------------------------------------------------------------
 * Replaces compute_sim_matrix x2 + F.mse_loss of the student side, reference distill.py:44-47,270-274:
 *   loss[0] = mean_ij (Sn_i.Sn_j - Tn_i.Tn_j)^2, Sn / Tn = L2-normalised rows of student [batch, dim_student] /
 *   teacher [batch, dim_teacher] (fp32);  d_student [batch, dim_student] = d loss / d student (NULL: loss only).
 * The caller applies DISTILL_LOSS_MULTIPLIER and the distillation weight.  dim_student <= 4096. */
int64_t czc_distill_workspace_floats(int32_t batch, int32_t dim_student, int32_t dim_teacher);
int czc_distill_sim_loss(const float* student, const float* teacher, int32_t batch, int32_t dim_student, int32_t dim_teacher,
                         float* workspace, int64_t workspace_floats, float* loss, float* d_student, czc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CZC_H_ */
