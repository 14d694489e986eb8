/* libavec_hip.so -- C ABI of the MI355X (gfx950) hot path for the AV Efficient Conformer.
 *
 * Drop-in boundary: the reference reaches its native code through
 * PyTorch ops emitted by `nn.Module.forward` of the classes registered in
 *   nnet/layers.py:1372-1396 (layer_dict), nnet/attentions.py:739-746 (att_dict),
 *   nnet/normalizations.py:306-316 (norm_dict), nnet/activations.py:71-82 (act_dict),
 *   nnet/blocks.py:312-314 (block_dict), nnet/losses.py:363-366 (loss_dict),
 *   nnet/optimizers.py:184-189 (optim_dict).
 * Each entry point below replaces the ATen op group one of those forwards (or its autograd
 * backward) dispatches; the reference file:line it stands in for is cited per function.
 *
 * Conventions
 *   - plain C symbols, raw device pointers, explicit sizes; no torch types.
 *   - the library never allocates, frees or retains device memory; outputs/workspaces are the
 *     caller's.  Every launch goes to the `hipStream_t` argument, asynchronously (graph-capture safe).
 *   - return 0 on success, <0 for argument errors, >0 = hipError_t; text via avec_last_error().
 *   - `dtype`: AVEC_F32 (0) = fp32 storage + fp32 MFMA (exact; parity mode),
 *              AVEC_BF16 (1) = bf16 storage + bf16 MFMA with fp32 accumulate (throughput mode).
 *     "act" below means a buffer of that dtype; `float*` buffers are fp32 in both modes
 *     (residual stream, statistics, parameters' gradients).
 */
#ifndef AVEC_HIP_H
#define AVEC_HIP_H
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else
typedef struct ihipStream_t* hipStream_t;
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define AVEC_F32 0
#define AVEC_BF16 1
#define AVEC_ABI_VERSION 4      /* 2: avec_epilogue_t grew (bnb_*, res_cls0); avec_struct_size() handshake.  3: the row-resident module chains (avec_ffn_chain_*, avec_ln_gemm, avec_layernorm_*_sum) were removed.  4: avec_glu_dwconv_fwd_bn added */
#define AVEC_STAT_REPLICAS 64   /* `stats` buffers handed to avec_gemm_nt hold this many [2N] replicas (block b adds to replica b % 64) */

int avec_version(void);
/* sizeof() of the structs that cross this boundary, as THIS library was compiled: a binding compares them with its own declarations at load time, so that a header / .so
 * pair that drifted apart fails loudly instead of reading past a shorter struct.  which: 0 avec_rows_t, 1 avec_epilogue_t, 2 avec_attn_t, 3 avec_tn_item_t,
 * 4 avec_tn_batched_t, 5 avec_ln_item_t, 6 avec_fp8_item_t, 7 avec_wgrad3x3_item_t; -1 for an unknown index. */
int avec_struct_size(int which);
const char* avec_last_error(void);
/* the kernel instance chosen by the last GEMM-family entry point called on this thread ("gemm_nt_glds_kernel<bf16,64,64,0,4,0,128>" ...): measurement aid, bench.py's roofline rows */
const char* avec_last_kernel(void);
/* Optional scratch for the two-pass column reductions (LayerNorm / BatchNorm / depthwise-conv parameter-gradient sums): a 256-byte aligned
 * device buffer (>= 64 KB; 32 MB serves every shape of the AV model) registered for the current device.  Without it those kernels fall back
 * to one float atomic per column per workgroup.  The buffer must stay alive and must not be used by two streams at once.
 * (base = NULL, bytes = 0 unregisters.) */
int avec_set_reduce_workspace(void* base, long long bytes);
/* A second workspace bound to one stream: kernels launched on `stream` use it instead of the device default, so two streams (the audio and the
 * visual branch of the AV encoder) can run reductions concurrently. */
int avec_set_reduce_workspace_stream(void* base, long long bytes, hipStream_t stream);

/* ---- row sources for the GEMM family ------------------------------------------------------ */
enum { AVEC_ROWS_PLAIN = 0, AVEC_ROWS_CONV_FWD = 1, AVEC_ROWS_CONV_BWD = 2 };
typedef struct avec_rows {
  long long ld;                 /* PLAIN: row stride (elements) */
  int rows_out, rows_in, step;  /* PLAIN: src_row = (m / rows_out) * rows_in + (m % rows_out) * step when step > 1
                                   (strided time sub-sampling: the k=1 stride-2 conv_res, nnet/blocks.py:273-277) */
  int H, W, C, KH, KW, stride, pad, OH, OW; /* CONV_*: NHWC geometry; explicit "same" zero padding of nnet/layers.py:250-261 is folded into the loader */
  int T3;                       /* reserved */
} avec_rows_t;

typedef struct avec_epilogue {
  void* out; long long ldo; int out_f32;       /* final output: act, or fp32 when out_f32 */
  void* out_pre; long long ldpre;              /* optional: value before activation (act) */
  const float* bias;                           /* [N] */
  int act;                                     /* 0 none, 1 Swish (nnet/activations.py:39-45), 2 ReLU */
  float drop_p; const unsigned long long* rng; unsigned rng_stream;  /* nn.Dropout: counter-based mask, rng = {seed, step} on device */
  const void* res; long long ldres; float alpha; int res_act; /* out = res + alpha * v (res fp32, or act when res_act; residuals nnet/blocks.py:292-301) */
  const void* dact_z; long long ldz; int dact; /* backward: v *= act'(z) */
  float* colsum;                               /* += per-column sum of v  (bias gradients) */
  float* stats;                                /* += [N] sum, [N] sum of squares (BatchNorm batch statistics) */
  /* BatchNorm-BACKWARD fusion (round 3; nnet/normalizations.py:90-170): this product IS the gradient d of a BatchNorm(+ReLU) output whose input was `bnb_y`.
   * With bnb_y set the epilogue order is  v = alpha * acc + res;  v *= mask;  stats += (sum v | sum v * y);  out = v   -- the ReLU mask comes from the saved
   * activation `dact_z` (dact = 2: v where z > 0) or, with bnb_mask = 1, from the pre-activation itself (scale * y + shift > 0, bnb_ss = [scale | shift | ...]).
   * avec_bn_bwd_finalize turns the replicated (sum d, sum d*y) into (sum d, sum d*xhat); avec_bn_bwd_apply(act = 0) then needs d and y only. */
  const void* bnb_y; long long ldby; const float* bnb_ss; int bnb_mask;
  /* backward-data of a stride-2 convolution (bf16, parity-class order): `res` holds one row per pixel of class (even row, even column) only -- the gradient that a
   * 1x1 / stride-2 shortcut convolution sends to the same input (nnet/blocks.py: ResNetBlock.residual), computed as a plain product on the subsampled grid --
   * instead of a full-size, three-quarters-zero tensor.  Rejected (< 0) when the launch cannot run in parity-class order. */
  int res_cls0;
  /* ReLU mask on the RESIDUAL operand (round 6): one bit per element of `res` (bit e of byte (row * ldres + col) / 8 = column col + e, the layout avec_bn_apply_fwd_mask
   * writes): out = alpha * v + (bit ? res : 0).  The backward-data product of a ResNetBlock's first convolution adds the block-output gradient that passed the final ReLU
   * (nnet/blocks.py:88-91) -- with this the masked gradient is never written out as a tensor of its own.  bf16 residual, register-direct epilogue kernels only (an error
   * otherwise). */
  const unsigned char* res_mask;
} avec_epilogue_t;

/* C[m][n] = epi(sum_k A[m][k] W[n][k]).  Replaces aten::addmm/mm of layers.Linear.forward (nnet/layers.py:64-76),
 * the k=1 Conv1d of nnet/modules.py:374,379, aten::convolution / convolution_backward(input) of layers.Conv2d
 * (nnet/layers.py:200-324) and the Conv3d stem (nnet/layers.py:326-503). */
int avec_gemm_nt(int dtype, const void* A, const avec_rows_t* a_rows, int a_mode, int a_f32,
                 const void* W, long long ldw, long long M, int N, int K,
                 const avec_epilogue_t* ep, hipStream_t stream);

/* ---- fp8 (OCP e4m3) forward Linear products: BASELINE config 5's "fp8 GEMMs" for the FFN / QKV / output / pointwise-conv projections
 * (nnet/modules.py:257-289,341-385, nnet/attentions.py:60-110; the reference trains them under fp16 autocast, nnet/model.py:404-414).
 * Per-tensor "current" scaling: amax slots are plain device floats holding max|x|; scale = max(amax, tiny) / 448 is derived by the consumers. */
typedef struct { const float* src; void* dst; long long n; int slot; int K, Kp; int reserved; } avec_fp8_item_t;   /* fp32 master [n / K][K] -> e4m3 rows of Kp >= K bytes (zero padded, Kp % 16 == 0, K % 4 == 0), amax in amax[slot] */
/* x [M][K] (fp32 / bf16, row stride ldx) -> q [M][K] e4m3 (row stride ldq bytes); compute_amax != 0: *amax is first raised to max|x| (the caller zeroed it),
 * else *amax is taken as given.  K % 8 == 0; columns K .. roundup16(K) of q are written as zeros (ldq >= roundup16(K)). */
int avec_fp8_quantize(int src_dtype, const void* x, long long ldx, void* q, long long ldq, long long M, int K, float* amax, int compute_amax, hipStream_t stream);
/* every weight of `table` (device memory, n_items entries): amax[slot] = max|w| (caller zeroed the slots), then dst = e4m3(w / scale). */
int avec_fp8_weights_refresh(const avec_fp8_item_t* table, int n_items, int blocks_per_item, float* amax, hipStream_t stream);
/* C = epi( (sum_k A[m][k] W[n][k]) * scale_a * scale_w ), A and W e4m3 (row strides in bytes), epilogue fields as avec_gemm_nt with bf16 "act" buffers. */
int avec_gemm_nt_fp8(const void* A, long long lda, const void* W, long long ldw, long long M, int N, int K,
                     const float* amax_a, const float* amax_w, const avec_epilogue_t* ep, hipStream_t stream);


/* O[i][j] += sum_m P[m][i] Q[m][j]  (fp32 atomics, split over m).  Replaces the weight-gradient aten::mm of
 * Linear backward and convolution_backward(weight). */
int avec_gemm_tn(int dtype, const void* P, long long ldp, const void* Q, const avec_rows_t* q_rows, int q_mode, int q_f32,
                 float* O, long long ldo, long long M, int I, int J, hipStream_t stream);
/* batched form: batch = outer * nb_inner + inner; strides6 = element strides {P_outer, P_inner, Q_outer, Q_inner, O_outer, O_inner}.
 * Used for the attention backward bmm's (dK = dS^T Q, dV = P^T dO per (batch, head); dE_h = sum_b skew(dS)^T Q), nnet/attentions.py:300-315. */
/* avec_gemm_tn plus p_colsum[i] += sum_m P[m][i] (fp32, optional): the bias gradient of a Linear / 1x1 conv comes out of its weight-gradient GEMM
 * (nn.Linear backward: grad_bias = grad_output.sum(0)) */
int avec_gemm_tn_bias(int dtype, const void* P, long long ldp, const void* Q, const avec_rows_t* q_rows, int q_mode, int q_f32,
                      float* O, long long ldo, float* p_colsum, long long M, int I, int J, hipStream_t stream);
/* Grouped weight gradients: up to AVEC_TN_GROUP_MAX independent products O_k[I_k][J_k] += P_k^T Q_k (+ optional column sums of P_k = bias gradients) as ONE launch --
 * the ~10 weight-gradient products of a ConformerBlock backward (nnet/blocks.py:289-306: two FFN modules, attention projections, two pointwise convs) only feed the
 * optimizer, so the caller may queue them and submit them together; one grid over all their tiles fills the chip where each product alone is latency-bound.
 * bf16 only (avec_gemm_tn_grouped_ok tells; others go through avec_gemm_tn_bias).  No alignment requirement.  Operand rows are read in 16-byte chunks up to the next
 * multiple of 8 elements: when I (J) is not a multiple of 8 and ldp (ldq) is smaller than that multiple, the chunk runs into the next row, and behind the LAST row
 * the caller must keep 16 bytes readable (what is read there never reaches a stored result). */
#define AVEC_TN_GROUP_MAX 32
typedef struct avec_tn_item {
  const void* P; const void* Q; float* O; float* p_colsum;
  long long ldp, ldq, ldo, M;
  int I, J;
  int q_rows_out, q_rows_in, q_step;   /* strided row remap of Q as in avec_rows_t (q_step <= 1: identity) */
  int reserved;
} avec_tn_item_t;
int avec_gemm_tn_grouped_ok(int dtype, const avec_tn_item_t* item);
int avec_gemm_tn_grouped(int dtype, const avec_tn_item_t* items, int n, hipStream_t stream);
int avec_gemm_tn_batched(int dtype, const void* P, long long ldp, const void* Q, long long ldq, float* O, long long ldo, long long M, int I, int J,
                         int nb_outer, int nb_inner, const long long* strides6, hipStream_t stream);
/* (batched: as above, rows of P / Q are read in whole 16-byte chunks up to the next multiple of the vector width -- 8 bf16 / 4 fp32 elements -- of I / J, counted from the
 * batch's first column; with batches laid side by side in a row (attention heads of width 45 or 90) the last batch's last chunk of the LAST row ends up to 12 bytes behind the
 * operand: the caller keeps 16 bytes readable there) */
/* the same products STORED in the activation dtype (one workgroup per tile reduces over all M rows: no split, no atomics, no zero-filled fp32 staging):
 * dK and dV of the attention backward go straight into the Q|K|V gradient matrix */
int avec_gemm_tn_batched_store(int dtype, const void* P, long long ldp, const void* Q, long long ldq, void* O_act, long long ldo, long long M, int I, int J,
                               int nb_outer, int nb_inner, const long long* strides6, hipStream_t stream);

/* up to 3 such batched products (accumulating into fp32 `O`, or stored into `O_act` when that is set) as ONE launch: dK, dV and dE of an attention layer's
 * backward pass (nnet/attentions.py:280-323 backward) are three launches of ~5 us otherwise.  A problem that needs another kernel is launched on its own. */
typedef struct {
  const void* P; long long ldp; const void* Q; long long ldq;
  float* O; void* O_act; long long ldo;       /* exactly one of O (accumulate) / O_act (store) */
  long long M; int I, J, nb_outer, nb_inner;
  const long long* strides6;
} avec_tn_batched_t;
int avec_gemm_tn_batched_multi(int dtype, const avec_tn_batched_t* items, int n, hipStream_t stream);

/* ---- SyncBatchNorm statistic exchange by peer writes over xGMI (avec_amd/csrc/peer.hip) ----------
 * all-reduce(sum) of a short fp32 vector (the (2C+1)-float / 2C-float vectors of nnet/normalizations.py:172-249) between the GPUs of one node
 * without RCCL and without the host: every rank writes {value, epoch} granules into its slot of an exchange page in EVERY rank's buffer
 * (buffers mapped by all ranks through HIP IPC), polls its own page and adds the slots in rank order (bit-identical on all ranks).
 * pages[r] = base of this exchange site's page pair in rank r's buffer: 2 pages of world * n granules (8 bytes each), `page_stride_granules` apart;
 * epoch = this site's visit counter in the caller's device memory (zero-initialised, advanced by the kernel: graph-capturable);
 * err_flag is set to 1 if a peer did not arrive within timeout_ms (<= 0: 20 s) -- the call never hangs the GPU. */
#define AVEC_PEER_MAX_WORLD 8
int avec_peer_exchange_sum(const float* in, float* out, int n, void* const* pages, long long page_stride_granules, int rank, int world,
                           unsigned* epoch, int* err_flag, int timeout_ms, hipStream_t stream);
/* the same exchange with the producer-side work folded in (one launch instead of two per SyncBatchNorm exchange): the exchanged vector is
 *   v[i] = sum_{r < n_replicas} in[r * n_in + i] (i < n_in: the AVEC_STAT_REPLICAS statistic copies collapsed on the fly), v[n_in] = tail when has_tail (the local count);
 * with dgamma / dbeta set (backward exchange of [sum dy | sum dy * xhat], n_in = 2 C) the LOCAL sums are first added to the affine gradients, as avec_bn_affine_grads does.
 * out = [n_in + has_tail] sums over ranks. */
int avec_peer_exchange_sum_fused(const float* in, int n_replicas, int n_in, int has_tail, float tail, float* dgamma, float* dbeta, int C, float* out,
                                 void* const* pages, long long page_stride_granules, int rank, int world, unsigned* epoch, int* err_flag, int timeout_ms,
                                 hipStream_t stream);
int avec_enable_peer_access(int peer_device);
/* exchange-buffer management: uncached (fine-grained) device memory + its 64-byte HIP IPC handle; peers map it with _open.  The caller owns the pointers. */
int avec_peer_buffer_alloc(void** ptr, long long bytes, void* ipc_handle_64b);
int avec_peer_buffer_open(const void* ipc_handle_64b, void** ptr);
int avec_peer_buffer_close(void* ptr);
int avec_peer_buffer_free(void* ptr);

/* ---- normalisation / elementwise (avec_amd/csrc/norm.hip) ---------------------------------- */
/* nn.LayerNorm(eps=1e-6) forward/backward: aten::native_layer_norm(_backward) emitted by nnet/modules.py:278,302,373
 * and nnet/blocks.py:267.  x fp32 [M][D]; y act or fp32; dx optionally accumulated (residual merge). */
int avec_layernorm_fwd(int dtype, const float* x, const float* gamma, const float* beta, void* y, int y_f32,
                       float* mean, float* rstd, long long M, int D, float eps, hipStream_t stream);
int avec_layernorm_bwd(int dtype, const void* dy, int dy_f32, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, const float* dres, float* dgamma, float* dbeta, long long M, int D, hipStream_t stream);
/* dx-only LayerNorm backward with a second output  prep = act(prep_alpha * dropmask(rng, rng_stream, prep_drop_p) * dx):  the gradient the module IN FRONT of this
 * LayerNorm needs at the start of its own backward (out = res + alpha * Dropout(acc + bias), nnet/blocks.py:292-301; otherwise an avec_grad_prep launch per module) */
int avec_layernorm_bwd_prep(int dtype, const void* dy, int dy_f32, const float* x, const float* mean, const float* rstd, const float* gamma,
                            float* dx, const float* dres, void* prep, float prep_alpha, float prep_drop_p, const unsigned long long* rng, unsigned rng_stream,
                            long long M, int D, hipStream_t stream);
/* Two consecutive LayerNorms in one launch (D <= 512): the one that closes a ConformerBlock (nnet/blocks.py:267,303: y1 = LN1(x), fp32) and the one that opens the
 * next block's first feed-forward module (nnet/modules.py:278: h2 = LN2(y1), act) -- rows are independent, the second reads the first's registers.
 * Backward: dy2 (act) = gradient of h2, dres2 (fp32, may be NULL) = the gradient that reaches y1 past LN2 (the residual path); writes dx2 = dres2 + LN2'(dy2)
 * (= the gradient of y1: also LN1's dy for avec_layernorm_param_grads_grouped) and dx1 = LN1'(dx2), plus the optional prepared gradient of dx1 (see _bwd_prep). */
int avec_layernorm_fwd2(int dtype, const float* x, const float* gamma1, const float* beta1, float eps1, float* y1, float* mean1, float* rstd1,
                        const float* gamma2, const float* beta2, float eps2, void* h2, float* mean2, float* rstd2, long long M, int D, hipStream_t stream);
int avec_layernorm_bwd2(int dtype, const void* dy2, const float* x2, const float* mean2, const float* rstd2, const float* gamma2, const float* dres2, float* dx2,
                        const float* x1, const float* mean1, const float* rstd1, const float* gamma1, float* dx1,
                        void* prep, float prep_alpha, float prep_drop_p, const unsigned long long* rng, unsigned rng_stream, long long M, int D, hipStream_t stream);
/* avec_layernorm_bwd with dgamma == dbeta == NULL computes dx only (one wave per row); the parameter gradients of up to AVEC_LN_GROUP_MAX such layers are
 * then produced by ONE launch: dgamma_k[c] += sum_m dy_k[m][c] * xhat_k[m][c], dbeta_k[c] += sum_m dy_k[m][c]  (native_layer_norm_backward's weight / bias terms). */
#define AVEC_LN_GROUP_MAX 40
typedef struct avec_ln_item {
  const void* dy; const float* x; const float* mean; const float* rstd; float* dgamma; float* dbeta;
  long long M; int D; int dy_f32;      /* dy: fp32 when dy_f32 (or dtype == AVEC_F32), else act */
} avec_ln_item_t;
int avec_layernorm_param_grads_grouped(int dtype, const avec_ln_item_t* items, int n, hipStream_t stream);
/* backward of out = res + alpha*Dropout(acc + bias): dacc (act) and dbias; nnet/modules.py:286-288 + nnet/blocks.py:292-301 */
int avec_grad_prep(int dtype, const float* dout, long long ld, void* dacc, float alpha, float drop_p, const unsigned long long* rng,
                   unsigned rng_stream, float* dbias, long long M, int N, hipStream_t stream);
int avec_colsum(int dtype, const void* x, long long ld, float* out, long long M, int N, hipStream_t stream);
int avec_strided_rows_add(float* dx, const float* src, int B, int T, int To, int D, int step, hipStream_t stream);
/* BatchNorm{1,2,3}d over channels-last [M][C] (aten::native_batch_norm(_backward), nnet/normalizations.py:42-170).
 * stats = n_replicas x [sum | sumsq] (the GEMM epilogue spreads its atomics over AVEC_STAT_REPLICAS copies); ss = [scale | shift | mean | rstd]; SyncBatchNorm (:172-249) = all-reduce stats/count/dstats between calls. */
int avec_bn_stats(int dtype, const void* y, float* stats, long long M, int C, hipStream_t stream);
/* SyncBatchNorm glue (normalizations.SyncBatchNorm, nnet/normalizations.py:172-249): out[2C+1] = collapsed [sum | sumsq] + local count (the vector
 * the caller all-reduces); dgamma/dbeta += the LOCAL sums of dstats before dstats is all-reduced */
int avec_bn_collapse(const float* stats, int n_replicas, float count, float* out, int C, hipStream_t stream);
int avec_bn_affine_grads(const float* dstats, float* dgamma, float* dbeta, int C, hipStream_t stream);
int avec_bn_finalize(const float* stats, int n_replicas, const float* count_ptr, float count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, long long* num_batches_tracked, float momentum, float eps, float* ss, int C, int training, hipStream_t stream);
/* dstats[c] = sum over the n_replicas copies of stats[c], dstats[C + c] = rstd[c] * (sum of stats[C + c] - mean[c] * dstats[c]):  (sum d, sum d*y) -> (sum d, sum d*xhat) */
/* measurement aid: out[slot] = 100 MHz wall clock, written by a one-wave kernel on `stream` (graph-capturable) */
int avec_stamp(unsigned long long* out, int slot, hipStream_t stream);
int avec_bn_bwd_finalize(const float* stats, int n_replicas, const float* ss, float* dstats, int C, hipStream_t stream);
int avec_bn_apply_fwd(int dtype, const void* y, const float* ss, const void* residual, int act, void* out, long long M, int C, hipStream_t stream);
int avec_bn_bwd_reduce(int dtype, const void* dout, const void* y, const void* out, const float* ss, int act, float* dstats, long long M, int C, hipStream_t stream);
int avec_bn_bwd_apply(int dtype, const void* dout, const void* y, const void* out, const float* ss, const float* gamma, const float* dstats,
                      const float* count_ptr, float count, int act, void* dy, void* dres, float* dgamma, float* dbeta, long long M, int C, hipStream_t stream);
/* ReLU mask as one bit per element (C % 8 == 0; byte k = elements 8k .. 8k+7 of the row-major matrix, bit e = out[8k + e] > 0): the forward pass of a
 * residual BatchNorm + ReLU (out = relu(y*scale + shift + residual), the end of a ResNet block, nnet/blocks.py ResNetBlock) writes it next to `out`, the
 * backward passes read M*C/8 bytes instead of the whole saved output tensor (act = ReLU implied, otherwise like avec_bn_bwd_reduce / avec_bn_bwd_apply).
 * residual_ss (optional, [scale | shift | ..] of ANOTHER BatchNorm): the residual is that layer's raw input, normalised on the fly -- the projection shortcut
 * conv1x1 + BatchNorm of a down-sampling block without a tensor of its own: out = relu(y*scale + shift + residual*rscale + rshift). */
int avec_bn_apply_fwd_mask(int dtype, const void* y, const float* ss, const void* residual, const float* residual_ss, void* out, unsigned char* mask, long long M, int C,
                           hipStream_t stream);
int avec_bn_bwd_reduce_mask(int dtype, const void* dout, const void* y, const unsigned char* mask, const float* ss, float* dstats, long long M, int C, hipStream_t stream);
int avec_bn_bwd_apply_mask(int dtype, const void* dout, const void* y, const unsigned char* mask, const float* ss, const float* gamma, const float* dstats,
                           const float* count_ptr, float count, void* dy, void* dres, float* dgamma, float* dbeta, long long M, int C, hipStream_t stream);
/* softmax of the InterCTC residual (nnet/modules.py:395-400) */
int avec_softmax_fwd(int dtype, const float* logits, void* probs, long long M, int V, hipStream_t stream);
int avec_softmax_bwd(int dtype, const void* dprobs, const float* logits, float* dlogits, const float* dadd, long long M, int V, hipStream_t stream);
int avec_cast_rows(int dtype, const float* src, long long ld_src, void* dst, long long ld_dst, long long M, int N, hipStream_t stream);
int avec_to_f32_rows(int dtype, const void* src, long long ld_src, float* dst, long long ld_dst, long long M, int N, int accum, hipStream_t stream);
/* Stand-alone activations on fp32 tensors (nnet/activations.py:39-69; on the hot path they are epilogues of the producing kernels): act 1 Swish, 2 ReLU,
 * 3 GLU over the last axis (x [rows][2C] -> out [rows][C]).  backward != 0: out = d(loss)/dx from dy (GLU: out is [rows][2C]). */
int avec_act_f32(int act, const float* x, const float* dy, float* out, long long rows, int C, int backward, hipStream_t stream);
int avec_dropout_f32(const float* x, float* y, float p, const unsigned long long* rng, unsigned rng_stream, long long n, hipStream_t stream);
/* Stand-alone layers (avec_amd/csrc/standalone.hip): layers.MaxPool3d with a (1, KH, KW) window on a channels-last tensor [frames][H][W][C] (nnet/layers.py:839-915: zero
 * padding pad0 / pad1 before / behind each spatial axis, then a valid max pool; idx = winner slot kh*KW + kw, 255 = the zero padding), and layers.Upsample(mode="nearest")
 * on rows [B][T][D] -> [B][T*P][D] (nnet/layers.py:1013-1043; backward != 0: the P-row sums).  On the hot path both run fused (stem3p.hip, patch attention). */
int avec_maxpool_hw_fwd(int dtype, const void* x, void* out, unsigned char* idx, long long frames, int H, int W, int C, int KH, int KW, int SH, int SW,
                        int pad0_h, int pad0_w, int pad1_h, int pad1_w, hipStream_t stream);
int avec_maxpool_hw_bwd(int dtype, const void* dy, const unsigned char* idx, void* dx, long long frames, int H, int W, int C, int KH, int KW, int SH, int SW,
                        int pad0_h, int pad0_w, int pad1_h, int pad1_w, hipStream_t stream);
int avec_upsample_rows(int dtype, const void* src, void* dst, long long B, int T, int D, int P, int backward, hipStream_t stream);
/* patch attention pooling (layers.AvgPool1d / Upsample, nnet/attentions.py:342-346,365-380) */
int avec_patch_pool_fwd(int dtype, const void* x, void* y, int B, int T, int D, int P, hipStream_t stream);
int avec_patch_pool_bwd(int dtype, const void* dy, void* dx, int B, int T, int D, int P, hipStream_t stream);
int avec_patch_unpool_add(int dtype, const void* o, const float* res, float* out, float drop_p, const unsigned long long* rng, unsigned rng_stream,
                          int B, int T, int D, int P, hipStream_t stream);
int avec_patch_unpool_bwd(int dtype, const float* dout, void* dob, float drop_p, const unsigned long long* rng, unsigned rng_stream,
                          int B, int T, int D, int P, hipStream_t stream);
/* layers.GlobalAvgPool2d (nnet/layers.py:1328-1342), channels-last */
int avec_avgpool_fwd(int dtype, const void* x, void* y, long long N, int HW, int C, hipStream_t stream);
int avec_avgpool_bwd(int dtype, const void* dy, void* dx, long long N, int HW, int C, hipStream_t stream);

/* ---- conformer convolution module middle (avec_amd/csrc/convmod.hip) ----------------------- */
/* nn.GLU(dim=-1) + depthwise layers.Conv1d(k, groups=C, stride, padding) (nnet/modules.py:375-376); w is tap-major [K][C].
 * pad_left = zero frames in front of the sequence (nnet/layers.py:137-156): K / 2 for "same", K - 1 for "causal" (streaming), 0 for "valid"-like use;
 * the output always has (T - 1) / stride + 1 frames, i.e. the right padding is whatever completes K - 1. */
int avec_glu_dwconv_fwd(int dtype, const void* u, const float* w, const float* bias, void* out, float* stats,
                        int B, int T, int C, int K, int stride, int pad_left, hipStream_t stream);
/* avec_glu_dwconv_fwd followed by avec_bn_finalize in training mode over the B * To output rows (local batch statistics): the finalize reads the column-reduction partials
 * directly (two launches instead of three in the conformer block's dependent chain).  stats: zeroed [2C], used only when the reduction runs on atomics. */
int avec_glu_dwconv_fwd_bn(int dtype, const void* u, const float* w, const float* bias, void* out, float* stats, int B, int T, int C, int K, int stride, int pad_left,
                           const float* gamma, const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                           float* ss, hipStream_t stream);
int avec_dwconv_glu_bwd(int dtype, const void* dc, const void* u, const float* w, void* du, float* dw, float* dbias,
                        int B, int T, int C, int K, int stride, int pad_left, hipStream_t stream);
/* avec_bn_bwd_apply (act = Swish, local batch statistics: count = B * T rows, dstats = the (sum d, sum d * xhat) of avec_bn_bwd_reduce) folded into avec_dwconv_glu_bwd's
 * staging pass (stride 1): da = gradient of the BatchNorm + Swish output, c = the BatchNorm input; dgamma / dbeta += dstats.  One launch and one tensor less in the block's chain. */
int avec_dwconv_glu_bwd_bn(int dtype, const void* da, const void* c, const float* ss, const float* gamma, const float* dstats, float count, const void* u, const float* w,
                           void* du, float* dw, float* dbias, float* dgamma, float* dbeta, int B, int T, int C, int K, int pad_left, hipStream_t stream);

/* ---- attention (avec_amd/csrc/attention.hip) ------------------------------------------------ */
typedef struct avec_attn {
  const void *q, *k, *v; long long ld;        /* act [B*T][ld]; head h = columns [h*d,(h+1)*d) */
  const void* e; long long lde;               /* act [2T-1][lde]: pos_layer(sinusoid rows p = T-1 .. -(T-1)) (nnet/embeddings.py:101-158) */
  const long long* lens; int len_div;         /* key j kept iff j < lens[b]/len_div (Mask.padding_mask, nnet/attentions.py:682-733; patch min-pool :357-362) */
  int q_full;                                 /* query rows >= q_full are fully masked (zero-padded last patch); 0 = T */
  const float* mask; long long mask_bstride;  /* optional dense (Bm,T,T) 0/1 mask (1 = keep) instead of lens */
  void* o; long long ldo; float* lse;         /* outputs: act [B*T][ldo], fp32 [B*H][T][2] = softmax row (max, sum) */
  const void* dout;                           /* backward input, act [B*T][ldo] */
  void *dq, *dk, *dv; long long lddq, ldd;    /* act gradients: dq (row stride lddq), dk / dv (row stride ldd) */
  float* de; long long ldde;                  /* fp32 [2T-1][ldde], accumulated */
  void *pbuf, *dsbuf; long long ldt;          /* backward scratch, act [B*H][T][ldt] each (probabilities, dS) */
  void* dsrel; long long ldr;                 /* optional act [H][B*T][ldr] (zero-filled): dS indexed by E row; when given, dK/dV/dE are left to avec_gemm_tn_batched */
  int B, H, T, d; float scale;
  int Tk;                                     /* keys / values per batch element, 0 = T.  Tk > T: a key/value cache (`hidden`, nnet/attentions.py:506-512) precedes the T new frames:
                                                 k, v = act [B*Tk][ld], e = act [Tk+T-1][lde] (row r <-> relative position Tk-1-r), dense mask (Bm,T,Tk); forward and the probability
                                                 pass of avec_relpos_attention_bwd (pbuf) only */
} avec_attn_t;
/* RelPos1dMultiHeadAttention.forwardQKV core (nnet/attentions.py:299-315): bmm + rel_to_abs + mask + softmax + bmm */
int avec_relpos_attention_fwd(int dtype, const avec_attn_t* args, hipStream_t stream);
int avec_relpos_attention_bwd(int dtype, const avec_attn_t* args, hipStream_t stream);

/* ---- front-ends (avec_amd/csrc/frontend.hip) ------------------------------------------------ */
/* AudioPreprocessing (nnet/preprocessing.py:57-85; torchaudio Spectrogram/MelScale restated): frames -> [DFT via avec_gemm_nt fp32] -> power/mel/log */
int avec_mel_frames(const float* audio, const float* window, float* frames, int B, long long L, int n_fft, int win, int hop, hipStream_t stream);
int avec_mel_power_log(const float* spec, const float* fb, float* out, int B, int F, int n_bins, int n_mels, hipStream_t stream);
/* SpecAugment (nnet/preprocessing.py:115-130) */
int avec_specaugment(float* mel, const long long* lens, int B, int n_mels, int F, int mF, int Fparam, int mT, float pS,
                     const unsigned long long* rng, unsigned rng_stream, hipStream_t stream);
/* debugging / parity aid: out[i] = the uniform in [0, 1) that the counter-based generator of the dropout and SpecAugment kernels yields for (rng = {seed, step}, rng_stream,
 * ids[i]).  SpecAugment (avec_specaugment) draws, per frequency mask q, value = u(2q) * F and min = u(2q + 1) * (n_mels - value); per sample b and time mask q,
 * value = u(1000 + 64 b + 2q) * T_b and min = u(1000 + 64 b + 2q + 1) * (len_b - value): tests replay these draws through the oracle's mask_along_axis restatement and
 * compare the masks bit for bit. */
int avec_debug_rng_uniform(const unsigned long long* rng, unsigned rng_stream, const long long* ids, int n, float* out, hipStream_t stream);
/* audio stem Conv2d(1->C,3x3,s2,"same")+BatchNorm2d+Swish (nnet/networks.py:359-368) in (B,T',C*F') layout */
int avec_audio_stem_conv_fwd(int dtype, const float* mel, const float* w, const float* bias, void* y, float* stats, int B, int n_mels, int F, int C, hipStream_t stream);
int avec_audio_stem_act_fwd(int dtype, const void* y, const float* ss, void* a, int B, int n_mels, int F, int C, hipStream_t stream);
int avec_audio_stem_bwd(int dtype, const void* da, const void* y, const float* mel, const float* ss, const float* gamma, float* dstats,
                        const float* count_ptr, float count, int phase, float* dw, float* dbias, float* dgamma, float* dbeta,
                        int B, int n_mels, int F, int C, hipStream_t stream);
/* BatchNorm3d+ReLU+MaxPool3d((1,3,3),(1,2,2),"same") after the Conv3d stem (nnet/networks.py:459-470, nnet/layers.py:839-915) */
/* im2col of the Cin=1 (5,7,7)/(1,2,2) stem: video fp32 [clips][T][H][W] -> A act [clips*T*OH*OW][ldk] (k = (kd*7+kh)*7+kw, zero padded to ldk) */
int avec_stem_im2col(int dtype, const float* video, void* A, long long clips, int T, int H, int W, int ldk, hipStream_t stream);
/* Direct (no im2col) bf16 MFMA kernels for the same stem (avec_amd/csrc/stem3d.hip): the input band is staged once in LDS and the operands are
 * gathered from it.  w_shadow: bf16 [64][36][8] = per (kd,kh) row the 7 kw taps + a zero slot, row 35 zero (ldw = 288); y: bf16 [clips*T*OH*OW][64]; stats (optional):
 * fp32 [2*64] += (sum, sum of squares) of y per channel; dw: fp32 [64][245] +=.  avec_stem3d_supported() tells whether the frame size fits. */
int avec_stem3d_supported(long long clips, int T, int H, int W);
int avec_stem3d_fwd(const float* video, const void* w_shadow, int ldw, const float* bias, void* y, float* stats, long long clips, int T, int H, int W, hipStream_t stream);
int avec_stem3d_wgrad(const float* video, const void* dy, float* dw, long long clips, int T, int H, int W, hipStream_t stream);
/* Round 3: the same stem WITHOUT the pre-pool activation in memory (avec_amd/csrc/stem3p.hip).  BatchNorm + ReLU is monotone per channel, so the max pool is taken on the raw
 * conv output (max for gamma >= 0, min for gamma < 0) inside the convolution kernel:
 *   avec_stem3p_fwd   : video_bf16 [clips][T][H][W] (W % 8 == 0), w_shadow bf16 [64][36][8] (slot 0 of every (kd,kh) row and row 35 zero, slots 1..7 = kw 0..6), gamma = the
 *                       BatchNorm weight -> zp act [clips*T][PH][PW][64] (raw conv output of each window's winner), idx u8 (window slot kh*3+kw), stats fp32 [2*64] += (sum, sumsq) of z
 *                       (then avec_bn_finalize + avec_bn_apply_fwd(zp, ReLU) give the pooled activation);
 *   avec_stem3p_reduce: dpool <- dpool * [scale*zp+shift > 0] in place, dstats [2*64] += (sum d, sum d * xhat) over the pooled tensor;
 *   avec_stem3p_dz    : recomputes z and writes dz [clips*T*OH*OW][64] = gamma*rstd*(route(dpool) - mean(dy) - xhat*mean(dy*xhat)) (the operand of avec_stem3d_wgrad); dgamma/dbeta += dstats. */
int avec_stem3p_supported(long long clips, int T, int H, int W);
int avec_stem3p_fwd(const void* video_bf16, const void* w_shadow, const float* bias, const float* gamma, void* zp, unsigned char* idx, float* stats,
                    long long clips, int T, int H, int W, hipStream_t stream);
int avec_stem3p_reduce(void* dpool, const void* zp, const float* ss, float* dstats, long long frames, int PH, int PW, hipStream_t stream);
/* fused: dw fp32 [64][245] += the stem's weight gradient straight from the pooled gradients (z recomputed per tile, dz formed in LDS, never written) */
int avec_stem3p_wgrad_supported(long long clips, int T, int H, int W);
int avec_stem3p_wgrad(const void* video_bf16, const void* w_shadow, const float* bias, const void* dpool_masked, const unsigned char* idx, const float* ss,
                      const float* gamma, const float* dstats, const float* count_ptr, float count, float* dw, float* dgamma, float* dbeta,
                      long long clips, int T, int H, int W, hipStream_t stream);
int avec_stem3p_dz(const void* video_bf16, const void* w_shadow, const float* bias, const void* dpool_masked, const unsigned char* idx, const float* ss,
                   const float* gamma, const float* dstats, const float* count_ptr, float count, void* dz, float* dgamma, float* dbeta,
                   long long clips, int T, int H, int W, hipStream_t stream);
/* ymax (optional, act, pooled shape): the conv output under each window's winning tap -- lets avec_stem_pool_bwd_reduce_pooled compute the BatchNorm-backward sums from
 * pooled-size tensors instead of the full-resolution conv output */
int avec_stem_pool_fwd(int dtype, const void* y, const float* ss, void* out, unsigned char* idx, void* ymax, long long frames, int H, int W, int C, hipStream_t stream);
int avec_stem_pool_bwd_reduce_pooled(int dtype, const void* dpool, const unsigned char* idx, const void* ymax, const float* ss, float* dstats,
                                     long long frames, int H, int W, int C, hipStream_t stream);
int avec_stem_pool_bwd(int dtype, const void* dpool, const unsigned char* idx, const void* y, const float* ss, const float* gamma, float* dstats,
                       const float* count_ptr, float count, int phase, void* dy, float* dgamma, float* dbeta, long long frames, int H, int W, int C, hipStream_t stream);

/* ---- 3x3 / stride 1 / 64->64 channel convolution of ResNet stage 1 (avec_amd/csrc/conv3x3.hip) ------ */
/* Same contract as avec_gemm_nt with ROWS_CONV_FWD (flip = 0: x NHWC bf16, w = forward shadow [64][9][64], y NHWC bf16, stats = AVEC_STAT_REPLICAS x [sum | sumsq]
 * of the fp32 results) or ROWS_CONV_BWD (flip = 1: x = dy, w = backward shadow [Cin][9][Cout], y = dx, res = optional bf16 tensor added to dx) for layers.Conv2d
 * (nnet/layers.py:200-306) with kernel 3x3, stride 1, "same" padding, 64 input and output channels, images of at most 512 pixels whose zero-bordered slab fits
 * 72 KB of LDS ((H+2)(W+2) <= 576, even): weights stay in LDS, every input byte crosses L2 -> LDS once.  bf16 only. */
int avec_conv3x3_c64_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride);
int avec_conv3x3_c64(const void* x, const void* w, void* y, const void* res, float* stats, long long images, int H, int W, int flip, hipStream_t stream);
/* the same with a residual whose ReLU mask (one bit per element, as avec_epilogue_t.res_mask) is applied while it is added: y = conv(x) + (bit ? res : 0) */
int avec_conv3x3_c64_res_masked(const void* x, const void* w, void* y, const void* res, const unsigned char* res_mask, long long images, int H, int W, int flip, hipStream_t stream);
/* weight gradient of the same layers (what avec_gemm_tn with ROWS_CONV_FWD computes): dw fp32 [64][9][64] (row stride 576) += sum over images / pixels of
 * dy[p][co] * x[p + tap - 1][ci]; x, dy NHWC bf16.  Needs avec_conv3x3_c64_supported and H * (W + 1) <= 512. */
int avec_wgrad3x3_c64(const void* x, const void* dy, float* dw, long long images, int H, int W, hipStream_t stream);
/* the same for the wider 3x3 / stride-1 layers with Cin == Cout == C, C a multiple of 128 (ResNet stages 2..4: 128 / 256 / 512 channels, 11x11 / 6x6 / 3x3 images):
 * dw fp32 [C][9][C] (row stride 9C) += ...; supported when the per-image slabs fit (H (W+1) <= 288 rows of reduction, see csrc/conv3x3.hip) */
int avec_wgrad3x3_c128_supported(int H, int W, int Cin, int Cout, int KH, int KW, int stride);
int avec_wgrad3x3_c128(const void* x, const void* dy, float* dw, long long images, int C, int H, int W, hipStream_t stream);
/* the same for up to AVEC_WGRAD_GROUP_MAX layers in ONE launch: the 256 workgroups are shared out by work, and the final fp32 atomics (256 x 73 728 sums per launch,
 * ~30 % of a single layer's launch) are paid once for all of them.  A binding queues (x, dy, dw) of the 3x3 stride-1 layers during the backward pass (the operands must stay
 * alive and unmodified) and submits them when the ResNet's backward is through. */
#define AVEC_WGRAD_GROUP_MAX 16
typedef struct { const void* x; const void* dy; float* dw; long long images; int C, H, W, reserved; } avec_wgrad3x3_item_t;
int avec_wgrad3x3_c128_grouped(const avec_wgrad3x3_item_t* items, int n, hipStream_t stream);
int avec_wgrad3x3_c64_grouped(const avec_wgrad3x3_item_t* items, int n, hipStream_t stream);      /* the 64-channel layers (C = 64 in every item) */

/* ---- video input pipeline (avec_amd/csrc/video_input.hip) ------------------ */
/* Replaces, for a whole batch, the per-sample dataloader work of LRS.__getitem__ (nnet/datasets.py:187-196,348-356): uint8 -> float / 255, Grayscale,
 * NormalizeVideo (nnet/transforms.py:40-52), RandomCrop / CenterCrop + RandomHorizontalFlip (AV cfg:82-89), TimeMaskSecond (nnet/transforms.py:108-126),
 * align_video_to_audio (nnet/transforms.py:169-180) and CollateFn's zero padding (nnet/collate_fn.py:143-146).
 * clips: device uint8, clip b = [tv][H][W][channels] at byte offset clip_off[b];  geom: device int [B][8] = {tv, H, W, crop_y, crop_x, flip, zero frames in front, n_masks};
 * masks: device int [B][max_masks][2] = (first, one-past-last) frame of each time mask, in clip frames, in application order (null when max_masks == 0);
 * out: fp32 [B][Tout][OH][OW] (every frame outside [pad_left, pad_left + tv) is zero);  frame_ws: fp32 [2][B][Tout] scratch (only with masks).
 * lut: device fp32 [channels][256] = w_c * (u / 255) with w = (0.2989, 0.587, 0.114) for RGB clips and 1 for gray ones (made with the reference's own fp32 operations, so
 * that the device only adds the three terms, subtracts mean and divides by stdv: bit-identical pixels);  mean_frame != 0: mask k is filled with the mean of the clip as left by masks 0..k-1, else with 0. */
int avec_video_input(const unsigned char* clips, const long long* clip_off, const int* geom, const int* masks, int max_masks, int channels, const float* lut, float mean, float stdv,
                     int mean_frame, float* out, float* frame_ws, int B, int Tout, int OH, int OW, hipStream_t stream);

/* ---- loss / optimizer (avec_amd/csrc/loss_optim.hip) ---------------------------------------- */
/* CTCLoss.forward (nnet/losses.py:311-334): per-utterance -log p, batch mean, and d/dlogits (unscaled) */
long long avec_ctc_workspace_floats(int B, int T, int Lmax);
int avec_ctc_loss(const float* logits, const long long* in_lens, const long long* targets, const long long* tgt_lens, float* nll, float* mean_out,
                  float* grad, float* workspace, int B, int T, int V, int Lmax, int blank, int zero_infinity, hipStream_t stream);
/* the same for up to 8 heads that share the batch and the labels (ConformerInterCTC: nnet/networks.py:285-305 feeds six CTC losses), one launch; every head must
 * fit the all-LDS kernel (avec_ctc_loss_multi_fits).  Arrays of n_heads host pointers / frame counts. */
int avec_ctc_loss_multi_fits(int T, int V, int Lmax);
int avec_ctc_loss_multi(int n_heads, const float* const* logits, const long long* const* in_lens, const int* T, float* const* nll, float* const* mean_out, float* const* grad,
                        const long long* targets, const long long* tgt_lens, const float* weights, float* total, int B, int V, int Lmax, int blank, int zero_infinity, hipStream_t stream);
/* (weights: n_heads host floats, total: optional device scalar, += sum_i weights[i] * mean loss of head i -- the weighted total of nnet/model.py:275-287 out of the same launch) */
int avec_scale_by_scalar(const float* g, const float* scalar_dev, float mul, float* out, long long n, hipStream_t stream);
/* out[k] = g[k] * (*scalar_dev) * mul[k] for n_tensors <= 8 tensors in one launch (arrays of host pointers / sizes / factors): the backward of avec_ctc_loss_multi's
 * weighted total, nnet/model.py:275-287 */
int avec_scale_by_scalar_multi(int n_tensors, const float* const* g, float* const* out, const long long* numel, const float* mul, const float* scalar_dev, hipStream_t stream);
/* losses.SoftmaxCrossEntropy (nnet/losses.py:258-290; the LRW word classifier): per-row cross entropy of fp32 logits [M][V] against int64 targets,
 * rows with target == ignore_index give 0; mean_out (optional) += loss/M; grad (optional) = softmax - onehot */
int avec_softmax_ce(const float* logits, const long long* targets, long long ignore_index, float* loss, float* mean_out, float* grad, long long M, int V, hipStream_t stream);
/* length arithmetic of the strided layers (nnet/preprocessing.py:77, nnet/modules.py:127-128, nnet/networks.py:298,302): out[i] = floor((in[i] - sub) / div) + add, int64 */
int avec_len_affine(const long long* in, long long* out, int n, long long sub, long long div, long long add, hipStream_t stream);
/* CTCGreedySearchDecoder argmax (nnet/decoders.py:97-120) */
int avec_argmax_rows(const float* x, long long* out, long long M, int V, hipStream_t stream);
/* optimizers.Adam.step (nnet/optimizers.py:71-75) over flat arenas; state_dev = {step, lr} */
int avec_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* state_dev, float beta1, float beta2, float eps,
                   float weight_decay, float grad_scale, int zero_grad, long long n, hipStream_t stream);
/* the same with a device-side guard: *skip_flag != 0 (the SyncBatchNorm peer exchange's error flag: a rank did not arrive and the sums of this step are NaN)
 * leaves parameters and moments untouched and only clears the gradient arena -- the step is skipped instead of poisoning the model state */
int avec_adam_step_guarded(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* state_dev, float beta1, float beta2, float eps,
                           float weight_decay, float grad_scale, int zero_grad, long long n, const int* skip_flag, hipStream_t stream);
/* Compute-dtype copies of the GEMM weights (master fp32 [A][Tm][C]): fwd = same order, bwd = [C][Tm][A].  table entry = 10 x int64: src_off, fwd_off|-1,
 * bwd_off|-1, A, Tm, C, first_block, n_blocks = Tm*ceil(A/64)*ceil(C/64) (one workgroup per 64x64 tile and tap), C_pad (row stride of the fwd
 * shadow when Tm==1), bwd row pitch (0: Tm*A; larger when several weights share one [C][G*A] backward matrix, e.g. Q|K|V) */
int avec_shadow_refresh(int dtype, const float* master, void* shadow, const long long* table_dev, int n_entries, long long total_blocks, hipStream_t stream);
/* partial refresh: `table_dev` points at the first of `n_entries` consecutive entries, whose blocks are [first_block, first_block + n_blocks) of the full table
 * (the weights of one sub-network right after ITS optimizer launch, while the rest of the backward pass is still running) */
int avec_shadow_refresh_range(int dtype, const float* master, void* shadow, const long long* table_dev, int n_entries, long long first_block, long long n_blocks, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AVEC_HIP_H */
