/*
 * ganffn.h — C ABI of libganffn.so: the MI355X (gfx950) implementation of GAN-FFN's
 * tri-modal generator/discriminator training step.
 *
 * The reference has no FFI: its seam is the Python nn.Module API of model.py as
 * consumed by train_IEMOCAP.py.  Every entry point below states the
 * reference code it replaces (file:line under /root/reference).  The Python side
 * (gan_ffn_amd/model.py, gan_ffn_amd/engine.py) binds these with ctypes and keeps the
 * reference's class names, constructor/forward signatures and state_dict layout.
 *
 * Conventions
 *  - All tensors are fp32, row-major, DEVICE pointers (hipMalloc'ed by the caller — in
 *    practice torch's caching allocator).  Activations are [T x C] with T = S*B tokens,
 *    token t = s*B + b, i.e. exactly the memory of the reference's (S, B, C) tensors
 *    (train_IEMOCAP.py:142-147).
 *  - The caller owns EVERY buffer (inputs, outputs, saved-for-backward, workspace,
 *    parameter/gradient/optimizer-state slabs).  The library allocates nothing, creates no
 *    streams or events, and keeps no mutable state between calls except caches that do not
 *    affect results — the thread-local last-error string and, per (kernel, device, host thread),
 *    the facts that hipFuncSetAttribute has opted a kernel in to > 48 KiB of LDS and how many
 *    workgroups of the persistent short-K GEMM the device holds at once — and ONE debug switch,
 *    ganffn_debug_set_ffn_mode (default 0; bits 0..6 select the older launch sequences of the
 *    d_model-100 sub-chains for A/B measurement, bits 8..19 force the chunk counts of two kernels
 *    — same results to rounding; documented at its declaration).  The library exports nothing
 *    that is not declared here: in-kernel time stamps and their ganffn_lab_* setters exist only in
 *    the lab build (`make LAB=1`, a different .so), and the library reads no environment variable.
 *  - Results are bit-reproducible: no kernel accumulates with floating-point atomics (weight
 *    gradients, bias gradients, LayerNorm gradients and loss sums are owner-computed or reduced
 *    in a fixed order), so two runs from the same state and RNG offset give identical bits.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the call
 *    returns without synchronising (so calls are capturable in a hipGraph).
 *  - Return value: 0 ok; < 0 argument/shape/alignment error (message via
 *    ganffn_last_error, thread-local); > 0 a hipError_t from a launch.
 *  - Parameters of one encoder stack live in ONE slab: L consecutive layer blocks of
 *    ganffn_layer_param_count(E, F) floats, each laid out per ganffn_layer_param_offsets.
 *    Gradient and Adam-state slabs use the same layout.  gan_ffn_amd/model.py exposes the
 *    slab through nn.Parameter views named like the reference's state_dict keys
 *    (transformer_encoder.layers.N.self_attn.in_proj_weight ...).
 *  - Dropout uses the counter-based Philox4x32-10 contract of oracle/philox.py /
 *    gan_ffn_amd/csrc/common.h (philox4, drop_mult4).  `rng` points to TWO device uint64: {seed, offset}; kernels read them
 *    at run time, so a captured graph replays with fresh masks after ganffn_rng_advance.
 */
#ifndef GANFFN_H
#define GANFFN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANFFN_VERSION 100 /* 0.1.0 */
#define GANFFN_MAX_SEQ 112 /* PositionalEncoding(max_len=110), model.py:1179, rounded up to 4 */

/* Encoder-stack configuration: PositionalEncoding + L post-LN nn.TransformerEncoderLayer
 * (model.py:1178-1197, 1210-1213).  p_* = 0 or train = 0 disables that dropout site. */
typedef struct ganffn_enc_cfg {
    int32_t S, B;       /* sequence length (<= 110) and batch (dialogues) */
    int32_t E, H, F, L; /* d_model, heads, dim_feedforward (2048), layers (8) */
    float p_pe;         /* PositionalEncoding dropout, 0.2 (model.py:1179) */
    float p_enc;        /* TransformerEncoderLayer dropout, 0.1 (torch default; model.py:1210) */
    float ln_eps;       /* 1e-5 */
    int32_t train;      /* 1: dropout active (module.train()), 0: eval */
} ganffn_enc_cfg;

/* Head configuration (the layers after the encoder stack).
 *  kind 0 = generator  : gelu -> drop -> gelu(drop(fc1)) -> gelu(drop(fc2))            model.py:1223-1228
 *  kind 1 = discriminator: gelu -> gelu(drop(fc1)) -> gelu(drop(fc2)) -> sigmoid(drop(fc3)) model.py:1322-1326 */
typedef struct ganffn_head_cfg {
    int32_t T;          /* tokens = S*B */
    int32_t E;          /* input width (d_model) */
    int32_t D1, D2;     /* fc1 / fc2 output widths (gen: 512|1024, D_h; disc: 64, 16) */
    int32_t kind;
    float p;            /* module dropout (0.2) */
    int32_t train;
} ganffn_head_cfg;

/* ---- library info ------------------------------------------------------------------ */
int ganffn_version(void);
const char* ganffn_last_error(void);

/* ---- parameter slab layout --------------------------------------------------------- */
/* floats per encoder layer; offsets[12] in this order:
 * in_proj_weight[3E,E] in_proj_bias[3E] out_proj.weight[E,E] out_proj.bias[E]
 * linear1.weight[F,E] linear1.bias[F] linear2.weight[E,F] linear2.bias[E]
 * norm1.weight[E] norm1.bias[E] norm2.weight[E] norm2.bias[E] */
int64_t ganffn_layer_param_count(int E, int F);
int ganffn_layer_param_offsets(int E, int F, int64_t* offsets12);

/* ---- sizes the caller must allocate ------------------------------------------------ */
int64_t ganffn_encoder_saved_floats(const ganffn_enc_cfg* cfg);     /* saved-for-backward */
int64_t ganffn_encoder_workspace_floats(const ganffn_enc_cfg* cfg); /* scratch, fwd or bwd */
/* float offset, inside `saved`, of layer `layer`'s FFN hidden activation h = drop(relu(x1 W1^T + b1)) [T x F]
 * (inspection hook: tests read the ReLU pattern the forward pass actually took); -1 on a bad cfg / layer */
int64_t ganffn_encoder_saved_hidden_offset(const ganffn_enc_cfg* cfg, int layer);
int64_t ganffn_head_saved_floats(const ganffn_head_cfg* cfg);
int64_t ganffn_head_workspace_floats(const ganffn_head_cfg* cfg);

/* ---- RNG state --------------------------------------------------------------------- */
/* rng[0] = seed, rng[1] = offset (device memory).  offset += delta, on the stream. */
int ganffn_rng_advance(uint64_t* rng, uint64_t delta, void* stream);

/* ---- A1: PositionalEncoding table (model.py:1182-1188) ----------------------------- */
/* pe[max_len x E] on device, computed as the reference does (fp32 sin/cos of p*exp(-2i ln1e4/E)). */
int ganffn_pe_table(float* pe, int max_len, int E, void* stream);

/* ---- A1+A2: encoder stack = PE + L encoder layers (model.py:1196-1197, 1224) -------- */
/* x_in [T x E]; pe [>=S x E]; params: L layer blocks; out [T x E] (output of the last layer);
 * saved: ganffn_encoder_saved_floats floats (needed by _bwd; pass NULL for inference-only,
 * then workspace is used and nothing is kept); rng_offset_add is added to rng[1] for this call. */
int ganffn_encoder_fwd(const ganffn_enc_cfg* cfg, const float* x_in, const float* pe,
                       const float* params, float* out, float* saved, float* workspace,
                       const uint64_t* rng, uint64_t rng_offset_add, void* stream);

/* Backward through layers [layer_lo, layer_hi) in reverse order.  dx [T x E] is in/out: on entry
 * dL/d(output of layer layer_hi-1), on exit dL/d(input of layer layer_lo); when layer_lo == 0 the
 * PE dropout backward is applied too, so dx is then dL/d(x_in).  grads (same layout as params)
 * is ACCUMULATED into (+=) when non-NULL; NULL skips every weight gradient (train_gen's pass
 * through the frozen discriminator, train_IEMOCAP.py:245-250: those grads are never used).
 * Splitting the range lets the caller start a layer's gradient all-reduce while earlier layers
 * are still in backward. */
int ganffn_encoder_bwd(const ganffn_enc_cfg* cfg, int layer_lo, int layer_hi, float* dx,
                       const float* params, float* grads, const float* saved, float* workspace,
                       const uint64_t* rng, uint64_t rng_offset_add, void* stream);

/* The same with need_dx_in = 0 when the stack's input needs no gradient (a network trained on data or on detached
 * fakes: train_disc / train_gen, train_IEMOCAP.py:200-252 — the networks' own inputs there are raw modalities or `.detach()`ed): with layer_lo == 0
 * the bottom layer's in-proj dgrad and the PE dropout backward are skipped, as autograd skips them for an input that does
 * not require grad, and dx is then left UNDEFINED on exit.  need_dx_in = 1 is ganffn_encoder_bwd. */
int ganffn_encoder_bwd2(const ganffn_enc_cfg* cfg, int layer_lo, int layer_hi, float* dx,
                        const float* params, float* grads, const float* saved, float* workspace,
                        const uint64_t* rng, uint64_t rng_offset_add, int need_dx_in, void* stream);

/* The whole-stack backward of a d_model-100 network with its weight gradients left UNREDUCED (round 5; the single-GPU step
 * runner): replaces `opt.zero_grad(); loss.backward()` + the gradient read of `opt.step()` of train_disc / train_gen
 * (train_IEMOCAP.py:216-226, 245-251) without the 14.5 MB zero-fill of the gradient slab and without the reduce launch.
 * The grouped weight-gradient launch cuts the token range into *n_parts chunks; chunk 0 WRITES (not +=) its partial
 * gradients into grads' encoder region, chunk z >= 1 into workspace + *part_offset + (z - 1) * *part_stride, a buffer shaped
 * like that region; the LayerNorm parameter gradients are written too.  So grads' encoder region [0, L * layer floats) need
 * NOT be zeroed by the caller (everything behind it — heads, `object` — still accumulates).  ganffn_adam_step_parts then adds
 * the chunks in order: exactly the sum, in exactly the association, that ganffn_encoder_bwd2's reduce launch forms — the
 * update is bit-identical.  The workspace must stay untouched between the two calls.
 * ganffn_encoder_bwd_parts_supported(cfg): 1 when this form exists for cfg (d_model 100, default kernel paths), else 0;
 * ganffn_encoder_bwd_parts_covered(E, F): floats at the head of each layer block that the chunks cover (weights and biases of
 * in-proj, out-proj, linear1, linear2; the LayerNorm parameters behind them are not chunked). */
int ganffn_encoder_bwd_parts_supported(const ganffn_enc_cfg* cfg);
int64_t ganffn_encoder_bwd_parts_covered(int E, int F);
int ganffn_encoder_bwd_parts(const ganffn_enc_cfg* cfg, float* dx, const float* params, float* grads, const float* saved,
                             float* workspace, const uint64_t* rng, uint64_t rng_offset_add, int need_dx_in,
                             int64_t* part_offset, int64_t* part_stride, int* n_parts, void* stream);

/* ---- A3-A6: heads ------------------------------------------------------------------- */
/* x [T x E] = encoder output.  w1[D1,E] b1[D1] w2[D2,D1] b2[D2]; disc only: w3[1,D2] b3[1].
 * out: gen -> fusion [T x D2]; disc -> prob [T x 1]. */
int ganffn_head_fwd(const ganffn_head_cfg* cfg, const float* x, const float* w1, const float* b1,
                    const float* w2, const float* b2, const float* w3, const float* b3, float* out,
                    float* saved, float* workspace, const uint64_t* rng, uint64_t rng_offset_add,
                    void* stream);
/* d_out: gen [T x D2], disc [T x 1].  dx [T x E] written.  g* accumulated (+=) when non-NULL. */
int ganffn_head_bwd(const ganffn_head_cfg* cfg, const float* d_out, const float* x,
                    const float* w1, const float* w2, const float* w3, float* gw1, float* gb1,
                    float* gw2, float* gb2, float* gw3, float* gb3, float* dx, const float* saved,
                    float* workspace, const uint64_t* rng, uint64_t rng_offset_add, void* stream);

/* ---- plain Linear (VisualDiscriminator.object model.py:1355-1356; GAN_FFN.fc model.py:1448) */
int ganffn_linear_fwd(const float* x, const float* w, const float* b, float* y, int T, int K, int N,
                      void* stream);
/* dx may be NULL; gw/gb accumulated (+=) when non-NULL.  workspace (may be NULL; ganffn_linear_bwd_workspace_floats
 * floats, 16-byte aligned) lets the weight gradient split the token range over more workgroups; the partial results are
 * reduced in a fixed order, so the result is bit-reproducible with or without it. */
int64_t ganffn_linear_bwd_workspace_floats(int T, int K, int N);
int ganffn_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* gw,
                      float* gb, int T, int K, int N, float* workspace, int64_t workspace_floats, void* stream);

/* ---- A10: BCELoss(mean) (train_IEMOCAP.py:300,220-223,248) -------------------------- */
/* loss_out[0] (+)= scale * mean_i( -(y*max(log p,-100) + (1-y)*max(log(1-p),-100)) ), y = target.
 * accumulate != 0 adds to loss_out (the D loss is (real + fake)/2: two calls with scale 0.5). */
int ganffn_bce_fwd(const float* prob, float target, int n, float scale, float* loss_out,
                   int accumulate, void* stream);
/* dprob[i] = scale/n * (p - y) / max(p*(1-p), 1e-12)   (torch's binary_cross_entropy_backward) */
int ganffn_bce_bwd(const float* prob, float target, int n, float scale, float* dprob, void* stream);
/* Two-valued periodic target: y_i = (i % period) < split ? target_a : target_b.  train_disc's loss
 * (BCE(D(real),1) + BCE(D(fake),0)) / 2 (train_IEMOCAP.py:220-223) is ONE call on the batched
 * [real | fake] pass (batch axis 2B: period 2B, split B, target_a 1, target_b 0, scale 1). */
int ganffn_bce2_fwd(const float* prob, float target_a, float target_b, int period, int split, int n,
                    float scale, float* loss_out, int accumulate, void* stream);
int ganffn_bce2_bwd(const float* prob, float target_a, float target_b, int period, int split, int n,
                    float scale, float* dprob, void* stream);

/* ---- A10: Adam over a flat slab (train_IEMOCAP.py:292-297; phase 2 :661) ------------- */
/* step: device int32 counter, incremented by this call BEFORE use (t = ++*step).
 * g = grad + weight_decay*p (L2-coupled, not AdamW).  n floats. */
int ganffn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                     int32_t* step, int64_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, float grad_scale, void* stream);
/* ganffn_adam_step over a slab whose encoder weight gradients ganffn_encoder_bwd_parts left unreduced: for element i of the
 * covered part of a layer block (i < enc_floats, i % layer_floats < covered_per_layer) the gradient is
 * grads[i] + parts[i] + parts[part_stride + i] + ... (n_parts - 1 extra chunks, in order); elsewhere grads[i].
 * n_parts == 1 is ganffn_adam_step. */
int ganffn_adam_step_parts(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t* step, int64_t n,
                           float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                           const float* parts, int64_t part_stride, int n_parts, int64_t enc_floats, int64_t layer_floats,
                           int64_t covered_per_layer, void* stream);
/* The same update on a SLICE of the slabs with t = *step + 1 and the counter left alone, and the bump on its own:
 * data-parallel training applies Adam bucket by bucket, each as soon as its gradient all-reduce has finished
 * (ganffn_adam_update per bucket, then one ganffn_adam_bump) — identical to one ganffn_adam_step over the whole slab. */
int ganffn_adam_update(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                       const int32_t* step, int64_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float grad_scale, void* stream);
int ganffn_adam_bump(int32_t* step, void* stream);

/* ---- A11: phase 2 (model.py:1441-1449, :74-81) --------------------------------------- */
/* fusion = a + b + c (elementwise, n floats) */
int ganffn_add3(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream);
/* logits [T x C] (token t = s*B+b) -> log_prob [T x C]; loss_out[0] = masked weighted NLL:
 * -sum_t w[y_t] m_t lp[t,y_t] / sum_t w[y_t] m_t, labels/umask are batch-major [B x S] as the
 * reference's collate makes them (dataloader.py:55-58); class_w may be NULL (unweighted).
 * dlogits (may be NULL) receives dL/dlogits. */
int ganffn_logsoftmax_nll(const float* logits, const int64_t* labels, const float* umask,
                          const float* class_w, float* log_prob, float* loss_out, float* dlogits,
                          float* workspace2, int S, int B, int C, void* stream);

/* ---- N2 (config 5): masked "general2" matching attention, all time steps as queries ---- */
/* Replaces BiModel.forward's loop over MatchingAttention(att_type="general2") (model.py:1043-1049 -> :169-182,
 * :193).  x = transform(mem) [S x B x D] (caller's linear), mem [S x B x D], mask [B x S] (umask).  Outputs:
 * att [S x B x D] pooled memory per query step, alpha [B x S x S] (dialogue, query step, memory step) — the
 * reference's per-step alpha lists stacked —, tanh_s [B x S x S] saved for backward.  S <= 128, D <= 1024.
 * Also the attention of MELDLSTMModel.forward (model.py:547-552). */
int ganffn_general2_attention_fwd(const float* x, const float* mem, const float* mask, float* att,
                                  float* alpha, float* tanh_s, int S, int B, int D, void* stream);
/* d_att [S x B x D] -> dx, dmem [S x B x D] (dmem = pooling path + score path; the gradient through
 * x = transform(mem) is the caller's linear backward).  du_ws: workspace of B*S*S floats. */
int ganffn_general2_attention_bwd(const float* d_att, const float* x, const float* mem, const float* mask,
                                  const float* alpha, const float* tanh_s, float* du_ws, float* dx,
                                  float* dmem, int S, int B, int D, void* stream);

/* ---- N2 (config 5): the DialogueRNN recurrence ------------------------------------------ */
/* Replaces DialogueRNN.forward / DialogueRNNCell.forward (model.py:828-972) in the configuration
 * train_IEMOCAP_DialogueRNN.py runs: context_attention = "general" (:586), listener_state = False (:595), two parties.
 * One call runs ndir (1 or 2) independent DialogueRNNs — BiModel's forward and reverse directions (model.py:1025-1033)
 * — through the same launches.  Per direction:
 *   U [S x B x D_m] (the reverse direction gets the reversed sequences, as BiModel._reverse_seq builds them),
 *   spk [S x B] int32 = argmax(qmask, party), mval [S x B] = qmask[s, b, spk] (0 on padded steps),
 *   e_out [S x B x D_e] emotion states (after dropout), alpha [B x S x S]: row t = attention weights of step t over the
 *   global history g_0 .. g_{t-1} (zero elsewhere; the reference's per-step alpha list is alpha[:, t, :t]).
 * Dropout (p = dropout_rec, train != 0) on g, the speaker's party state and e, Philox rows t*B + b. */
typedef struct ganffn_drnn_cfg {
    int32_t S, B;        /* steps (<= 112), dialogues (<= 32) */
    int32_t Dm, H, He;   /* D_m, D_g = D_p, D_e  (100, 500, 100; multiples of 4) */
    float p;             /* dropout_rec */
    int32_t train;
} ganffn_drnn_cfg;
/* the 13 parameter tensors of one DialogueRNNCell (state_dict names under dialogue_cell.): g_cell / p_cell / e_cell
 * weight_ih [3H x (D_m+H)] | [3H x (D_m+H)] | [3He x H], weight_hh [3H x H] | [3H x H] | [3He x He], bias_ih, bias_hh,
 * attention.transform.weight [H x D_m] */
typedef struct ganffn_drnn_params {
    const float *g_wih, *g_whh, *g_bih, *g_bhh, *p_wih, *p_whh, *p_bih, *p_bhh, *e_wih, *e_whh, *e_bih, *e_bhh, *att_w;
} ganffn_drnn_params;
typedef struct ganffn_drnn_grads {   /* accumulated into (+=); all NULL = input gradient only */
    float *g_wih, *g_whh, *g_bih, *g_bhh, *p_wih, *p_whh, *p_bih, *p_bhh, *e_wih, *e_whh, *e_bih, *e_bhh, *att_w;
} ganffn_drnn_grads;
int64_t ganffn_drnn_saved_floats(const ganffn_drnn_cfg* cfg);      /* per direction */
int64_t ganffn_drnn_workspace_floats(const ganffn_drnn_cfg* cfg);  /* per direction; fwd and bwd */
/* every array argument has ndir entries */
int ganffn_drnn_fwd(const ganffn_drnn_cfg* cfg, int ndir, const float* const* U, const int32_t* const* spk,
                    const float* const* mval, const ganffn_drnn_params* params, float* const* e_out,
                    float* const* alpha, float* const* saved, float* const* workspace, const uint64_t* rng,
                    uint64_t rng_offset_add, void* stream);
/* d_e [S x B x D_e] -> dU [S x B x D_m] (written), parameter gradients accumulated */
int ganffn_drnn_bwd(const ganffn_drnn_cfg* cfg, int ndir, const float* const* d_e, const float* const* U,
                    const int32_t* const* spk, const float* const* mval, const ganffn_drnn_params* params,
                    const ganffn_drnn_grads* grads, float* const* dU, const float* const* alpha,
                    const float* const* saved, float* const* workspace, const uint64_t* rng,
                    uint64_t rng_offset_add, void* stream);

/* Data movement of BiModel.forward around the recurrence (model.py:1008-1062), one launch each (csrc/drnn_head.hip):
 * ganffn_seq_reverse: out[s, b, :] (+)= s < lens[b] ? x[lens[b]-1-s, b, :] : 0 — BiModel._reverse_seq and, being its own
 *   transpose, its gradient (accumulate != 0 adds into out); x, out [S x B x D], D % 4 == 0.
 * ganffn_drnn_join_fwd: emotions [S x B x 2 D_e] = cat(dropout(e_f), dropout(reverse(e_b))) with p = dropout_rec + ...
 *   (BiModel.dropout_rec), two dropout sites; ganffn_drnn_join_bwd: the gradients of e_f and of e_b (in the reverse
 *   direction's own order) from d_emotions, same masks.
 * ganffn_mask_pos_inplace: d[i] = aux[i] > 0 ? d[i] * mscale : 0 — backward through dropout(relu(.)) from the saved output. */
int ganffn_seq_reverse(const float* x, const int32_t* lens, float* out, int S, int B, int D, int accumulate, void* stream);
int ganffn_drnn_join_fwd(const float* e_f, const float* e_b, const int32_t* lens, float* emotions, int S, int B, int De,
                         float p, uint32_t site_f, uint32_t site_b, const uint64_t* rng, uint64_t rng_offset_add, int train,
                         void* stream);
int ganffn_drnn_join_bwd(const float* d_emotions, const int32_t* lens, float* d_e_f, float* d_e_b, int S, int B, int De,
                         float p, uint32_t site_f, uint32_t site_b, const uint64_t* rng, uint64_t rng_offset_add, int train,
                         void* stream);
int ganffn_mask_pos_inplace(float* d, const float* aux, float mscale, int64_t n, void* stream);

/* one launch of the recurrence's skinny product, `copies` (<= 8) independent problems sharing A: C_i[M x N] = A[M x K]
 * W_i^T (nn = 0, W_i [N x K]) or A W_i (nn = 1, W_i [K x N]); W / C hold the copies back to back; M <= 32 (unit tests and
 * the roofline leg of bench.py --config drnn) */
int ganffn_drnn_skinny(int nn, int copies, const float* A, const float* W, float* C, int M, int N, int K, void* stream);

/* ---- building blocks exported for unit tests ----------------------------------------- */
/* C[M x N] = A[M x K] * W[N x K]^T + bias (bias may be NULL) */
int ganffn_gemm_nt(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                   void* stream);
/* C[M x N] = A[M x K] * B[K x N] */
int ganffn_gemm_nn(const float* A, const float* Bm, float* C, int M, int N, int K, void* stream);
/* C[M x N] += At[K x M]^T * B[K x N];  colsum[M] += sum_k At[k][m] when colsum != NULL */
int ganffn_gemm_tn_acc(const float* At, const float* Bm, float* C, float* colsum, int M, int N, int K,
                       void* stream);
/* h[T x F] = dropout_p(relu(x[T x E] * W1[F x E]^T + b1)): linear1 + activation + dropout of the encoder layer's
 * feed-forward block (torch TransformerEncoderLayer._ff_block; call site model.py:1210), one fused GEMM */
int ganffn_ffn_linear1_fwd(const float* x, const float* w1, const float* b1, float* h, int T, int E, int F,
                           float p, uint32_t site, const uint64_t* rng, uint64_t rng_offset_add, int train,
                           void* stream);
/* n (<= 40) independent problems C_i[M_i x N_i] += At_i[K_i x M_i]^T B_i[K_i x N_i] (+ column sums) in ONE launch: the
 * deferred weight-gradient GEMMs of all encoder layers of a backward pass (dense leading dimensions).  workspace
 * (ganffn_gemm_tn_grouped_workspace_floats() floats, or NULL): lets a group with few output tiles split the token range
 * over several workgroups per tile (partial slabs + one ordered reduce launch; deterministic).  NULL: one owner
 * workgroup per tile. */
int64_t ganffn_gemm_tn_grouped_workspace_floats(void);
int ganffn_gemm_tn_grouped(int n, const float* const* At, const float* const* Bm, float* const* C,
                           float* const* colsum, const int* M, const int* N, const int* K, float* workspace,
                           int64_t workspace_floats, void* stream);
/* Attention core of nn.MultiheadAttention (call sites model.py:1210,1244,1276,1307,1340,1377):
 * qkv [T x 3E] -> o [T x E]; site = dropout site id; p = 0 disables dropout.  lse [B*H x S] receives the log-sum-exp of
 * every score row (may be NULL when no backward follows).  The backward takes the forward's o and lse back (head_dim
 * <= 32 uses them instead of recomputing the softmax statistics; larger heads ignore them, NULL allowed there). */
int ganffn_attention_fwd(const float* qkv, float* o, float* lse, int S, int B, int E, int H, float p,
                         uint32_t site, const uint64_t* rng, uint64_t rng_offset_add, void* stream);
int ganffn_attention_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, float* d_qkv,
                         int S, int B, int E, int H, float p, uint32_t site, const uint64_t* rng,
                         uint64_t rng_offset_add, void* stream);
/* The same pair with the attention-dropout keep bits handed from the forward to the backward (what the encoder stack does
 * inside its saved-for-backward block): keep = ganffn_attention_keep_words(B, H) uint32, written by a forward with p > 0
 * and read by the backward INSTEAD of re-evaluating the Philox calls (head_dim <= 32; larger heads ignore it).  The
 * bits are the same, so both pairs give identical results; keep = NULL is exactly the pair above. */
int64_t ganffn_attention_keep_words(int B, int H);
int ganffn_attention_fwd_keep(const float* qkv, float* o, float* lse, uint32_t* keep, int S, int B, int E, int H,
                              float p, uint32_t site, const uint64_t* rng, uint64_t rng_offset_add, void* stream);
int ganffn_attention_bwd_keep(const float* qkv, const float* o, const float* lse, const float* d_o,
                              const uint32_t* keep, float* d_qkv, int S, int B, int E, int H, float p, uint32_t site,
                              const uint64_t* rng, uint64_t rng_offset_add, void* stream);
/* z = x + drop(y); xhat = (z-mean)*rstd; out = xhat*w + b   (norm1/norm2 of the encoder layer) */
int ganffn_add_dropout_layernorm_fwd(const float* x, const float* y, const float* w, const float* b,
                                     float* out, float* xhat, float* rstd, int T, int E, float eps,
                                     float p, uint32_t site, const uint64_t* rng,
                                     uint64_t rng_offset_add, void* stream);
/* d_out -> dz (residual branch) and dy = drop_bwd(dz); gw/gb accumulated when non-NULL */
int ganffn_add_dropout_layernorm_bwd(const float* d_out, const float* xhat, const float* rstd,
                                     const float* w, float* dz, float* dy, float* gw, float* gb, int T,
                                     int E, float p, uint32_t site, const uint64_t* rng,
                                     uint64_t rng_offset_add, void* stream);
/* out = keep ? x/(1-p) : 0 over a [R x C] tensor (Philox contract) — test hook for the mask layout */
int ganffn_dropout(const float* x, float* out, int R, int C, float p, uint32_t site,
                   const uint64_t* rng, uint64_t rng_offset_add, void* stream);

/* [T x K] x [K x 100] with a long K (linear2 forward / linear1 dgrad of the d_model-100 feed-forward block,
 * csrc/gemm_n100.hip): K is cut into *n_slabs (<= max_slabs <= 16) chunks, chunk z writes its partial product to
 * slabs + z * slab_stride ([T x 100], plain stores); the consumer adds the slabs in order (no atomics).  w_kmajor = 0:
 * W is [100 x K] (rows of K: out = A W^T), 1: W is [K x 100] (out = A W).  bias [100] (or NULL) is added by chunk 0.
 * K % 32 == 0, K >= 256. */
int ganffn_gemm_n100(const float* A, const float* W, int w_kmajor, const float* bias, float* slabs, int64_t slab_stride,
                     int T, int K, int max_slabs, int* n_slabs, void* stream);

/* Test / measurement hook: the generic 64 x 64-tile GEMM kernel exactly as the encoder stack launches it for the layers that
 * have no specialised kernel (the d_model-512 generator, nn.Linear call sites model.py:1244-1252 via torch's
 * TransformerEncoderLayer).  mode 0: C = A[M x K] W[N x K]^T, mode 1: C = A[M x K] W[K x N].  epi 0: + bias; 1: bias + ReLU +
 * dropout (linear1); 3: C = acc * (aux > 0 ? 1/(1-p) : 0) (dgrad through ReLU + dropout, aux = the saved activation [M x N]).
 * max_slabs > 1 with epi 0: K is split the way the encoder splits few-tile long-K products, slab z at C + z * slab_stride,
 * *n_slabs = slabs written (their sum is the product).  bench.py replays one iteration's launch mix through it. */
int ganffn_gemm_hook(int mode, int epi, const float* A, const float* W, const float* bias, const float* aux, float* C,
                     int64_t slab_stride, int M, int N, int K, float p, uint32_t site, const uint64_t* rng,
                     uint64_t rng_offset_add, int train, int max_slabs, int* n_slabs, void* stream);

/* ---- N4: one bidirectional LSTM layer (csrc/lstm.hip) -----------------------------------------------------------------
 * Replaces the recurrence of `nn.LSTM(input_size, hidden_size, num_layers = 4, bidirectional = True, dropout = p)` inside
 * MELDLSTMModel (/root/reference/model.py:520-562; built at /root/reference/train_MELD.py:147-151 with D_m = 600, D_e = 300),
 * one layer per call (the caller applies the inter-layer dropout and chains four calls).  x [S x B x In] padded, no mask (the
 * reference hands nn.LSTM the padded batch: model.py:546); out [S x B x 2H] = [h forward | h reverse]; torch's gate order
 * i, f, g, o: w_ih[d] [4H x In], w_hh[d] [4H x H], b_ih[d], b_hh[d] [4H], d = 0 forward, 1 reverse.  B <= 32 per call.
 * saved (ganffn_lstm_saved_floats): activated gates and cell states of every step, read by the backward;
 * workspace (ganffn_lstm_workspace_floats) is scratch.  The backward ACCUMULATES (+=) into gw_* / gb_* (any of them, or an
 * array, may be NULL) and writes dx [S x B x In] (NULL: not wanted).  Deterministic: no atomics. */
typedef struct ganffn_lstm_cfg {
    int32_t S, B, In, H;
} ganffn_lstm_cfg;
int64_t ganffn_lstm_saved_floats(const ganffn_lstm_cfg* cfg);
int64_t ganffn_lstm_workspace_floats(const ganffn_lstm_cfg* cfg);
int ganffn_lstm_layer_fwd(const ganffn_lstm_cfg* cfg, const float* x, const float* const* w_ih, const float* const* w_hh,
                          const float* const* b_ih, const float* const* b_hh, float* out, float* saved, float* workspace,
                          void* stream);
int ganffn_lstm_layer_bwd(const ganffn_lstm_cfg* cfg, const float* d_out, const float* x, const float* out,
                          const float* const* w_ih, const float* const* w_hh, float* dx, float* const* gw_ih,
                          float* const* gw_hh, float* const* gb_ih, float* const* gb_hh, const float* saved,
                          float* workspace, void* stream);

/* Measurement hook: the K = 100 -> 2048 products of the d_model-100 feed-forward block (csrc/gemm.hip gemm_wres_kernel) with the
 * arguments ganffn_encoder_fwd / _bwd give them (linear1 / linear2 of nn.TransformerEncoderLayer, call sites model.py:1210,1307).
 * which 0: out[T x 2048] = dropout_p(relu(a[T x 100] w[2048 x 100]^T + bias)); hmask != NULL: also the 1-bit [out > 0] pattern
 *          (ceil(T/32) * 2048 floats) a saved pass leaves for the backward.
 * which 1: out[T x 2048] = (a[T x 100] w[100 x 2048]) * pattern / (1-p) — the linear2 dgrad; pattern from hmask when given,
 *          else from h_saved [T x 2048] > 0. */
int ganffn_ffn_k100_hook(int which, const float* a, const float* w, const float* bias, float* out, void* hmask,
                         const float* h_saved, int T, float p, uint32_t site, const uint64_t* rng, uint64_t rng_offset_add,
                         int train, void* stream);

/* A/B measurement hook (process-wide), a bit mask; 0 = the default path.  It is the library's ONLY mutable process state besides
 * the per-(kernel, device) "LDS opt-in done" masks: one 32-bit word held in a relaxed atomic (csrc/common.h `Mode`); every entry
 * point takes one snapshot of it per call, so a setter racing with a launch makes that launch take one path or the other, never
 * a mixture.  Bits 8..23 are lab knobs (forced tile / chunk counts), the others select an older or alternative launch
 * sequence that stays parity-tested. */
/*
 *   bits 0, 7, 22: reserved, ignored (until round 5 they selected the fused feed-forward kernels ffn.hip / ffn3.hip: measured
 *          slower in the step three times over and removed from the product; the kernels are in the history);
 *   bit 1: run the token-local chains around the LayerNorms of a d_model-100 layer (out-proj + LN1, LN2 + next in-proj and
 *          their backward mirrors, csrc/rowchain.hip) as separate GEMM + LayerNorm launches;
 *   bit 2: run the [T x 2048] x [2048 x 100] products on the generic 64 x 64 tiles instead of csrc/gemm_n100.hip;
 *   bit 3: run the grouped weight-gradient launch of a d_model-100 pass on the generic 64 x 64 tiles instead of
 *          csrc/gemm_tn100.hip;
 *   bit 4: csrc/gemm_tn100.hip adds its partial slabs in the kernel's last-arriving workgroup per tile instead of in a second
 *          launch (same bits; measured slower: the device-scope fences cost more than the launch);
 *   bit 5: run the discriminator head as separate GELU / GEMM / tail launches instead of csrc/disc_head.hip;
 *   bit 6: run the head of a d_model-100 encoder stack (positional encoding + dropout, layer 0's in-proj) as two launches
 *          instead of one (csrc/rowchain.hip);
 *   bits 8..15: forced K-chunk count of csrc/gemm_n100.hip (0 = choose; clamped to the caller's slab capacity);
 *   bits 16..19: forced token-chunk count of csrc/gemm_tn100.hip (0 = choose; clamped to 8 and to the workspace);
 *   bits 20..21: csrc/gemm_n100.hip with 4 (value 1) or 8 (value 2) waves per workgroup (0 = choose);
 *   bit 23: csrc/gemm_n100.hip and csrc/gemm_tn100.hip with rows 96..99 of the 100-wide dimension on a padded seventh 16-row
 *           MFMA tile (round 3's form) instead of v_mfma_f32_4x4x1_16B_f32 (measured: the step 34.4 -> 33.75 ms without it).
 *   bit 24: run the out-proj of the wide (d_model != 100) stacks unsplit instead of as two K halves summed by the LayerNorm kernel;
 *   bit 25: the linear2 dgrad takes its ReLU / dropout pattern from the saved hidden activation instead of the 1-bit copy that
 *           linear1's epilogue leaves beside it in the saved block (same predicate, same bits; 24.6 MB less to read per layer
 *           at the headline size).
 *   bit 28: ganffn_encoder_bwd_parts_supported() answers 0: weight gradients always go through the reduce launch (same bits).
 * Every combination is parity-tested; results agree to rounding.  The switch must not change between a forward pass and
 * the backward pass that consumes its saved block. */
int ganffn_debug_set_ffn_mode(int bits);

#ifdef __cplusplus
}
#endif
#endif /* GANFFN_H */
