// api.hip — C ABI of libganffn.so (declared in include/ganffn.h): argument checking and the launch
// sequences of the encoder stack, the heads and the plain linears.  No allocation, no synchronisation.
#include "common.h"

#include <string.h>

namespace ganffn {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// from elementwise.hip
int launch_gelu_bwd_drop(const float* d, const float* u, float* out, int R, int C, float p, uint32_t site,
                         const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_disc_tail_fwd(const float* a2, const float* w3, const float* b3, float* prob, int T, int D2, float p,
                         const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_disc_tail_bwd(const float* dprob, const float* prob, const float* a2, const float* u2, const float* w3,
                         float* d_pre2, float* gw3, float* gb3, int T, int D2, float p, const uint64_t* rng, uint64_t add,
                         int train, hipStream_t st, float* gpart);
int launch_disc_tail_reduce(const float* gpart, int nblk_, int D2, float* gw3, float* gb3, hipStream_t st);
// disc_head.hip: the d_model-100 discriminator head (100 -> 64 -> 16 -> 1) as one kernel per direction
bool disc_head_fused_supported(int E, int D1, int D2);
int disc_head_blocks(int T);
int launch_disc_head_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                         const float* b3, float* g0, float* u1, float* a1, float* u2, float* a2, float* prob, float* out, int T,
                         float p, const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_disc_head_bwd(const float* dprob, const float* x, const float* w1, const float* w2, const float* w3, const float* u1,
                         const float* u2, const float* a2, const float* prob, float* dx, float* d_pre1, float* d_pre2, float* gpart,
                         int T, float p, const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int T, int K, int N, hipStream_t st);
int launch_small_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* gw, float* gb, int T, int K,
                            int N, hipStream_t st);

// ------------------------------------------------------------------------------------------
// parameter slab layout of one encoder layer
// ------------------------------------------------------------------------------------------
struct LayerOff {
    int64_t in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b, total;
};
static LayerOff layer_off(int E, int F) {
    LayerOff o;
    int64_t p = 0;
    auto take = [&](int64_t n) { int64_t r = p; p += (n + 3) & ~int64_t(3); return r; };
    o.in_w = take((int64_t)3 * E * E);
    o.in_b = take(3 * E);
    o.out_w = take((int64_t)E * E);
    o.out_b = take(E);
    o.w1 = take((int64_t)F * E);
    o.b1 = take(F);
    o.w2 = take((int64_t)E * F);
    o.b2 = take(E);
    o.n1w = take(E);
    o.n1b = take(E);
    o.n2w = take(E);
    o.n2b = take(E);
    o.total = p;
    return o;
}

// ------------------------------------------------------------------------------------------
// saved-for-backward layout of one encoder stack
//   X[0..L]      (L+1) * T*E     X[0] = PE output, X[l+1] = output of layer l
//   per layer:   qkv 3TE | attn_o TE | x1 TE | xhat1 TE | xhat2 TE | h TF | rstd1 T4 | rstd2 T4 | lse (T H)4 | keep B H 448
// ------------------------------------------------------------------------------------------
struct SavedOff {
    int64_t X, layers, per_layer, qkv, attn_o, x1, xhat1, xhat2, h, rstd1, rstd2, lse, keep, hmask, total;
};
static SavedOff saved_off(const ganffn_enc_cfg* c) {
    SavedOff s;
    const int64_t T = (int64_t)c->S * c->B, TE = T * c->E, TF = T * c->F, T4 = (T + 3) & ~int64_t(3);
    s.X = 0;
    s.layers = (int64_t)(c->L + 1) * TE;
    int64_t p = 0;
    s.qkv = p; p += 3 * TE;
    s.attn_o = p; p += TE;
    s.x1 = p; p += TE;
    s.xhat1 = p; p += TE;
    s.xhat2 = p; p += TE;
    s.h = p; p += TF;
    s.rstd1 = p; p += T4;
    s.rstd2 = p; p += T4;
    s.lse = p; p += (T * c->H + 3) & ~int64_t(3);     // attention log-sum-exp [B*H x S]
    s.keep = p; p += (int64_t)c->B * c->H * ATTN_KEEP_WORDS;   // attention-dropout keep words (uint32) of a train-mode pass
    s.hmask = p; p += (epi_mask_words(T, c->F) / 2 + 3) & ~int64_t(3);   // [h > 0] of the hidden activation, 1 bit each (uint16 words)
    s.per_layer = p;
    s.total = s.layers + (int64_t)c->L * s.per_layer;
    return s;
}

static int check_cfg(const ganffn_enc_cfg* c) {
    GF_CHECK_ARG(c, "null cfg");
    GF_CHECK_ARG(c->S >= 1 && c->S <= 110, "S=%d out of range [1,110] (PositionalEncoding max_len, model.py:1179)", c->S);
    GF_CHECK_ARG(c->B >= 1, "B=%d", c->B);
    GF_CHECK_ARG(c->E >= 4 && (c->E & 3) == 0 && c->E <= 640, "E=%d must be a multiple of 4 and <= 640", c->E);
    GF_CHECK_ARG(c->H >= 1 && c->E % c->H == 0, "E=%d not divisible by H=%d", c->E, c->H);
    GF_CHECK_ARG(c->F >= 4 && (c->F & 3) == 0, "F=%d must be a multiple of 4", c->F);
    GF_CHECK_ARG(c->L >= 1 && c->L <= 64, "L=%d", c->L);
    GF_CHECK_ARG(c->p_pe >= 0.f && c->p_pe < 1.f && c->p_enc >= 0.f && c->p_enc < 1.f, "dropout p out of [0,1)");
    return 0;
}

// A/B switches: one atomic word (common.h `Mode`); what each bit selects and what was measured is in include/ganffn.h and
// DESIGN.md section 6.
std::atomic<uint32_t> g_mode_word{0};
GF_LAB_ONLY(extern unsigned long long* g_n100_stamps; extern unsigned long long* g_wres_stamps;)
constexpr int MAX_SPLITS = 16;  // partial-output slabs: K chunks of gemm_n100 (<= 16) / split-K GEMMs (<= 8)

static int64_t a4(int64_t n) { return (n + 3) & ~int64_t(3); }

// per LayerNorm backward launch: per-block partial sums of the weight / bias gradient
static int64_t ln_part_floats(const ganffn_enc_cfg* c) {
    const int T = c->S * c->B;
    const int nb = ln_bwd_blocks(T) > rc_blocks(T) ? ln_bwd_blocks(T) : rc_blocks(T);
    return (int64_t)nb * 2 * c->E;
}

// K splits of the in-proj dgrad [T x 3E] x [3E x E]: its T/64 x E/64 output tiles are far fewer than the chip's workgroup
// slots (94 at d_model 100, T = 3008), so K is cut into ~128-wide chunks (at most 8) whose partial outputs the consuming
// LayerNorm backward adds in order
static int inproj_dgrad_splits(int T, int E) {
    const long tiles = (long)((T + 63) / 64) * ((E + 63) / 64);
    if (tiles >= 768) return 1;
    long s = (768 + tiles - 1) / tiles;
    const long maxs = (3L * E) / 128;
    if (s > maxs) s = maxs;
    if (s > 8) s = 8;
    return s < 1 ? 1 : (int)s;
}

static int64_t enc_ws_floats(const ganffn_enc_cfg* c) {
    const int64_t T = (int64_t)c->S * c->B, TE = T * c->E, TF = T * c->F;
    // L x (dh | dyA dyB d_qkv(3)) | dz2 dz1 d_attn | tmp slabs | L x 2 LayerNorm partial-sum blocks
    // + the partial-slab workspace of the grouped weight-gradient launch (narrow groups split the token range)
    // + L transposed {in-proj, out-proj} weight blocks (rowchain backward, d_model 100)
    const int64_t rcw = rc_supported(c->E) ? (int64_t)c->L * rc_pack_floats() + 8 : 0;
    const int64_t bwd = (int64_t)c->L * (TF + 5 * TE) + (3 + MAX_SPLITS) * TE + (int64_t)c->L * 2 * ln_part_floats(c) +
                        gemm_tn_grouped_part_floats() + 8 + rcw;
    const SavedOff s = saved_off(c);
    const int64_t fwd_nosave = 2 * TE + s.per_layer + MAX_SPLITS * TE;          // X ping-pong + one layer's saved set + tmp slabs
    return (bwd > fwd_nosave ? bwd : fwd_nosave) + 64;
}

}  // namespace ganffn

using namespace ganffn;

extern "C" int ganffn_version(void) { return GANFFN_VERSION; }
extern "C" const char* ganffn_last_error(void) { return err_buf(); }

extern "C" int64_t ganffn_layer_param_count(int E, int F) { return layer_off(E, F).total; }
extern "C" int ganffn_layer_param_offsets(int E, int F, int64_t* o) {
    GF_CHECK_ARG(o, "null offsets");
    const LayerOff l = layer_off(E, F);
    const int64_t v[12] = {l.in_w, l.in_b, l.out_w, l.out_b, l.w1, l.b1, l.w2, l.b2, l.n1w, l.n1b, l.n2w, l.n2b};
    memcpy(o, v, sizeof(v));
    return 0;
}

extern "C" int64_t ganffn_encoder_saved_floats(const ganffn_enc_cfg* c) {
    if (check_cfg(c) != 0) return -1;
    return saved_off(c).total;
}
extern "C" int64_t ganffn_encoder_saved_hidden_offset(const ganffn_enc_cfg* c, int layer) {
    if (check_cfg(c) != 0 || layer < 0 || layer >= c->L) return -1;
    const SavedOff so = saved_off(c);
    return so.layers + (int64_t)layer * so.per_layer + so.h;
}
extern "C" int64_t ganffn_encoder_workspace_floats(const ganffn_enc_cfg* c) {
    if (check_cfg(c) != 0) return -1;
    return enc_ws_floats(c);
}

// ------------------------------------------------------------------------------------------
// encoder stack forward
// ------------------------------------------------------------------------------------------
extern "C" int ganffn_encoder_fwd(const ganffn_enc_cfg* c, const float* x_in, const float* pe, const float* params,
                                  float* out, float* saved, float* workspace, const uint64_t* rng, uint64_t add,
                                  void* stream) {
    const Mode md = mode();
    GF_TRY(check_cfg(c));
    GF_CHECK_ARG(x_in && pe && params && out && workspace, "encoder_fwd: null pointer");
    GF_CHECK_ARG(aligned16(x_in) && aligned16(params) && aligned16(out) && aligned16(workspace) && (!saved || aligned16(saved)),
                 "encoder_fwd: buffers must be 16-byte aligned");
    const bool drop = c->train && (c->p_pe > 0.f || c->p_enc > 0.f);
    GF_CHECK_ARG(!drop || rng, "encoder_fwd: rng required in train mode");
    hipStream_t st = (hipStream_t)stream;
    const int S = c->S, B = c->B, E = c->E, H = c->H, F = c->F, L = c->L, T = S * B;
    const int64_t TE = (int64_t)T * E;
    const LayerOff lo = layer_off(E, F);
    const SavedOff so = saved_off(c);
    const int train = c->train;

    float* tmp;      // [MAX_SPLITS][T x E] GEMM output (partial slabs) before residual+LN
    float* Xcur;
    if (saved) {
        tmp = workspace;
        Xcur = saved + so.X;
    } else {
        tmp = workspace + 2 * TE + so.per_layer;
        Xcur = workspace;
    }
    // d_model 100: out-proj + residual + dropout + LN1 is one kernel, and LN2 carries the NEXT layer's in-proj (rowchain.hip);
    // the positional encoding + dropout at the head of the stack carries layer 0's
    const bool rc = rc_supported(E) && !md.rc_off();
    auto layer_saved = [&](int l) { return saved ? saved + so.layers + (int64_t)l * so.per_layer : workspace + 2 * TE; };
    if (rc && !md.pe_off()) {
        GF_TRY(launch_rc_pe_inproj_fwd(x_in, pe, Xcur, params + lo.in_w, params + lo.in_b, layer_saved(0) + so.qkv, T, B, c->p_pe, rng,
                                       add, train, st));
    } else {
        GF_TRY(launch_pe_dropout(x_in, pe, Xcur, S, B, E, c->p_pe, rng, add, train, st));
        if (rc) {
            EpiArgs e0;
            e0.bias = params + lo.in_b;
            GF_TRY(launch_gemm_nt(Xcur, E, params + lo.in_w, E, layer_saved(0) + so.qkv, 3 * E, T, 3 * E, E, EPI_NONE, e0, st));
        }
    }

    for (int l = 0; l < L; ++l) {
        const float* P = params + (int64_t)l * lo.total;
        float* sv = layer_saved(l);
        float* Xnext = saved ? saved + so.X + (int64_t)(l + 1) * TE : workspace + ((l & 1) ? 0 : TE);
        if (l == L - 1) Xnext = out;
        const uint32_t site = SITE_LAYER0 + 4 * l;
        EpiArgs ea;
        // qkv = X W_in^T + b_in
        ea.bias = P + lo.in_b;
        if (!rc) GF_TRY(launch_gemm_nt(Xcur, E, P + lo.in_w, E, sv + so.qkv, 3 * E, T, 3 * E, E, EPI_NONE, ea, st));
        // attention core
        // (keep words only when a backward will follow: saved != null)
        GF_TRY(launch_attention_fwd(sv + so.qkv, sv + so.attn_o, sv + so.lse, saved ? reinterpret_cast<uint32_t*>(sv + so.keep) : nullptr, S, B, E,
                                    H, c->p_enc, site + 0, rng, add, train, st));
        // out-proj, residual + dropout + LN1
        if (rc) {
            GF_TRY(launch_rc_outproj_ln_fwd(sv + so.attn_o, P + lo.out_w, P + lo.out_b, Xcur, P + lo.n1w, P + lo.n1b, sv + so.x1,
                                            sv + so.xhat1, sv + so.rstd1, T, c->ln_eps, c->p_enc, site + 1, rng, add, train, st));
        } else {
            ea.bias = P + lo.out_b;
            // (the 512-wide out-proj is 376 output tiles on 256 CUs: K in two halves, summed by the LayerNorm kernel)
            int osplits = md.outproj_nosplit() ? 1 : gemm_splitk_factor(T, E, E);
            GF_TRY(launch_gemm_nt(sv + so.attn_o, E, P + lo.out_w, E, tmp, E, T, E, E, EPI_NONE, ea, st, &osplits, TE));
            GF_TRY(launch_add_drop_ln_fwd(Xcur, tmp, P + lo.n1w, P + lo.n1b, sv + so.x1, sv + so.xhat1, sv + so.rstd1, T, E,
                                          c->ln_eps, c->p_enc, site + 1, rng, add, train, st, osplits, TE));
        }
        // FFN: h = drop(relu(x1 W1^T + b1)); y = h W2^T + b2
        int splits = 1;
        {
            EpiArgs e1;
            e1.bias = P + lo.b1; e1.p = c->p_enc; e1.site = site + 2; e1.rng = rng; e1.rng_add = add; e1.train = train;
            if (saved) e1.mask_out = reinterpret_cast<uint16_t*>(sv + so.hmask);     // the pattern the linear2 dgrad will ask for
            GF_TRY(launch_gemm_nt(sv + so.x1, E, P + lo.w1, E, sv + so.h, F, T, F, E, EPI_RELU_DROP, e1, st));
            if (n100_supported(E, F) && !md.n100_off()) {
                splits = MAX_SPLITS;                   // K chunks = output slabs, summed by the LayerNorm kernel
                GF_TRY(launch_gemm_n100(sv + so.h, F, P + lo.w2, F, 0, P + lo.b2, tmp, TE, T, F, &splits, st));
            } else {
                ea.bias = P + lo.b2;
                splits = gemm_splitk_factor(T, E, F);      // few output tiles, K = 2048: split K, LN sums the slabs
                GF_TRY(launch_gemm_nt(sv + so.h, F, P + lo.w2, F, tmp, E, T, E, F, EPI_NONE, ea, st, &splits, TE));
            }
        }
        if (rc) {
            // LN2, then (all but the last layer) qkv of layer l + 1 from the fresh rows while they are in the workgroup
            const float* Pn = (l + 1 < L) ? P + lo.total : nullptr;
            GF_TRY(launch_rc_ln_inproj_fwd(tmp, splits, TE, sv + so.x1, P + lo.n2w, P + lo.n2b, Xnext, sv + so.xhat2, sv + so.rstd2,
                                           Pn ? Pn + lo.in_w : nullptr, Pn ? Pn + lo.in_b : nullptr,
                                           Pn ? layer_saved(l + 1) + so.qkv : nullptr, T, c->ln_eps, c->p_enc, site + 3, rng, add,
                                           train, st));
        } else {
            GF_TRY(launch_add_drop_ln_fwd(sv + so.x1, tmp, P + lo.n2w, P + lo.n2b, Xnext, sv + so.xhat2, sv + so.rstd2, T, E,
                                          c->ln_eps, c->p_enc, site + 3, rng, add, train, st, splits, TE));
        }
        Xcur = Xnext;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// encoder stack backward over layers [lo_l, hi_l)
// ------------------------------------------------------------------------------------------
extern "C" int ganffn_encoder_bwd2(const ganffn_enc_cfg* c, int layer_lo, int layer_hi, float* dx, const float* params,
                                   float* grads, const float* saved, float* workspace, const uint64_t* rng, uint64_t add,
                                   int need_dx_in, void* stream);
extern "C" int ganffn_encoder_bwd(const ganffn_enc_cfg* c, int layer_lo, int layer_hi, float* dx, const float* params,
                                  float* grads, const float* saved, float* workspace, const uint64_t* rng, uint64_t add,
                                  void* stream) {
    return ganffn_encoder_bwd2(c, layer_lo, layer_hi, dx, params, grads, saved, workspace, rng, add, 1, stream);
}
static int encoder_bwd_impl(const ganffn_enc_cfg* c, int layer_lo, int layer_hi, float* dx, const float* params,
                            float* grads, const float* saved, float* workspace, const uint64_t* rng, uint64_t add,
                            int need_dx_in, void* stream, TnSlabs* slabs);
extern "C" int ganffn_encoder_bwd2(const ganffn_enc_cfg* c, int layer_lo, int layer_hi, float* dx, const float* params,
                                   float* grads, const float* saved, float* workspace, const uint64_t* rng, uint64_t add,
                                   int need_dx_in, void* stream) {
    return encoder_bwd_impl(c, layer_lo, layer_hi, dx, params, grads, saved, workspace, rng, add, need_dx_in, stream, nullptr);
}
// does ganffn_encoder_bwd_parts leave the weight gradients of this stack unreduced?  (d_model 100 on the rowchain / tn100 kernels,
// at most 10 layers: one grouped weight-gradient launch for the whole pass)
static bool bwd_parts_supported(const ganffn_enc_cfg* c, const Mode md) {
    return rc_supported(c->E) && !md.rc_off() && !md.tn100_off() && !md.tn100_in_kernel_sum() && !md.adam_slabs_off() && c->F == 2048 &&
           4 * c->L <= 40;
}
extern "C" int ganffn_encoder_bwd_parts_supported(const ganffn_enc_cfg* c) {
    if (check_cfg(c) != 0) return -1;
    return bwd_parts_supported(c, mode()) ? 1 : 0;
}
extern "C" int64_t ganffn_encoder_bwd_parts_covered(int E, int F) { return layer_off(E, F).n1w; }
extern "C" int ganffn_encoder_bwd_parts(const ganffn_enc_cfg* c, float* dx, const float* params, float* grads, const float* saved,
                                        float* workspace, const uint64_t* rng, uint64_t add, int need_dx_in, int64_t* part_offset,
                                        int64_t* part_stride, int* n_parts, void* stream) {
    GF_TRY(check_cfg(c));
    GF_CHECK_ARG(grads && part_offset && part_stride && n_parts, "encoder_bwd_parts: null pointer");
    GF_CHECK_ARG(bwd_parts_supported(c, mode()), "encoder_bwd_parts: this stack's weight gradients are reduced in the launch (ask ganffn_encoder_bwd_parts_supported)");
    TnSlabs sl{grads, (long)c->L * layer_off(c->E, c->F).total, 1, nullptr, 0};
    GF_TRY(encoder_bwd_impl(c, 0, c->L, dx, params, grads, saved, workspace, rng, add, need_dx_in, stream, &sl));
    *n_parts = sl.n_parts;
    *part_offset = sl.part ? sl.part - workspace : 0;
    *part_stride = sl.part_stride;
    return 0;
}
static int encoder_bwd_impl(const ganffn_enc_cfg* c, int layer_lo, int layer_hi, float* dx, const float* params,
                            float* grads, const float* saved, float* workspace, const uint64_t* rng, uint64_t add,
                            int need_dx_in, void* stream, TnSlabs* slabs) {
    const Mode md = mode();
    GF_TRY(check_cfg(c));
    GF_CHECK_ARG(dx && params && saved && workspace, "encoder_bwd: null pointer");
    GF_CHECK_ARG(0 <= layer_lo && layer_lo < layer_hi && layer_hi <= c->L, "encoder_bwd: bad layer range [%d,%d)", layer_lo, layer_hi);
    GF_CHECK_ARG(aligned16(dx) && aligned16(params) && aligned16(saved) && aligned16(workspace) && (!grads || aligned16(grads)),
                 "encoder_bwd: buffers must be 16-byte aligned");
    const bool drop = c->train && (c->p_pe > 0.f || c->p_enc > 0.f);
    GF_CHECK_ARG(!drop || rng, "encoder_bwd: rng required in train mode");
    hipStream_t st = (hipStream_t)stream;
    const int S = c->S, B = c->B, E = c->E, H = c->H, F = c->F, T = S * B;
    const int64_t TE = (int64_t)T * E, TF = (int64_t)T * F;
    const LayerOff lo = layer_off(E, F);
    const SavedOff so = saved_off(c);
    const int train = c->train;
    const float pdrop = (train ? c->p_enc : 0.f);

    // Weight gradients are DEFERRED: every layer keeps what its 4 wgrad GEMMs read (dh, dyA, dyB, d_qkv) in its own
    // buffer set, and one grouped launch at the end of the range computes all of them (4 x layers problems).  The
    // LayerNorm weight / bias gradients are reduced the same way: per-block partial sums now, one ordered reduce
    // launch for the whole range at the end.  Nothing in here uses atomics: gradients are bit-reproducible.
    const int64_t SET = TF + 5 * TE;              // dh | dyA | dyB | d_qkv(3)
    float* set0 = workspace;
    float* dz2 = set0 + (int64_t)c->L * SET;      // [T x E] LN2 input gradient (residual branch into x1)
    float* dz1 = dz2 + TE;            // [T x E] LN1 input gradient (residual branch into X[l])
    float* d_attn = dz1 + TE;         // [T x E]
    float* tmp = d_attn + TE;         // [MAX_SPLITS][T x E] partial slabs of dh W1
    float* lnp0 = tmp + (int64_t)MAX_SPLITS * TE;   // [L][2] LayerNorm partial-sum blocks
    const int64_t LNP = ln_part_floats(c);
    float* tnp = lnp0 + (int64_t)c->L * 2 * LNP;     // grouped-wgrad partial slabs
    tnp = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(tnp) + 15) & ~(uintptr_t)15);
    const bool rc = rc_supported(E) && !md.rc_off();
    float* rcw = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(tnp + gemm_tn_grouped_part_floats()) + 15) & ~(uintptr_t)15);
    const int64_t RCW = rc_pack_floats();              // per layer: in_w^T [E x 3E] | out_w^T [E x E]
    if (rc) GF_TRY(launch_rc_pack(params + (int64_t)layer_lo * lo.total, lo.total, lo.in_w, lo.out_w, rcw, layer_hi - layer_lo, st));
    const int lnblk = rc ? rc_blocks(T) : ln_bwd_blocks(T);
    TnDesc tn[40];
    int ntn = 0;
    float* r_gw[2 * 64]; float* r_gb[2 * 64]; const float* r_part[2 * 64]; int r_nb[2 * 64];
    int nred = 0;

    const float* dxin = dx;       // gradient wrt the current layer's output: the caller's dx, then in-proj dgrad slabs
    int dxin_slabs = 1;
    for (int l = layer_hi - 1; l >= layer_lo; --l) {
        const float* P = params + (int64_t)l * lo.total;
        float* G = grads ? grads + (int64_t)l * lo.total : nullptr;
        const float* sv = saved + so.layers + (int64_t)l * so.per_layer;
        const float* Xl = saved + so.X + (int64_t)l * TE;
        const uint32_t site = SITE_LAYER0 + 4 * l;
        float* bs = set0 + (int64_t)(l - layer_lo) * SET;
        float* dh = bs;               // [T x F]
        float* dyA = dh + TF;         // [T x E] d(FFN output)
        float* dyB = dyA + TE;        // [T x E] d(attention block output)
        float* d_qkv = dyB + TE;      // [T x 3E]
        float* lnp2 = lnp0 + (int64_t)(2 * (l - layer_lo)) * LNP;
        float* lnp1 = lnp2 + LNP;
        EpiArgs none;
        // LN2 backward: dL/dX[l+1] -> dz2 (to x1), dyA (to FFN output).  dL/dX[l+1] is the caller's dx for the top layer
        // of the range and otherwise the in-proj dgrad of the layer above: split-K partial slabs in `tmp`, summed here
        if (rc) {
            // top layer of the range: the caller's dx; below it: the in-proj dgrad of the layer above (its d_qkv is still in
            // that layer's buffer set, its residual-branch gradient in dz1) runs inside this kernel
            if (l == layer_hi - 1)
                GF_TRY(launch_rc_ln_bwd(nullptr, nullptr, dx, 1, 0, nullptr, sv + so.xhat2, sv + so.rstd2, P + lo.n2w, dz2, dyA,
                                        G ? lnp2 : nullptr, nullptr, nullptr, T, c->p_enc, site + 3, rng, add, train, st));
            else
                GF_TRY(launch_rc_ln_bwd(bs + SET + TF + 2 * TE, rcw + (int64_t)(l + 1 - layer_lo) * RCW, nullptr, 0, 0, dz1,
                                        sv + so.xhat2, sv + so.rstd2, P + lo.n2w, dz2, dyA, G ? lnp2 : nullptr, nullptr, nullptr, T,
                                        c->p_enc, site + 3, rng, add, train, st));
        } else {
            GF_TRY(launch_add_drop_ln_bwd(dxin, sv + so.xhat2, sv + so.rstd2, P + lo.n2w, dz2, dyA, G ? G + lo.n2w : nullptr,
                                          G ? G + lo.n2b : nullptr, T, E, c->p_enc, site + 3, rng, add, train, st, dxin_slabs, TE,
                                          nullptr, G ? lnp2 : nullptr));
        }
        if (G) { r_gw[nred] = G + lo.n2w; r_gb[nred] = G + lo.n2b; r_part[nred] = lnp2; r_nb[nred] = lnblk; ++nred; }
        // linear2 wgrad: gW2[E,F] += dyA^T h ; gb2 += colsum(dyA)
        if (G) tn[ntn++] = TnDesc{dyA, E, sv + so.h, F, G + lo.w2, F, G + lo.b2, E, F, T};
        const float mscale = (pdrop > 0.f) ? 1.0f / (1.0f - pdrop) : 1.0f;
        int splits = 1;
        {
            EpiArgs em;
            em.aux_in = sv + so.h;
            em.mscale = mscale;
            if (!md.mask_float()) em.mask_in = reinterpret_cast<const uint16_t*>(sv + so.hmask);     // linear1's epilogue left the pattern as bits
            GF_TRY(launch_gemm_nn(dyA, E, P + lo.w2, F, dh, F, T, F, E, EPI_MASK_POS, em, st));
        }
        // linear1 wgrad: gW1[F,E] += dh^T x1 ; gb1 += colsum(dh)
        if (G) tn[ntn++] = TnDesc{dh, F, sv + so.x1, E, G + lo.w1, E, G + lo.b1, F, E, T};
        {
            // d x1 = dh W1 (split-K slabs) + dz2, consumed directly by the LN1 backward
            if (n100_supported(E, F) && !md.n100_off()) {
                splits = MAX_SPLITS;
                GF_TRY(launch_gemm_n100(dh, F, P + lo.w1, E, 1, nullptr, tmp, TE, T, F, &splits, st));
            } else {
                splits = gemm_splitk_factor(T, E, F);
                GF_TRY(launch_gemm_nn(dh, F, P + lo.w1, E, tmp, E, T, E, F, EPI_NONE, none, st, &splits, TE));
            }
        }
        if (rc) {
            // LN1 backward + the out-proj dgrad (d_attn = dyB W_o) on the rows while they are in the workgroup
            GF_TRY(launch_rc_ln_bwd(nullptr, nullptr, tmp, splits, TE, dz2, sv + so.xhat1, sv + so.rstd1, P + lo.n1w, dz1, dyB,
                                    G ? lnp1 : nullptr, rcw + (int64_t)(l - layer_lo) * RCW + 3 * (int64_t)E * E, d_attn, T, c->p_enc,
                                    site + 1, rng, add, train, st));
        } else {
            GF_TRY(launch_add_drop_ln_bwd(tmp, sv + so.xhat1, sv + so.rstd1, P + lo.n1w, dz1, dyB, G ? G + lo.n1w : nullptr,
                                          G ? G + lo.n1b : nullptr, T, E, c->p_enc, site + 1, rng, add, train, st, splits, TE, dz2,
                                          G ? lnp1 : nullptr));
        }
        if (G) { r_gw[nred] = G + lo.n1w; r_gb[nred] = G + lo.n1b; r_part[nred] = lnp1; r_nb[nred] = lnblk; ++nred; }
        // out-proj wgrad + dgrad
        if (G) tn[ntn++] = TnDesc{dyB, E, sv + so.attn_o, E, G + lo.out_w, E, G + lo.out_b, E, E, T};
        if (!rc) GF_TRY(launch_gemm_nn(dyB, E, P + lo.out_w, E, d_attn, E, T, E, E, EPI_NONE, none, st));
        // attention core backward
        GF_TRY(launch_attention_bwd(sv + so.qkv, sv + so.attn_o, sv + so.lse, d_attn, reinterpret_cast<const uint32_t*>(sv + so.keep), d_qkv, S, B, E,
                                    H, c->p_enc, site + 0, rng, add, train, st));
        // in-proj wgrad + dgrad; dX[l] = d_qkv W_in + dz1
        if (G) tn[ntn++] = TnDesc{d_qkv, 3 * E, Xl, E, G + lo.in_w, E, G + lo.in_b, 3 * E, E, T};
        if (ntn == 40 || (l == layer_lo && ntn > 0)) {
            GF_TRY(launch_gemm_tn_grouped(tn, ntn, st, tnp, gemm_tn_grouped_part_floats(), slabs));
            ntn = 0;
        }
        EpiArgs eadd;
        eadd.aux_in = dz1;            // dX[l] = d_qkv W_in + dz1 (residual, added by split 0) in the GEMM epilogue
        if (l > layer_lo && rc) {
            // (the in-proj dgrad of this layer runs inside the LN2 backward kernel of layer l - 1)
        } else if (l > layer_lo) {
            // few output tiles (T/64 x E/64): split K into slabs that the next layer's LN2 backward sums on the fly
            // (`tmp` is free here: its previous content, this layer's dh W1 slabs, went into the LN1 backward above)
            int sp = inproj_dgrad_splits(T, E);
            GF_TRY(launch_gemm_nn(d_qkv, 3 * E, P + lo.in_w, E, tmp, E, T, E, 3 * E, EPI_NONE, eadd, st, &sp, TE));
            dxin = tmp;
            dxin_slabs = sp;
        } else if (layer_lo > 0 || need_dx_in) {
            GF_TRY(launch_gemm_nn(d_qkv, 3 * E, P + lo.in_w, E, dx, E, T, E, 3 * E, EPI_NONE, eadd, st));
        }   // (else: the stack's input needs no gradient — autograd would not compute this product either)
    }
    // (unreduced mode: nobody zeroed the gradient slab's encoder region — the LayerNorm parameter gradients are WRITTEN too)
    if (nred > 0) GF_TRY(launch_ln_param_reduce(nred, r_gw, r_gb, r_part, r_nb, E, st, slabs != nullptr));
    if (layer_lo == 0 && need_dx_in) GF_TRY(launch_dropout_bwd_inplace(dx, T, E, c->p_pe, SITE_PE, rng, add, train, st));
    return 0;
}

// ------------------------------------------------------------------------------------------
// heads
// saved layout: gen : t0 [T,E] | u1 [T,D1] | a1 [T,D1] | u2 [T,D2]
//               disc: g0 [T,E] | u1 [T,D1] | a1 [T,D1] | u2 [T,D2] | a2 [T,D2] | prob [T4]
// ------------------------------------------------------------------------------------------
static int check_head(const ganffn_head_cfg* c) {
    GF_CHECK_ARG(c, "null head cfg");
    GF_CHECK_ARG(c->T >= 1 && c->E >= 4 && (c->E & 3) == 0, "head: bad T=%d E=%d", c->T, c->E);
    GF_CHECK_ARG(c->D1 >= 4 && (c->D1 & 3) == 0 && c->D2 >= 4 && (c->D2 & 3) == 0, "head: D1=%d D2=%d must be multiples of 4", c->D1, c->D2);
    GF_CHECK_ARG(c->kind == 0 || c->kind == 1, "head: kind=%d", c->kind);
    GF_CHECK_ARG(c->kind == 0 || c->D2 <= 32, "disc head: D2=%d > 32", c->D2);
    GF_CHECK_ARG(c->p >= 0.f && c->p < 1.f, "head: p out of [0,1)");
    return 0;
}
extern "C" int64_t ganffn_head_saved_floats(const ganffn_head_cfg* c) {
    if (check_head(c) != 0) return -1;
    const int64_t T = c->T, T4 = (T + 3) & ~int64_t(3);
    int64_t n = T * c->E + 2 * T * c->D1 + T * c->D2;
    if (c->kind == 1) n += T * c->D2 + T4;
    return n;
}
// workspace of the head backward: d_pre1 [T x D1] | d_pre2 [T x D2] | split-K partial slabs of the two weight-gradient
// GEMMs | per-block partial sums of the discriminator tail
static int64_t head_part2(const ganffn_head_cfg* c) { return a4(gemm_tn_part_floats(c->D2, c->D1, c->T)); }
static int64_t head_part1(const ganffn_head_cfg* c) { return a4(gemm_tn_part_floats(c->D1, c->E, c->T)); }
extern "C" int64_t ganffn_head_workspace_floats(const ganffn_head_cfg* c) {
    if (check_head(c) != 0) return -1;
    const int64_t T = c->T;
    const int64_t tail_blocks = disc_head_blocks((int)T) > (T + 255) / 256 ? disc_head_blocks((int)T) : (T + 255) / 256;
    return a4(T * c->D1) + a4(T * c->D2) + head_part2(c) + head_part1(c) + tail_blocks * 36 + 64;
}

extern "C" int ganffn_head_fwd(const ganffn_head_cfg* c, const float* x, const float* w1, const float* b1, const float* w2,
                               const float* b2, const float* w3, const float* b3, float* out, float* saved, float* workspace,
                               const uint64_t* rng, uint64_t add, void* stream) {
    GF_TRY(check_head(c));
    const Mode md = mode();
    GF_CHECK_ARG(x && w1 && b1 && w2 && b2 && out && saved, "head_fwd: null pointer");
    GF_CHECK_ARG(c->kind == 0 || (w3 && b3), "head_fwd: disc needs fc3");
    GF_CHECK_ARG(!(c->train && c->p > 0.f) || rng, "head_fwd: rng required in train mode");
    hipStream_t st = (hipStream_t)stream;
    const int T = c->T, E = c->E, D1 = c->D1, D2 = c->D2, train = c->train;
    float* s0 = saved;                      // t0 / g0
    float* u1 = s0 + (int64_t)T * E;
    float* a1 = u1 + (int64_t)T * D1;
    float* u2 = a1 + (int64_t)T * D1;
    EpiArgs e;
    e.p = c->p; e.rng = rng; e.rng_add = add; e.train = train;
    if (c->kind == 0) {
        GF_TRY(launch_gelu_drop_fwd(x, s0, T, E, c->p, SITE_HEAD0, rng, add, train, st));
        e.bias = b1; e.site = SITE_HEAD1; e.aux_out = u1;
        GF_TRY(launch_gemm_nt(s0, E, w1, E, a1, D1, T, D1, E, EPI_DROP_GELU, e, st));
        e.bias = b2; e.site = SITE_HEAD2; e.aux_out = u2;
        GF_TRY(launch_gemm_nt(a1, D1, w2, D1, out, D2, T, D2, D1, EPI_DROP_GELU, e, st));
    } else if (disc_head_fused_supported(E, D1, D2) && !md.dhead_off()) {
        float* a2 = u2 + (int64_t)T * D2;
        float* prob = a2 + (int64_t)T * D2;
        GF_TRY(launch_disc_head_fwd(x, w1, b1, w2, b2, w3, b3, s0, u1, a1, u2, a2, prob, out, T, c->p, rng, add, train, st));
    } else {
        float* a2 = u2 + (int64_t)T * D2;
        float* prob = a2 + (int64_t)T * D2;
        GF_TRY(launch_gelu_drop_fwd(x, s0, T, E, 0.f, SITE_HEAD0, rng, add, 0, st));  // gelu only (model.py:1322)
        e.bias = b1; e.site = SITE_HEAD1; e.aux_out = u1;
        GF_TRY(launch_gemm_nt(s0, E, w1, E, a1, D1, T, D1, E, EPI_DROP_GELU, e, st));
        e.bias = b2; e.site = SITE_HEAD2; e.aux_out = u2;
        GF_TRY(launch_gemm_nt(a1, D1, w2, D1, a2, D2, T, D2, D1, EPI_DROP_GELU, e, st));
        GF_TRY(launch_disc_tail_fwd(a2, w3, b3, prob, T, D2, c->p, rng, add, train, st));
        hipError_t er = hipMemcpyAsync(out, prob, (size_t)T * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (er != hipSuccess) return fail((int)er, "head_fwd: memcpy failed");
    }
    (void)workspace;
    return 0;
}

extern "C" int ganffn_head_bwd(const ganffn_head_cfg* c, const float* d_out, const float* x, const float* w1, const float* w2,
                               const float* w3, float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3,
                               float* dx, const float* saved, float* workspace, const uint64_t* rng, uint64_t add,
                               void* stream) {
    GF_TRY(check_head(c));
    const Mode md = mode();
    GF_CHECK_ARG(d_out && x && w1 && w2 && dx && saved && workspace, "head_bwd: null pointer");
    GF_CHECK_ARG(aligned16(workspace), "head_bwd: workspace must be 16-byte aligned");
    GF_CHECK_ARG(c->kind == 0 || w3, "head_bwd: disc needs fc3");
    GF_CHECK_ARG(!(c->train && c->p > 0.f) || rng, "head_bwd: rng required in train mode");
    hipStream_t st = (hipStream_t)stream;
    const int T = c->T, E = c->E, D1 = c->D1, D2 = c->D2, train = c->train;
    const float* s0 = saved;
    const float* u1 = s0 + (int64_t)T * E;
    const float* a1 = u1 + (int64_t)T * D1;
    const float* u2 = a1 + (int64_t)T * D1;
    float* d_pre1 = workspace;                       // [T x D1]
    float* d_pre2 = d_pre1 + a4((int64_t)T * D1);    // [T x D2]
    float* part2 = d_pre2 + a4((int64_t)T * D2);     // split-K slabs of gw2
    float* part1 = part2 + head_part2(c);            // split-K slabs of gw1
    float* tailp = part1 + head_part1(c);            // discriminator tail: per-block sums
    if (c->kind == 1 && disc_head_fused_supported(E, D1, D2) && !md.dhead_off()) {
        // one kernel: dprob -> d_pre3 -> d_pre2 -> d_pre1 -> dx; the fc1 / fc2 weight gradients (token reductions) stay GEMMs
        const float* a2 = u2 + (int64_t)T * D2;
        const float* prob = a2 + (int64_t)T * D2;
        GF_TRY(launch_disc_head_bwd(d_out, x, w1, w2, w3, u1, u2, a2, prob, dx, gw1 ? d_pre1 : nullptr, gw2 ? d_pre2 : nullptr,
                                    gw3 ? tailp : nullptr, T, c->p, rng, add, train, st));
        if (gw3) GF_TRY(launch_disc_tail_reduce(tailp, disc_head_blocks(T), D2, gw3, gb3, st));
        if (gw2) GF_TRY(launch_gemm_tn_acc(d_pre2, D2, a1, D1, gw2, D1, gb2, D2, D1, T, st, part2, head_part2(c)));
        if (gw1) GF_TRY(launch_gemm_tn_acc(d_pre1, D1, s0, E, gw1, E, gb1, D1, E, T, st, part1, head_part1(c)));
        return 0;
    }
    if (c->kind == 0) {
        // d_pre2 = d_out * gelu'(u2) * m2
        GF_TRY(launch_gelu_bwd_drop(d_out, u2, d_pre2, T, D2, c->p, SITE_HEAD2, rng, add, train, st));
    } else {
        const float* a2 = u2 + (int64_t)T * D2;
        const float* prob = a2 + (int64_t)T * D2;
        GF_TRY(launch_disc_tail_bwd(d_out, prob, a2, u2, w3, d_pre2, gw3, gb3, T, D2, c->p, rng, add, train, st, tailp));
    }
    if (gw2) GF_TRY(launch_gemm_tn_acc(d_pre2, D2, a1, D1, gw2, D1, gb2, D2, D1, T, st, part2, head_part2(c)));
    EpiArgs e;
    e.p = c->p; e.rng = rng; e.rng_add = add; e.train = train;
    e.aux_in = u1; e.site = SITE_HEAD1;
    GF_TRY(launch_gemm_nn(d_pre2, D2, w2, D1, d_pre1, D1, T, D1, D2, EPI_GELU_BWD_DROP, e, st));
    if (gw1) GF_TRY(launch_gemm_tn_acc(d_pre1, D1, s0, E, gw1, E, gb1, D1, E, T, st, part1, head_part1(c)));
    e.aux_in = x;
    if (c->kind == 0) {
        e.site = SITE_HEAD0;
        GF_TRY(launch_gemm_nn(d_pre1, D1, w1, E, dx, E, T, E, D1, EPI_GELU_BWD_DROP, e, st));
    } else {
        GF_TRY(launch_gemm_nn(d_pre1, D1, w1, E, dx, E, T, E, D1, EPI_GELU_BWD, e, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// plain linear
// ------------------------------------------------------------------------------------------
static bool mfma_ok(int K, int N) { return (K & 3) == 0 && (N & 3) == 0; }

extern "C" int ganffn_linear_fwd(const float* x, const float* w, const float* b, float* y, int T, int K, int N, void* stream) {
    GF_CHECK_ARG(x && w && y && T > 0 && K > 0 && N > 0, "linear_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if ((K & 3) == 0 && aligned16(x) && aligned16(w)) {
        EpiArgs e;
        e.bias = b;
        return launch_gemm_nt(x, K, w, K, y, N, T, N, K, EPI_NONE, e, st);
    }
    return launch_small_linear_fwd(x, w, b, y, T, K, N, st);
}

extern "C" int64_t ganffn_linear_bwd_workspace_floats(int T, int K, int N) {
    return mfma_ok(K, N) ? a4(gemm_tn_part_floats(N, K, T)) : 0;
}

extern "C" int ganffn_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* gw, float* gb, int T,
                                 int K, int N, float* workspace, int64_t workspace_floats, void* stream) {
    GF_CHECK_ARG(dy && x && w && T > 0 && K > 0 && N > 0, "linear_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (mfma_ok(K, N) && aligned16(dy) && aligned16(x) && aligned16(w)) {
        EpiArgs e;
        if (dx) GF_TRY(launch_gemm_nn(dy, N, w, K, dx, K, T, K, N, EPI_NONE, e, st));
        // weight gradient: split over tokens into partial slabs when the caller lends a workspace (deterministic
        // ordered reduce), one workgroup per tile over the whole token range otherwise
        const bool ws_ok = workspace != nullptr && aligned16(workspace);
        if (gw) GF_TRY(launch_gemm_tn_acc(dy, N, x, K, gw, K, gb, N, K, T, st, ws_ok ? workspace : nullptr, ws_ok ? (long)workspace_floats : 0));
        return 0;
    }
    return launch_small_linear_bwd(dy, x, w, dx, gw, gb, T, K, N, st);
}

// ------------------------------------------------------------------------------------------
// building blocks for unit tests
// ------------------------------------------------------------------------------------------
extern "C" int ganffn_gemm_nt(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, void* stream) {
    EpiArgs e;
    e.bias = bias;
    return launch_gemm_nt(A, K, W, K, C, N, M, N, K, EPI_NONE, e, (hipStream_t)stream);
}
extern "C" int ganffn_gemm_nn(const float* A, const float* Bm, float* C, int M, int N, int K, void* stream) {
    EpiArgs e;
    return launch_gemm_nn(A, K, Bm, N, C, N, M, N, K, EPI_NONE, e, (hipStream_t)stream);
}
extern "C" int ganffn_gemm_tn_acc(const float* At, const float* Bm, float* C, float* colsum, int M, int N, int K, void* stream) {
    return launch_gemm_tn_acc(At, M, Bm, N, C, N, colsum, M, N, K, (hipStream_t)stream);
}
extern "C" int ganffn_ffn_linear1_fwd(const float* x, const float* w1, const float* b1, float* h, int T, int E, int F, float p,
                                      uint32_t site, const uint64_t* rng, uint64_t add, int train, void* stream) {
    GF_CHECK_ARG(x && w1 && b1 && h && T > 0 && E > 0 && F > 0, "ffn_linear1_fwd: bad arguments");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "ffn_linear1_fwd: rng required when dropout is active");
    EpiArgs e;
    e.bias = b1; e.p = p; e.site = site; e.rng = rng; e.rng_add = add; e.train = train;
    return launch_gemm_nt(x, E, w1, E, h, F, T, F, E, EPI_RELU_DROP, e, (hipStream_t)stream);
}
// test / measurement hook: the generic 64 x 64 GEMM kernel as the encoder stack launches it (bench.py replays the d_model-512
// generator's launch mix through this)
extern "C" int ganffn_gemm_hook(int mode, int epi, const float* A, const float* W, const float* bias, const float* aux, float* C,
                                int64_t slab_stride, int M, int N, int K, float p, uint32_t site, const uint64_t* rng, uint64_t add,
                                int train, int max_slabs, int* n_slabs, void* stream) {
    GF_CHECK_ARG((mode == 0 || mode == 1) && (epi == EPI_NONE || epi == EPI_RELU_DROP || epi == EPI_MASK_POS),
                 "gemm_hook: mode %d / epilogue %d not offered by the hook", mode, epi);
    GF_CHECK_ARG(epi != EPI_MASK_POS || aux, "gemm_hook: the mask epilogue needs the saved activation");
    GF_CHECK_ARG(!(epi == EPI_RELU_DROP && train && p > 0.f) || rng, "gemm_hook: rng required when dropout is active");
    EpiArgs e;
    e.bias = bias; e.aux_in = aux; e.mscale = (train && p > 0.f && epi == EPI_MASK_POS) ? 1.f / (1.f - p) : 1.f;
    e.p = p; e.site = site; e.rng = rng; e.rng_add = add; e.train = train;
    int splits = 1;
    if (max_slabs > 1 && epi == EPI_NONE) {
        splits = gemm_splitk_factor(M, N, K);
        if (splits > max_slabs) splits = max_slabs;
    }
    int* sp = (max_slabs > 1 && epi == EPI_NONE) ? &splits : nullptr;
    if (mode == 0) GF_TRY(launch_gemm_nt(A, K, W, K, C, N, M, N, K, epi, e, (hipStream_t)stream, sp, (long)slab_stride));
    else GF_TRY(launch_gemm_nn(A, K, W, N, C, N, M, N, K, epi, e, (hipStream_t)stream, sp, (long)slab_stride));
    if (n_slabs) *n_slabs = splits;
    return 0;
}
// measurement hook: the K = 100 -> 2048 products of the d_model-100 feed-forward block with the arguments the encoder stack
// gives them (bench.py --replay-family ffn_k100, tools/family_pmc.sh)
extern "C" int ganffn_ffn_k100_hook(int which, const float* a, const float* w, const float* bias, float* out, void* hmask,
                                    const float* h_saved, int T, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train,
                                    void* stream) {
    GF_CHECK_ARG((which == 0 || which == 1) && a && w && out && T > 0, "ffn_k100_hook: bad arguments");
    const int E = 100, F = 2048;
    EpiArgs e;
    if (which == 0) {
        GF_CHECK_ARG(bias, "ffn_k100_hook: linear1 needs its bias");
        GF_CHECK_ARG(!(train && p > 0.f) || rng, "ffn_k100_hook: rng required when dropout is active");
        e.bias = bias; e.p = p; e.site = site; e.rng = rng; e.rng_add = add; e.train = train;
        e.mask_out = reinterpret_cast<uint16_t*>(hmask);
        return launch_gemm_nt(a, E, w, E, out, F, T, F, E, EPI_RELU_DROP, e, (hipStream_t)stream);
    }
    GF_CHECK_ARG(hmask || h_saved, "ffn_k100_hook: the linear2 dgrad needs the pattern bits or the saved activation");
    e.aux_in = h_saved;
    e.mask_in = reinterpret_cast<const uint16_t*>(hmask);
    e.mscale = (train && p > 0.f) ? 1.f / (1.f - p) : 1.f;
    return launch_gemm_nn(a, E, w, F, out, F, T, F, E, EPI_MASK_POS, e, (hipStream_t)stream);
}
extern "C" int ganffn_gemm_n100(const float* A, const float* W, int w_kmajor, const float* bias, float* slabs, int64_t slab_stride,
                                int T, int K, int max_slabs, int* n_slabs, void* stream) {
    GF_CHECK_ARG(n_slabs && max_slabs >= 1 && max_slabs <= MAX_SPLITS, "gemm_n100: max_slabs=%d out of [1,%d]", max_slabs, MAX_SPLITS);
    GF_CHECK_ARG(n100_supported(100, K), "gemm_n100: K=%d must be a multiple of 32, >= 256", K);
    int s = max_slabs;
    GF_TRY(launch_gemm_n100(A, K, W, w_kmajor ? 100 : K, w_kmajor, bias, slabs, (long)slab_stride, T, K, &s, (hipStream_t)stream));
    *n_slabs = s;
    return 0;
}
#ifdef GANFFN_LAB
// lab builds only (make LAB=1 -> lib/libganffn_lab.so; never in the product library, not declared in include/ganffn.h):
// in-kernel time stamps of gemm_n100 (5 x uint64 per workgroup) and gemm_wres (4 x uint64)
extern "C" int ganffn_lab_set_n100_stamps(void* dev_buf) { g_n100_stamps = (unsigned long long*)dev_buf; return 0; }
extern "C" int ganffn_lab_set_wres_stamps(void* dev_buf) { g_wres_stamps = (unsigned long long*)dev_buf; return 0; }
#endif
extern "C" int ganffn_debug_set_ffn_mode(int bits) {
    g_mode_word.store((uint32_t)bits, std::memory_order_relaxed);
    return 0;
}
extern "C" int64_t ganffn_gemm_tn_grouped_workspace_floats(void) { return gemm_tn_grouped_part_floats(); }
extern "C" int ganffn_gemm_tn_grouped(int n, const float* const* At, const float* const* Bm, float* const* C, float* const* colsum,
                                      const int* M, const int* N, const int* K, float* workspace, int64_t workspace_floats,
                                      void* stream) {
    GF_CHECK_ARG(n >= 1 && n <= 40 && At && Bm && C && M && N && K, "gemm_tn_grouped: bad arguments");
    TnDesc d[40];
    for (int i = 0; i < n; ++i) d[i] = TnDesc{At[i], M[i], Bm[i], N[i], C[i], N[i], colsum ? colsum[i] : nullptr, M[i], N[i], K[i]};
    return launch_gemm_tn_grouped(d, n, (hipStream_t)stream, workspace, (long)workspace_floats);
}
extern "C" int ganffn_attention_fwd(const float* qkv, float* o, float* lse, int S, int B, int E, int H, float p, uint32_t site,
                                    const uint64_t* rng, uint64_t add, void* stream) {
    return launch_attention_fwd(qkv, o, lse, nullptr, S, B, E, H, p, site, rng, add, 1, (hipStream_t)stream);
}
extern "C" int64_t ganffn_attention_keep_words(int B, int H) { return (int64_t)B * H * ATTN_KEEP_WORDS; }
extern "C" int ganffn_attention_fwd_keep(const float* qkv, float* o, float* lse, uint32_t* keep, int S, int B, int E, int H, float p,
                                         uint32_t site, const uint64_t* rng, uint64_t add, void* stream) {
    return launch_attention_fwd(qkv, o, lse, keep, S, B, E, H, p, site, rng, add, 1, (hipStream_t)stream);
}
extern "C" int ganffn_attention_bwd_keep(const float* qkv, const float* o, const float* lse, const float* d_o, const uint32_t* keep,
                                         float* d_qkv, int S, int B, int E, int H, float p, uint32_t site, const uint64_t* rng,
                                         uint64_t add, void* stream) {
    return launch_attention_bwd(qkv, o, lse, d_o, keep, d_qkv, S, B, E, H, p, site, rng, add, 1, (hipStream_t)stream);
}
extern "C" int ganffn_attention_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, float* d_qkv, int S,
                                    int B, int E, int H, float p, uint32_t site, const uint64_t* rng, uint64_t add,
                                    void* stream) {
    return launch_attention_bwd(qkv, o, lse, d_o, nullptr, d_qkv, S, B, E, H, p, site, rng, add, 1, (hipStream_t)stream);
}
