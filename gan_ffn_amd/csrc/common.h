// common.h — shared device/host helpers for libganffn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/ganffn.h"

// Lab instrumentation (in-kernel time stamps and their setters) exists only in lab builds (`make LAB=1` ->
// lib/libganffn_lab.so, selected by tools/lab scripts through GANFFN_LIB); the product library compiles none of it.
#ifdef GANFFN_LAB
#define GF_LAB_ONLY(...) __VA_ARGS__
#else
#define GF_LAB_ONLY(...)
#endif

namespace ganffn {

// The library's only mutable process state besides the per-(kernel, device) "LDS opt-in done" masks: ONE word, the A/B mode
// mask of ganffn_debug_set_ffn_mode (include/ganffn.h lists the bits; default 0 = the product path).  It is a relaxed atomic —
// a setter racing with a launch makes that launch take either path, never a torn mixture of switches — and every entry
// point / launcher takes ONE snapshot (`const Mode md = mode();`) and decides from it.
struct Mode {
    uint32_t bits;
    // (bits 0, 7 and 22 selected the fused feed-forward kernels ffn.hip / ffn3.hip until round 5: measured slower in the step three
    //  times over, removed from the product — history + DESIGN.md section 3; the bits are reserved and ignored)
    bool rc_off() const { return bits & 2u; }                       // bit 1: separate GEMM + LayerNorm launches instead of rowchain.hip
    bool n100_off() const { return bits & 4u; }                     // bit 2: generic tiles instead of gemm_n100.hip
    bool tn100_off() const { return bits & 8u; }                    // bit 3: generic tiles instead of gemm_tn100.hip
    bool tn100_in_kernel_sum() const { return bits & 16u; }         // bit 4: last-arriver slab sum (measured slower)
    bool dhead_off() const { return bits & 32u; }                   // bit 5: discriminator head as separate launches
    bool pe_off() const { return bits & 64u; }                      // bit 6: positional encoding and layer 0's in-proj as two launches
    int n100_force_splits() const { return (int)((bits >> 8) & 0xFFu); }     // bits 8..15 (lab): K-chunk count of gemm_n100
    int tn100_force_splits() const { return (int)((bits >> 16) & 0xFu); }    // bits 16..19 (lab): token-chunk count of tn100
    int n100_force_kw() const { const int k = (int)((bits >> 20) & 3u); return k == 3 ? 2 : k; }   // bits 20..21 (lab)
    bool n100_pad7() const { return bits & (1u << 23); }            // bit 23: padded seventh tile instead of the 4x4x1 tail
    bool outproj_nosplit() const { return bits & (1u << 24); }      // bit 24: the wide out-proj unsplit
    bool mask_float() const { return bits & (1u << 25); }           // bit 25: linear2 dgrad reads the saved activation, not the bits
    bool adam_slabs_off() const { return bits & (1u << 28); }       // bit 28: tn100 slabs through the reduce launch (round 4's form)
};
extern std::atomic<uint32_t> g_mode_word;
inline Mode mode() { return Mode{g_mode_word.load(std::memory_order_relaxed)}; }

// ---------------------------------------------------------------------------------------
// error reporting (thread-local message, int return codes; no exceptions across the ABI)
// ---------------------------------------------------------------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);

#define GF_CHECK_ARG(cond, ...)                         \
    do {                                                \
        if (!(cond)) return ::ganffn::fail(-1, __VA_ARGS__); \
    } while (0)

#define GF_LAUNCH_CHECK()                                                              \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess)                                                         \
            return ::ganffn::fail((int)e__, "%s:%d launch failed: %s", __FILE__, __LINE__, \
                                  hipGetErrorString(e__));                             \
    } while (0)

#define GF_HIP(expr)                                                                                \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return ::ganffn::fail((int)e__, "%s:%d %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
    } while (0)

#define GF_TRY(expr)              \
    do {                          \
        int r__ = (expr);         \
        if (r__ != 0) return r__; \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Opt a kernel in to more than 48 KiB of dynamic LDS: once per (kernel, device, host thread), not on every launch.
// KERN is the kernel itself (a non-type template argument), so every instantiation has its own flag word.
template <auto KERN>
static inline int lds_optin(size_t lds, const char* what) {
    if (lds <= 48 * 1024) return 0;
    static thread_local uint64_t done_mask = 0;       // one bit per device ordinal
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return ::ganffn::fail((int)e, "%s: hipGetDevice failed: %s", what, hipGetErrorString(e));
    if (dev < 64 && ((done_mask >> dev) & 1u)) return 0;
    e = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return ::ganffn::fail((int)e, "%s: hipFuncSetAttribute failed: %s", what, hipGetErrorString(e));
    if (dev < 64) done_mask |= uint64_t(1) << dev;
    return 0;
}

// ---------------------------------------------------------------------------------------
// buffer loads / stores: a 128-bit descriptor in scalar registers + a 32-bit byte offset per lane + a scalar byte offset —
// no 64-bit per-lane address arithmetic (fp32 MFMAs share the vector ALU: every VALU instruction next to them costs MFMA time,
// and hipcc's 64-bit address temporaries alias load destinations and bring low s_waitcnt vmcnt to the top of K-loop steps:
// DESIGN.md section 0d rows 2c / 2d).  The hardware range check covers the LANE offset only (not the scalar one); callers
// that pass bytes = 0xFFFFFFFF clamp their offsets into the operand themselves, and the launchers refuse operands >= 4 GiB.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load_f4(__amdgpu_buffer_rsrc_t rs, uint32_t voff_bytes, uint32_t soff_bytes) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff_bytes, (int)soff_bytes, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void buf_store_u32(__amdgpu_buffer_rsrc_t rs, uint32_t voff_bytes, uint32_t soff_bytes, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, rs, (int)voff_bytes, (int)soff_bytes, 0);
}

// ---------------------------------------------------------------------------------------
// dropout sites (mirror of oracle/ganffn_oracle.py SITE_*)
// ---------------------------------------------------------------------------------------
enum : uint32_t {
    SITE_PE = 0,
    SITE_HEAD0 = 1,
    SITE_HEAD1 = 2,
    SITE_HEAD2 = 3,
    SITE_HEAD3 = 4,
    SITE_LAYER0 = 16,  // + 4*layer + {0 attn-prob, 1 post-attn, 2 ffn-mid, 3 post-ffn}
};

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (contract: oracle/philox.py)
// ---------------------------------------------------------------------------------------
struct DropCtx {
    uint32_t k0, k1;   // seed lo/hi
    uint32_t o0, o1;   // offset lo/hi
    uint32_t site;
    uint32_t thr;      // drop iff word < thr
    float scale;       // 1/(1-p)
    int on;
};

__host__ __device__ inline uint32_t drop_threshold(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 4294967296.0;
    if (t >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)t;  // floor
}

__device__ __forceinline__ DropCtx make_drop(const uint64_t* rng, uint64_t add, uint32_t site, float p, int on) {
    DropCtx d;
    d.on = on && (p > 0.f);
    d.site = site;
    d.thr = drop_threshold(p);
    d.scale = d.on ? 1.0f / (1.0f - p) : 1.0f;
    uint64_t seed = 0, off = 0;
    if (d.on) {
        seed = rng[0];
        off = rng[1] + add;
    }
    d.k0 = (uint32_t)seed;
    d.k1 = (uint32_t)(seed >> 32);
    d.o0 = (uint32_t)off;
    d.o1 = (uint32_t)(off >> 32);
    return d;
}

// k0 / k1 (the seed halves) MUST be wave-uniform: they are scalar operands of the round's xor (every caller takes them from
// make_drop, which reads the generator state through scalar loads); the counter words c0..c3 are per lane.
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                        uint32_t k1, uint32_t (&out)[4]) {
    // one v_mad_u64_u32 per 32x32->64 product (hipcc would emit v_mul_hi_u32 + v_mul_lo_u32, both quarter-rate)
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0, p1, cy0, cy1;
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p0), "=s"(cy0) : "s"(M0), "v"(c0));
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p1), "=s"(cy1) : "s"(M1), "v"(c2));
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // hi ^ c ^ k in ONE instruction: gfx950's v_bitop3_b32 with the truth table of a 3-input xor (0x96); hipcc emits two
        // v_xor_b32 (gfx9 has no v_xor3) — 40 instead of 20 per call, and every one of them is paid in MFMA time in the GEMM
        // epilogues.  The round key is wave-uniform (make_drop reads it through scalar loads): the one scalar operand allowed.
        uint32_t n0, n2;
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"(hi1), "v"(c1), "s"(k0));
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"(hi0), "v"(c3), "s"(k1));
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep-multipliers (0 or 1/(1-p)) for the 4 consecutive rows 4*rg .. 4*rg+3 of column c in a [R x C] tensor
__device__ __forceinline__ void drop_mult4(const DropCtx& d, uint32_t rowgroup, uint32_t C, uint32_t c, float (&m)[4]) {
    if (!d.on) {
        m[0] = m[1] = m[2] = m[3] = 1.f;
        return;
    }
    uint32_t w[4];
    philox4(rowgroup * C + c, d.site, d.o0, d.o1, d.k0, d.k1, w);
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = (w[i] >= d.thr) ? d.scale : 0.f;
}

// single element (row r, col c) — used by row-wise kernels; 4x the Philox work of drop_mult4
__device__ __forceinline__ float drop_mult1(const DropCtx& d, uint32_t r, uint32_t C, uint32_t c) {
    if (!d.on) return 1.f;
    uint32_t w[4];
    philox4((r >> 2) * C + c, d.site, d.o0, d.o1, d.k0, d.k1, w);
    const uint32_t x = (r & 3) == 0 ? w[0] : (r & 3) == 1 ? w[1] : (r & 3) == 2 ? w[2] : w[3];
    return (x >= d.thr) ? d.scale : 0.f;
}

// ---------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) {  // nn.GELU() exact (erf) form
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------
// internal launchers shared between translation units
// ---------------------------------------------------------------------------------------
enum Epilogue : int {
    EPI_NONE = 0,          // C = acc (+bias)
    EPI_RELU_DROP = 1,     // C = drop(relu(acc+bias))                       (FFN linear1)
    EPI_DROP_GELU = 2,     // aux = drop(acc+bias); C = gelu(aux)            (head fc1/fc2)
    EPI_MASK_POS = 3,      // C = acc * (aux > 0 ? mscale : 0)               (FFN dgrad through relu+dropout)
    EPI_GELU_BWD_DROP = 4, // C = drop_bwd(acc) * gelu'(aux)  i.e. d(pre-dropout) of u=drop(v), a=gelu(u) chain
    EPI_GELU_BWD = 5,      // C = acc * gelu'(aux)
    EPI_GELU_BWD_DROP0 = 6 // C = acc * dropmult * gelu'(aux): aux = pre-gelu x, out=drop(gelu(x)) (gen head site 0)
};

struct EpiArgs {
    const float* bias = nullptr;  // [N]
    float* aux_out = nullptr;     // [M x N] (EPI_DROP_GELU)
    const float* aux_in = nullptr;// [M x N]
    float mscale = 1.f;           // EPI_MASK_POS
    // 1-bit form of the ReLU / dropout pattern of a [M x N] activation: EPI_RELU_DROP writes it (when non-null) beside the
    // activation, EPI_MASK_POS reads it INSTEAD of aux_in (when non-null).  Word ((m / 32) * N + n) * 2 + h, h = (m / 4) & 1,
    // holds rows 32 (m / 32) + 4 h + (i & 3) + 8 (i >> 2) of column n in bit i — one lane's 16 accumulator registers of a
    // 32 x 32 MFMA tile; epi_mask_words(M, N) uint16 words.
    uint16_t* mask_out = nullptr;
    const uint16_t* mask_in = nullptr;
    float p = 0.f;                // dropout prob
    uint32_t site = 0;
    const uint64_t* rng = nullptr;
    uint64_t rng_add = 0;
    int train = 0;
};

inline long epi_mask_words(long M, long N) { return ((M + 31) / 32) * N * 2; }

// C[MxN] = A[MxK] * W[NxK]^T (NT)
int launch_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                   int epi, const EpiArgs& ea, hipStream_t st, int* splits_io = nullptr, long slab_stride = 0);
int gemm_splitk_factor(int M, int N, int K);
// C[MxN] = A[MxK] * B[KxN] (NN)
int launch_gemm_nn(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K,
                   int epi, const EpiArgs& ea, hipStream_t st, int* splits_io = nullptr, long slab_stride = 0);
// C[MxN] += At[KxM]^T * B[KxN]  (TN); colsum[M] += sum_k At[k][m].  Deterministic: with a partial-slab workspace
// (gemm_tn_part_floats floats) the token range is split over workgroups and reduced in split order, without one a
// single workgroup per output tile runs the whole range.  No atomics.
long gemm_tn_part_floats(int M, int N, int K);
int launch_gemm_tn_acc(const float* At, int lda, const float* Bm, int ldb, float* C, int ldc, float* colsum,
                       int M, int N, int K, hipStream_t st, float* part_ws = nullptr, long part_floats = 0);

// skinny products (dialogue_rnn.hip): M <= 32 rows of A (dialogues) against a weight matrix, up to 8 problems per launch — one
// link of a recurrence's launch chain (DialogueRNN cells, lstm.hip's recurrent products)
struct SkinnyProb {
    const float* A; int lda;      // [M x K]
    const float* W; int ldw;      // NT: [N x K];  NN: [K x N]
    const float* Cin; int ldcin;  // optional addend [M x N]
    const float* Cin2;            // optional second addend [M x N], leading dimension ldc (may be C itself: in-place +=)
    const float* bias;            // optional [N]
    float* C; int ldc;            // [M x N]
    int M, N, K;
};
struct SkinnyGroup {
    SkinnyProb p[8];
};
// nn = false: C = A W^T (+ Cin + Cin2 + bias), W [N x K];  nn = true: C = A W (+ Cin + Cin2), W [K x N]
int launch_skinny(const SkinnyGroup& grp, int nprob, bool nn, hipStream_t st);

// grouped wgrad: n independent TN problems in one launch
struct TnDesc {
    const float* At; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    float* colsum;
    int M, N, K;
};
// part_ws (optional, gemm_tn_grouped_part_floats() floats): when the group has too few output tiles to load every CU
// evenly, the token range is split over 2..8 workgroups per tile; each writes its partial tile to the workspace and one
// grouped reduce launch adds the partials in split order (deterministic).  Without a workspace: one owner per tile.
// Unreduced weight gradients of a d_model-100 pass (round 5; single-GPU step): in  — grad_base / range_floats: the gradient
// slab region every problem's C and colsum lie in (densely, ldc == N); out — n_parts token chunks were computed: chunk 0 WROTE
// (not added) its partial gradients into the slab region itself, chunk z >= 1 into part + (z - 1) * part_stride, shaped like the
// region.  ganffn_adam_step_parts adds them in chunk order: the sum the reduce launch would have formed, bit for bit.
struct TnSlabs {
    float* grad_base; long range_floats;
    int n_parts; float* part; long part_stride;
};
int launch_gemm_tn_grouped(const TnDesc* d, int n, hipStream_t st, float* part_ws = nullptr, long part_floats = 0, TnSlabs* slabs = nullptr);
// gemm_tn100.hip: the same for groups whose every problem has a 100-wide dimension (d_model-100 encoder passes), on
// 112-wide 16x16x4 tiles; needs the partial-slab workspace
bool tn100_supported(const TnDesc* d, int n);
int launch_gemm_tn100_grouped(const TnDesc* d, int n, hipStream_t st, float* part_ws, long part_floats, TnSlabs* slabs = nullptr);
long gemm_tn_grouped_part_floats();

// lse [B*H x S]: log-sum-exp of every score row, written by the forward (may be NULL: not kept) and, together with the
// forward's output o, read by the backward of the small-head kernels (attention16.hip); the head_dim 60/64 kernels
// (attention.hip) recompute the softmax statistics and ignore both.
// keep [B*H x ATTN_KEEP_WORDS] (or NULL): the dropout keep bits of the probability matrix, written by a train-mode
// small-head forward and read back by its backward instead of recomputing the Philox calls (attention16.hip; the
// head_dim 60/64 kernels ignore it).  Same bits either way.
constexpr int ATTN_KEEP_WORDS = 28 * 16;
int launch_attention_fwd(const float* qkv, float* o, float* lse, uint32_t* keep, int S, int B, int E, int H, float p, uint32_t site,
                         const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_attention_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, const uint32_t* keep, float* d_qkv,
                         int S, int B, int E, int H, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train,
                         hipStream_t st);
bool attn16_supported(int E, int H, int S);
int launch_attn16_fwd(const float* qkv, float* o, float* lse, uint32_t* keepw, int S, int B, int E, int H, float p, uint32_t site,
                      const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_attn16_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, const uint32_t* keepw, float* d_qkv,
                      int S, int B, int E, int H, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train,
                      hipStream_t st);

int launch_pe_dropout(const float* x, const float* pe, float* out, int S, int B, int E, float p,
                      const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_dropout_bwd_inplace(float* dx, int R, int C, float p, uint32_t site, const uint64_t* rng, uint64_t add,
                               int train, hipStream_t st);
int launch_dropout(const float* x, float* out, int R, int C, float p, uint32_t site, const uint64_t* rng,
                   uint64_t add, int train, hipStream_t st);
// y / d_out may be given as `nslab` partial slabs `slab_stride` floats apart (split-K GEMM output): they are summed
// on the fly; d_out additionally takes an optional addend (the residual-branch gradient).
int launch_add_drop_ln_fwd(const float* x, const float* y, const float* w, const float* b, float* out, float* xhat,
                           float* rstd, int T, int E, float eps, float p, uint32_t site, const uint64_t* rng,
                           uint64_t add, int train, hipStream_t st, int nslab = 1, long slab_stride = 0);
// gw / gb: per-block partial sums go to gpart (ln_bwd_blocks(T) * 2 * E floats), to be added by launch_ln_param_reduce;
// gpart == NULL runs one workgroup that adds its sums directly.  No atomics either way.
int ln_bwd_blocks(int T);
int launch_add_drop_ln_bwd(const float* d_out, const float* xhat, const float* rstd, const float* w, float* dz,
                           float* dy, float* gw, float* gb, int T, int E, float p, uint32_t site,
                           const uint64_t* rng, uint64_t add, int train, hipStream_t st, int nslab = 1,
                           long slab_stride = 0, const float* addend = nullptr, float* gpart = nullptr);
// overwrite: gw / gb = the sums (the caller did not zero them) instead of +=
int launch_ln_param_reduce(int n, float* const* gw, float* const* gb, const float* const* part, const int* nblk, int E,
                           hipStream_t st, bool overwrite = false);
int launch_add_inplace(float* a, const float* b, int64_t n, hipStream_t st);

// gemm_n100.hip — [T x K] x [K x 100] with a long K on 16x16x4 MFMAs (112-wide feature tile), K cut into output slabs
bool n100_supported(int N, int K);
int n100_splits(int T, int K, int max_splits, int w_kmajor);
int launch_gemm_n100(const float* A, int lda, const float* W, int ldw, int w_kmajor, const float* bias, float* C, long slab_stride,
                     int T, int K, int* splits_io, hipStream_t st);

// rowchain.hip — d_model 100: out-proj + residual + dropout + LayerNorm1, LayerNorm2 + the next layer's in-proj, and the
// mirror-image backward chains, one kernel each (16 token rows per workgroup)
bool rc_supported(int E);
long rc_pack_floats();          // floats of one layer's transposed {in-proj, out-proj} weights (backward)
int rc_blocks(int T);           // workgroups = partial rows of the LayerNorm parameter gradients per launch
int launch_rc_pack(const float* params, long layer_stride, long off_in, long off_out, float* wt, int nl, hipStream_t st);
int launch_rc_outproj_ln_fwd(const float* attn_o, const float* wo, const float* bo, const float* x, const float* gamma,
                             const float* beta, float* out, float* xhat, float* rstd, int T, float eps, float p, uint32_t site,
                             const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_rc_pe_inproj_fwd(const float* x_in, const float* pe, float* out, const float* w_in, const float* b_in, float* qkv, int T,
                            int B, float p, const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_rc_ln_inproj_fwd(const float* y, int nslab, long slab_stride, const float* x, const float* gamma, const float* beta,
                            float* out, float* xhat, float* rstd, const float* w_in, const float* b_in, float* qkv, int T,
                            float eps, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train, hipStream_t st);
int launch_rc_ln_bwd(const float* d_qkv, const float* w_in_t, const float* d_out, int nslab, long slab_stride, const float* addend,
                     const float* xhat, const float* rstd, const float* gamma, float* dz, float* dy, float* gpart,
                     const float* wo_t, float* d_attn, int T, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train,
                     hipStream_t st);
int launch_gelu_drop_fwd(const float* x, float* out, int R, int C, float p, uint32_t site, const uint64_t* rng,
                         uint64_t add, int train, hipStream_t st);

}  // namespace ganffn
