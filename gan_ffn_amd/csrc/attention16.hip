// attention16.hip — self-attention core for SMALL heads (head_dim <= 32: the d_model = 100 / 10-head networks, i.e. five of
// the six networks of the GAN step, and the 300-d MELD-dimension stack), forward and backward, on
// v_mfma_f32_16x16x4_f32 (exact fp32).
//
// Replaces the attention core of torch's nn.MultiheadAttention inside nn.TransformerEncoderLayer
// (/root/reference/model.py:1210,1276,1307,1340,1377): P = softmax(q k^T / sqrt(hd)) over keys, NO masks, dropout(0.1)
// on P in train mode, O = P v.
//
// Why a second kernel family (attention.hip keeps head_dim 60/64): with hd = 10 a 32x32x2 MFMA product that has the head
// dimension on its M axis uses 10 of 32 output rows, and one wave per 32 queries leaves one wave per SIMD with every
// latency (LDS, Philox, exp) exposed.  Here a wave owns 16 queries (forward) or 16 keys (backward): 16x16x4 tiles waste
// 10/16 instead of 10/32, a (dialogue, head) problem is 6-7 waves instead of 3, so a CU holds 2-3 waves per SIMD and one
// wave's VALU work (softmax, Philox) runs under another's MFMAs.
//
// Forward (query on the lane): S^T = K Q^T leaves a lane's query with all its keys in that lane's registers (softmax
// max/sum = in-lane + two cross-group shuffles), and the probability tile is directly the B operand of O^T = V^T P^T
// (an accumulator tile feeds the next MFMA; k index = key, permuted consistently on the LDS side).  The forward also
// writes the log-sum-exp of every (dialogue, head, query) row.
// Backward (KEY on the lane): S = Q K^T and dP = dO V^T have the key on the lane and 4 consecutive queries in the 4
// accumulator registers, so
//   * P = exp(S - LSE) needs no reduction (LSE saved by the forward; D_i = sum_d dO_id O_id from the saved output);
//   * the Philox contract (one call = 4 consecutive queries at one key) is exactly one call per accumulator tile per
//     lane — no mask exchange between lanes;
//   * the tiles are directly the B operands of dV^T = dO^T P~ and dK^T = Q^T dS (a wave owns its 16 keys' dK, dV:
//     no reduction across waves, no atomics);  only dS crosses LDS once, for dQ = dS K.
// Layout: qkv [T x 3E] packed q|k|v per token (t = s*B + b), head h = columns h*hd .. h*hd+hd-1; lse [B*H x S].
#include "common.h"

namespace ganffn {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

template <int HD>
struct A16 {
    static constexpr int KS = (HD + 3) / 4;                 // k-steps of 4 over the head dimension
    static constexpr int NTD = (HD + 15) / 16;              // 16-row output tiles over the head dimension
    // row stride of the [*, hd] LDS matrices: >= 4 KS, and = 4 (mod 8) so that rows 4 apart sit 16 banks apart
    // (the "4 consecutive keys x 16 head-dim columns" operand reads of both half-waves are then conflict-free)
    static constexpr int LD = (KS & 1) ? 4 * KS : 4 * KS + 4;
    static constexpr int TAIL = 16 * NTD;                   // floats readable past the last row (discarded outputs)
};
static_assert(A16<10>::LD == 12 && A16<30>::LD == 36 && A16<16>::LD == 20 && A16<12>::LD == 12, "LD rule");

struct HeadSrc16 {
    float* dst;
    const float* src;
    int ld_src;
    float scale;
};

// Stage NM [S x HD] head slices into LDS images [16 NT rows][LD] (rows >= S and columns >= HD zero), in two phases so that
// the caller can put data-independent work (the Philox calls) between issuing the global loads and waiting for them.
// load(): every global load of all matrices, UNCONDITIONAL, from clamped addresses, nothing else — no use of a loaded
// value before store().  (Round 4: the first version wrote `ok ? q * scale : 0` in load(); hipcc turned each such select
// into a load under an exec-mask branch and, where the scale multiply followed, put an `s_waitcnt vmcnt(0)` right behind
// it: the forward of a d_model-100 pass paid FIVE dependent global round trips before its first LDS write, the backward
// three.  Out-of-range elements are now zeroed by a 0 / scale FACTOR at store time — a multiply cannot be folded into
// "do not load" — and a sched_barrier keeps the loads together.)
template <int HD, int NT, int NM, int NW = NT>   // NW: waves of the workgroup doing the staging
struct HeadStage {
    static constexpr int LD = A16<HD>::LD, PR = LD / 2, ROWS = 16 * NT, PER = ROWS * PR, NTH = 64 * NW;
    static constexpr int U = (PER + NTH - 1) / NTH;
    float2 v[NM][U];
    __device__ __forceinline__ void load(const HeadSrc16 (&m)[NM], int S, int B, int b, int tid) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(tid + u * NTH, PER - 1);
            const int s = i / PR, j = i - s * PR;
            const size_t row = (size_t)(min(s, S - 1) * B + b);
            const int col = min(2 * j, HD - 2);
#pragma unroll
            for (int mi = 0; mi < NM; ++mi) v[mi][u] = *reinterpret_cast<const float2*>(m[mi].src + row * m[mi].ld_src + col);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void store(const HeadSrc16 (&m)[NM], int S, int tid) const {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = tid + u * NTH;
            if (PER % NTH == 0 || i < PER) {
                const int s = i / PR, j = i - s * PR;
                const float in = (s < S && 2 * j < HD) ? 1.f : 0.f;
#pragma unroll
                for (int mi = 0; mi < NM; ++mi) {
                    const float f = in * m[mi].scale;
                    *reinterpret_cast<float2*>(m[mi].dst + s * LD + 2 * j) = make_float2(v[mi][u].x * f, v[mi][u].y * f);
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < NM; ++mi)
            if (tid < A16<HD>::TAIL) m[mi].dst[ROWS * LD + tid] = 0.f;
    }
};

// acc[t] (+)= X_t Y^T over the head dimension: A = rows 16t + c of Xm (one tile per t), B = row `yrow` of Ym for this lane.
// D[m][n]: m = row of X inside tile t (4g + reg), n = this lane's Y row.
template <int HD, int NT>
__device__ __forceinline__ void dot_tiles(floatx4 (&acc)[NT], const float* __restrict__ Xm, const float* __restrict__ yb,
                                          int c, int g) {
    constexpr int LD = A16<HD>::LD, KS = A16<HD>::KS;
    constexpr int BT = KS <= 4 ? KS : (KS % 4 == 0 ? 4 : (KS % 3 == 0 ? 3 : (KS % 5 == 0 ? 5 : 1)));
    const float* pa = Xm + c * LD + g;
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += BT) {
        float av[BT][NT], bv[BT];
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            bv[j] = yb[4 * (k0 + j)];
#pragma unroll
            for (int t = 0; t < NT; ++t) av[j][t] = pa[t * 16 * LD + 4 * (k0 + j)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < BT; ++j)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][t], bv[j], acc[t], 0, 0, 0);
    }
}

// out[dt] (rows d = 16 dt + 4g + reg, column = this lane's n) += sum over rows x = 16t + 4g' + reg of Mm[x][d] * P[t][reg]
// (P = accumulator tiles used as the B operand; k index = g').  Two interleaved accumulator chains per output tile.
template <int HD, int NT>
__device__ __forceinline__ void apply_tiles(floatx4 (&out)[A16<HD>::NTD], const floatx4 (&P)[NT], const float* __restrict__ Mm,
                                            int c, int g) {
    constexpr int LD = A16<HD>::LD, NTD = A16<HD>::NTD;
    floatx4 alt[NTD];
#pragma unroll
    for (int dt = 0; dt < NTD; ++dt) alt[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float* pa = Mm + 4 * g * LD + c;
    float av[2][4][NTD];
    auto fetch = [&](int t, float (&dst)[4][NTD]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < NTD; ++dt) dst[r][dt] = pa[(16 * t + r) * LD + 16 * dt];
    };
    fetch(0, av[0]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) fetch(t + 1, av[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < NTD; ++dt) {
                if (r & 1) alt[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t & 1][r][dt], P[t][r], alt[dt], 0, 0, 0);
                else out[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t & 1][r][dt], P[t][r], out[dt], 0, 0, 0);
            }
    }
#pragma unroll
    for (int dt = 0; dt < NTD; ++dt) out[dt] += alt[dt];
}

// value of lane r of this lane's quad (DPP quad_perm broadcast)
template <int R>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, R * 0x55, 0xF, 0xF, true);
}

// store 4 consecutive head-dim values d0 .. d0+3 (this lane's accumulator registers) of one token row
template <int HD>
__device__ __forceinline__ void store4(float* __restrict__ row, int d0, const floatx4& v, float mul) {
    if (d0 + 1 < HD) *reinterpret_cast<float2*>(row + d0) = make_float2(v[0] * mul, v[1] * mul);
    if (d0 + 3 < HD) *reinterpret_cast<float2*>(row + d0 + 2) = make_float2(v[2] * mul, v[3] * mul);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// forward: one workgroup per (dialogue, head, block of WPB query tiles); a wave owns 16 queries.  WPB < NT cuts a
// (dialogue, head) problem into NT / WPB workgroups that each stage the head's K and V again (a few KB out of L2) — the
// 320 problems of a d_model-100 pass are 1.25 per CU as whole workgroups (64 CUs hold two), finer workgroups spread evenly
// ------------------------------------------------------------------------------------------
template <int HD, int NT, int WPB>
__global__ __launch_bounds__(64 * WPB) void attn16_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                             float* __restrict__ lse, uint32_t* __restrict__ keepw, int S, int B,
                                                             int E, int H, float p, uint32_t site,
                                                             const uint64_t* __restrict__ rng, uint64_t add, int train) {
    constexpr int LD = A16<HD>::LD, NTD = A16<HD>::NTD, MAT = 16 * NT * LD + A16<HD>::TAIL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(NT % WPB == 0, "query tiles per workgroup must divide the tile count");
    constexpr int NQB = NT / WPB;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x / NQB, w = (blockIdx.x - bh * NQB) * WPB + (tid >> 6);     // w: query tile of this wave
    const int b = bh / H, head = bh - b * H;
    float* Qs = smem;
    float* Ks = Qs + MAT;
    float* Vs = Ks + MAT;
    const int ld3 = 3 * E;
    const DropCtx dc = make_drop(rng, add, site, p, train);
    uint32_t mine = 0;
    {
        const HeadSrc16 m3[3] = {{Qs, qkv + head * HD, ld3, rsqrtf((float)HD)}, {Ks, qkv + E + head * HD, ld3, 1.f},
                                 {Vs, qkv + 2 * E + head * HD, ld3, 1.f}};
        HeadStage<HD, NT, 3, WPB> stg;
        stg.load(m3, S, B, b, tid);
        if (dc.on) {
            // Dropout keep-bits, computed while the global loads are in flight (they depend on indices only).  One Philox
            // call = 4 consecutive queries (the 4 lanes of a quad) at one key.  Lane ql of a quad evaluates the calls of
            // register r == ql (keys 16t + 4g + ql); bit 4t + qq of `mine` = keep(query 4Q + qq, that key).
            const int ql = c & 3;
            const uint32_t rowgroup = (uint32_t)(bh * 28 + 4 * w + (c >> 2));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint32_t wd[4];
                philox4(rowgroup * 128u + (uint32_t)(16 * t + 4 * g + ql), dc.site, dc.o0, dc.o1, dc.k0, dc.k1, wd);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) mine |= (wd[qq] >= dc.thr ? 1u : 0u) << (4 * t + qq);
            }
            // keep words for the backward (A16_KEEP_WORDS per (dialogue, head)): word [query group Q = 4w + c/4][4g + ql],
            // nibble t = the four queries of the group at key 16t + 4g + ql — one coalesced dword store per lane; the
            // backward then needs no Philox call at all (its lane (key 16w' + c', group g') reads word [4t + g'][c'])
            if (keepw != nullptr) keepw[(size_t)rowgroup * 16 + 4 * g + ql] = mine;
        }
        stg.store(m3, S, tid);
    }
    __syncthreads();

    // S^T tiles: pr[t][reg] = score(key 16t + 4g + reg, query 16w + c)
    floatx4 pr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pr[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    dot_tiles<HD, NT>(pr, Ks, Qs + (16 * w + c) * LD + g, c, g);

    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (16 * t + 4 * g + r >= S) pr[t][r] = -INFINITY;
            m = fmaxf(m, pr[t][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __expf(pr[t][r] - m);     // exp(-inf) = 0 for padded keys
            pr[t][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    const int qi = 16 * w + c;
    if (lse != nullptr && g == 0 && qi < S) lse[(size_t)bh * S + qi] = m + __logf(sum);

    if (dc.on) {
        const int ql = c & 3;
        const uint32_t mq[4] = {quad_bcast<0>(mine), quad_bcast<1>(mine), quad_bcast<2>(mine), quad_bcast<3>(mine)};
        const float ps = inv * dc.scale;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[t][r] = ((mq[r] >> (4 * t + ql)) & 1u) ? pr[t][r] * ps : 0.f;
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[t][r] *= inv;
    }

    // O^T[d][query] = sum_key V[key][d] P~^T[key][query]
    floatx4 oacc[NTD];
#pragma unroll
    for (int dt = 0; dt < NTD; ++dt) oacc[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    apply_tiles<HD, NT>(oacc, pr, Vs, c, g);
    if (qi < S) {
        float* orow = o + (size_t)(qi * B + b) * E + head * HD;
#pragma unroll
        for (int dt = 0; dt < NTD; ++dt) store4<HD>(orow, 16 * dt + 4 * g, oacc[dt], 1.f);
    }
}

// ------------------------------------------------------------------------------------------
// backward: wave w owns keys 16w .. 16w+15 (dK, dV) and, after the dS hand-over, queries 16w .. 16w+15 (dQ)
// ------------------------------------------------------------------------------------------
// SAVED: the dropout keep bits come from the words the forward stored (keepw) instead of Philox calls — the same bits, so
// both forms give identical results (tests/test_hip_ops.py::test_attention_fwd_bwd compares them with torch.equal)
template <int HD, int NT, bool SAVED>
__global__ __launch_bounds__(64 * NT) void attn16_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                             const float* __restrict__ lse, const float* __restrict__ d_o,
                                                             const uint32_t* __restrict__ keepw, float* __restrict__ d_qkv,
                                                             int S, int B, int E, int H, float p, uint32_t site,
                                                             const uint64_t* __restrict__ rng, uint64_t add, int train) {
    constexpr int LD = A16<HD>::LD, NTD = A16<HD>::NTD, KS = A16<HD>::KS, ROWS = 16 * NT, MAT = ROWS * LD + A16<HD>::TAIL;
    constexpr int LDS_S = ROWS + 4;                      // dS image [query][key]: rows 4 apart sit 16 banks apart
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.x, b = bh / H, head = bh - b * H;
    // K, LSE and D live for the whole kernel; scaled q, V and dO are dead once dV / dK are accumulated, and the dS image
    // takes their place (one more barrier) — at S = 94 that is 43 KB instead of 57 KB a workgroup: three per CU, not two
    float* Ks = smem;
    float* Ls = Ks + MAT;        // [ROWS] log-sum-exp per query (+big for padded rows)
    float* Ds = Ls + ROWS;       // [ROWS] D_i = sum_d dO_id O_id
    uint32_t* Wk = reinterpret_cast<uint32_t*>(Ds + ROWS);    // [NT][64] keep words (saved-mask form)
    float* Qs = Ds + ROWS + 64 * NT;       // scaled q
    float* Vs = Qs + MAT;
    float* Os = Vs + MAT;        // dO
    float* SS = Qs;              // [ROWS][LDS_S] dS, over q / V / dO
    const int ld3 = 3 * E;
    const float scale = rsqrtf((float)HD);
    const DropCtx dc = make_drop(rng, add, site, p, train);
    uint32_t wdt[NT][4];                 // Philox words of this lane's (key, 4-query group) pairs (saved masks: all-ones / zero)
    uint32_t kw1 = 0;                    // saved masks: keep word [4w + g][c] — wave w fetches word row w for the whole workgroup
#pragma unroll
    for (int t = 0; t < NT; ++t) wdt[t][0] = wdt[t][1] = wdt[t][2] = wdt[t][3] = 0xFFFFFFFFu;
    {
        const HeadSrc16 m4[4] = {{Qs, qkv + head * HD, ld3, scale}, {Ks, qkv + E + head * HD, ld3, 1.f},
                                 {Vs, qkv + 2 * E + head * HD, ld3, 1.f}, {Os, d_o + head * HD, E, 1.f}};
        // D and LSE of this wave's 16 query rows, straight from global memory (issued with the staging loads)
        const int qi = 16 * w + c;
        const size_t rowo = (size_t)(min(qi, S - 1) * B + b) * E + head * HD;
        float part = 0.f;
        float ov[KS], dv[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int d = min(4 * kk + g, HD - 1);
            ov[kk] = o[rowo + d];
            dv[kk] = d_o[rowo + d];
        }
        const float lv = lse[(size_t)bh * S + min(qi, S - 1)];
        HeadStage<HD, NT, 4> stg;
        stg.load(m4, S, B, b, tid);
        if (SAVED) {
            // every wave needs a nibble of ALL 4 NT keep words of its (c, g) column: each wave fetches one word row (one
            // coalesced dword per lane) and the rows meet in LDS (all six words per lane straight from memory measured
            // slower than the Philox calls at 640 problems: 25.9 against 23.9 us)
            kw1 = keepw[((size_t)bh * 28 + 4 * w + g) * 16 + c];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!SAVED && dc.on) {
            // the keep words of this lane's (key, 4-query group) pairs: exactly one Philox call per accumulator tile, and
            // data-independent — evaluated while the global loads are in flight
#pragma unroll
            for (int t = 0; t < NT; ++t)
                philox4((uint32_t)(bh * 28 + 4 * t + g) * 128u + (uint32_t)(16 * w + c), dc.site, dc.o0, dc.o1, dc.k0, dc.k1, wdt[t]);
        }
        stg.store(m4, S, tid);
        if (SAVED) Wk[tid] = kw1;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) part += (4 * kk + g < HD && qi < S) ? ov[kk] * dv[kk] : 0.f;
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (g == 0) {
            Ds[qi] = part;
            Ls[qi] = qi < S ? lv : 1e30f;
        }
    }
    __syncthreads();
    if (SAVED) {
        // the same registers the Philox form fills: all-ones (kept) or zero (dropped: 0 < thr) — the rest of the kernel is
        // one code path (a separate bit-test path compiled to 82 VGPRs and ran SLOWER than the Philox calls at 640 problems)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint32_t nib = Wk[64 * t + lane] >> (4 * w);
#pragma unroll
            for (int r = 0; r < 4; ++r) wdt[t][r] = ((nib >> r) & 1u) ? 0xFFFFFFFFu : 0u;
        }
    }

    const int kj = 16 * w + c;                     // this lane's key
    floatx4 ps[NT], dp[NT];
    // initial accumulators: -LSE[query] for the scores (P = exp(acc) needs no subtraction), 0 for dP
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 l4 = *reinterpret_cast<const float4*>(Ls + 16 * t + 4 * g);
        ps[t] = floatx4{-l4.x, -l4.y, -l4.z, -l4.w};
        dp[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    dot_tiles<HD, NT>(ps, Qs, Ks + kj * LD + g, c, g);      // S[query 16t+4g+reg][key kj] - LSE
    dot_tiles<HD, NT>(dp, Os, Vs + kj * LD + g, c, g);      // dP~[query][key] = dO . V

    const bool keyok = kj < S;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 d4 = *reinterpret_cast<const float4*>(Ds + 16 * t + 4 * g);
        const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = keyok ? __expf(ps[t][r]) : 0.f;             // probability (pre-dropout)
            const bool keep = !dc.on || wdt[t][r] >= dc.thr;
            const float dpk = keep ? dp[t][r] * dc.scale : 0.f;          // dP = keep * scale * dP~
            ps[t][r] = keep ? pv * dc.scale : 0.f;                       // P~
            dp[t][r] = pv * (dpk - dd[r]);                               // dS
        }
    }

    // dV^T[d][key] = sum_q dO[q][d] P~[q][key];  dK^T[d][key] = sum_q (scale Q)[q][d] dS[q][key]
    {
        floatx4 av[NTD], ak[NTD];
#pragma unroll
        for (int dt = 0; dt < NTD; ++dt) { av[dt] = floatx4{0.f, 0.f, 0.f, 0.f}; ak[dt] = floatx4{0.f, 0.f, 0.f, 0.f}; }
        const float* pv_ = Os + 4 * g * LD + c;
        const float* pk_ = Qs + 4 * g * LD + c;
        float a1[2][4][NTD], a2[2][4][NTD];
        auto fetch = [&](int t, float (&d1)[4][NTD], float (&d2)[4][NTD]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < NTD; ++dt) {
                    d1[r][dt] = pv_[(16 * t + r) * LD + 16 * dt];
                    d2[r][dt] = pk_[(16 * t + r) * LD + 16 * dt];
                }
        };
        fetch(0, a1[0], a2[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) fetch(t + 1, a1[(t + 1) & 1], a2[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < NTD; ++dt) {
                    av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t & 1][r][dt], ps[t][r], av[dt], 0, 0, 0);
                    ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t & 1][r][dt], dp[t][r], ak[dt], 0, 0, 0);
                }
        }
        if (keyok) {
            float* row = d_qkv + (size_t)(kj * B + b) * ld3 + head * HD;
#pragma unroll
            for (int dt = 0; dt < NTD; ++dt) {
                store4<HD>(row + 2 * E, 16 * dt + 4 * g, av[dt], 1.f);
                store4<HD>(row + E, 16 * dt + 4 * g, ak[dt], 1.f);
            }
        }
    }
    // dS -> LDS [query][key], over the operands every wave has finished reading
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) SS[(16 * t + 4 * g + r) * LDS_S + kj] = dp[t][r];
    __syncthreads();

    // dQ^T[d][query 16w + c] = scale * sum_key K[key][d] dS[query][key]
    {
        floatx4 pq[NT];
        const float* ss = SS + (16 * w + c) * LDS_S + 4 * g;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 q4 = *reinterpret_cast<const float4*>(ss + 16 * t);
            pq[t] = floatx4{q4.x, q4.y, q4.z, q4.w};
        }
        floatx4 aq[NTD];
#pragma unroll
        for (int dt = 0; dt < NTD; ++dt) aq[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
        apply_tiles<HD, NT>(aq, pq, Ks, c, g);
        const int qi = 16 * w + c;
        if (qi < S) {
            float* row = d_qkv + (size_t)(qi * B + b) * ld3 + head * HD;
#pragma unroll
            for (int dt = 0; dt < NTD; ++dt) store4<HD>(row, 16 * dt + 4 * g, aq[dt], scale);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// head_dim 10 / 30: always.  head_dim 60 / 64: short sequences only — measured against attention.hip (tools/lab/attn_big.py,
// B = 32): S = 33, hd 60: forward 17.1 -> 9.6 us, backward 30.4 -> 19.4 us; S = 94, hd 64: forward 17.2 -> 36.9 us
// (16 k-steps per score tile on 16-wide tiles), backward 34.1 -> 32.2 us.
bool attn16_supported(int E, int H, int S) {
    const int hd = H > 0 ? E / H : 0;
    if (!(H > 0 && E % H == 0)) return false;
    if (hd == 10 || hd == 30) return true;
    return (hd == 60 || hd == 64) && S <= 48;
}

// Hand the dropout keep bits from the forward to the backward (instead of re-evaluating the Philox calls there) only while
// the launch leaves most CUs with ONE backward workgroup: measured (tools/lab/attn_ab.py, S = 94, head_dim 10, p = 0.1)
// 16.7 -> 15.2 us at 320 problems, but 24.0 -> 25.1 us at 640 — there the Philox calls run in the shadow of the staging
// loads, while unpacking the saved words sits behind the barrier on the critical path of co-resident workgroups.
static bool attn16_use_keep(int B, int H) { return (long)B * H <= 384; }

template <int HD>
static size_t fwd_lds(int nt) { return (size_t)3 * (16 * nt * A16<HD>::LD + A16<HD>::TAIL) * sizeof(float); }
template <int HD>
static size_t bwd_lds(int nt) {
    const size_t mat = (size_t)16 * nt * A16<HD>::LD + A16<HD>::TAIL, img = (size_t)16 * nt * (16 * nt + 4);
    return (mat + 2 * 16 * nt + 64 * nt + (img > 3 * mat ? img : 3 * mat)) * sizeof(float);
}

template <int HD, int NT>
static int launch16_fwd(const float* qkv, float* o, float* lse, uint32_t* keepw, int S, int B, int E, int H, float p, uint32_t site,
                        const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    const size_t lds = fwd_lds<HD>(NT);
    if (!attn16_use_keep(B, H)) keepw = nullptr;
    // query tiles per workgroup: measured at hd = 10, S = 94 (tools/lab/attn_wpb.py, lab build): 320 problems 11.1 us as
    // whole workgroups, 10.1 / 9.9 / 11.5 us cut in 2 / 3 / 6; 640 problems 15.0 us whole, 16.4 / 17.9 / 21.5 us cut —
    // the cut pays while the problems do not fill the chip, then the repeated K / V staging costs more than the balance gains
#define GF_A16_FWD(W)                                                                                               \
    {                                                                                                               \
        GF_TRY((lds_optin<attn16_fwd_kernel<HD, NT, W>>(lds, "attention_fwd")));                                    \
        hipLaunchKernelGGL((attn16_fwd_kernel<HD, NT, W>), dim3(B * H * (NT / W)), dim3(64 * W), lds, st, qkv, o, lse, keepw, S, \
                           B, E, H, p, site, rng, add, train);                                                          \
    }
    if ((long)B * H < 512 && NT % 2 == 0 && NT > 2) GF_A16_FWD((NT % 2 == 0 ? 2 : NT))
    else GF_A16_FWD(NT)
#undef GF_A16_FWD
    GF_LAUNCH_CHECK();
    return 0;
}
template <int HD, int NT>
static int launch16_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, const uint32_t* keepw, float* d_qkv,
                        int S, int B, int E, int H, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train,
                        hipStream_t st) {
    const size_t lds = bwd_lds<HD>(NT);
    if (keepw != nullptr && train && p > 0.f && attn16_use_keep(B, H)) {          // the forward of this pass stored its keep words
        GF_TRY((lds_optin<attn16_bwd_kernel<HD, NT, true>>(lds, "attention_bwd")));
        hipLaunchKernelGGL((attn16_bwd_kernel<HD, NT, true>), dim3(B * H), dim3(64 * NT), lds, st, qkv, o, lse, d_o, keepw, d_qkv, S,
                           B, E, H, p, site, rng, add, train);
    } else {
        GF_TRY((lds_optin<attn16_bwd_kernel<HD, NT, false>>(lds, "attention_bwd")));
        hipLaunchKernelGGL((attn16_bwd_kernel<HD, NT, false>), dim3(B * H), dim3(64 * NT), lds, st, qkv, o, lse, d_o, keepw, d_qkv, S,
                           B, E, H, p, site, rng, add, train);
    }
    GF_LAUNCH_CHECK();
    return 0;
}

#define NT16_SWITCH3(FN, HD, ...)                           \
    switch ((S + 15) / 16) {                                \
        case 1: return FN<HD, 1>(__VA_ARGS__);              \
        case 2: return FN<HD, 2>(__VA_ARGS__);              \
        default: return FN<HD, 3>(__VA_ARGS__);             \
    }
#define NT16_SWITCH(FN, HD, ...)                            \
    switch ((S + 15) / 16) {                                \
        case 1: return FN<HD, 1>(__VA_ARGS__);              \
        case 2: return FN<HD, 2>(__VA_ARGS__);              \
        case 3: return FN<HD, 3>(__VA_ARGS__);              \
        case 4: return FN<HD, 4>(__VA_ARGS__);              \
        case 5: return FN<HD, 5>(__VA_ARGS__);              \
        case 6: return FN<HD, 6>(__VA_ARGS__);              \
        default: return FN<HD, 7>(__VA_ARGS__);             \
    }

int launch_attn16_fwd(const float* qkv, float* o, float* lse, uint32_t* keepw, int S, int B, int E, int H, float p, uint32_t site,
                      const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_CHECK_ARG(attn16_supported(E, H, S) && S >= 1 && S <= GANFFN_MAX_SEQ, "attn16_fwd: unsupported E=%d H=%d S=%d", E, H, S);
    GF_CHECK_ARG((long)B * H * 28 * 128 < (1l << 32), "attention: B*H too large for the Philox counter");
    if (E / H == 64) { NT16_SWITCH3(launch16_fwd, 64, qkv, o, lse, keepw, S, B, E, H, p, site, rng, add, train, st) }
    if (E / H == 60) { NT16_SWITCH3(launch16_fwd, 60, qkv, o, lse, keepw, S, B, E, H, p, site, rng, add, train, st) }
    if (E / H == 10) { NT16_SWITCH(launch16_fwd, 10, qkv, o, lse, keepw, S, B, E, H, p, site, rng, add, train, st) }
    NT16_SWITCH(launch16_fwd, 30, qkv, o, lse, keepw, S, B, E, H, p, site, rng, add, train, st)
}

int launch_attn16_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, const uint32_t* keepw, float* d_qkv,
                      int S, int B, int E, int H, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train,
                      hipStream_t st) {
    GF_CHECK_ARG(attn16_supported(E, H, S) && S >= 1 && S <= GANFFN_MAX_SEQ, "attn16_bwd: unsupported E=%d H=%d S=%d", E, H, S);
    GF_CHECK_ARG(o && lse, "attention_bwd: head_dim %d needs the forward's output and log-sum-exp", E / H);
    GF_CHECK_ARG((long)B * H * 28 * 128 < (1l << 32), "attention: B*H too large for the Philox counter");
    if (E / H == 64) { NT16_SWITCH3(launch16_bwd, 64, qkv, o, lse, d_o, keepw, d_qkv, S, B, E, H, p, site, rng, add, train, st) }
    if (E / H == 60) { NT16_SWITCH3(launch16_bwd, 60, qkv, o, lse, d_o, keepw, d_qkv, S, B, E, H, p, site, rng, add, train, st) }
    if (E / H == 10) { NT16_SWITCH(launch16_bwd, 10, qkv, o, lse, d_o, keepw, d_qkv, S, B, E, H, p, site, rng, add, train, st) }
    NT16_SWITCH(launch16_bwd, 30, qkv, o, lse, d_o, keepw, d_qkv, S, B, E, H, p, site, rng, add, train, st)
}

}  // namespace ganffn
