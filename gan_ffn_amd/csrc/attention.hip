// attention.hip — per-(dialogue, head) self-attention core, forward and backward, fp32 MFMA over LDS.
//
// Replaces the attention core of torch's nn.MultiheadAttention as used by the reference's
// nn.TransformerEncoderLayer stacks (/root/reference/model.py:1210,1244,1276,1307,1340,1377):
//   P = softmax(q k^T / sqrt(hd)) over keys, NO masks (padded utterances attend and are attended to),
//   dropout(0.1) on P in train mode, O = P v.
// The sequence is one dialogue (S <= 110 utterances), so one workgroup owns one (b, h) problem with
// q, k, v resident in LDS; nothing S x S ever touches HBM, and the backward recomputes P (and the
// Philox dropout mask) instead of saving it.
//
// Formulation: everything is computed TRANSPOSED, with the QUERY index on the MFMA lane (column) axis.
// One wave owns 32 queries.  S^T = K Q^T puts, for each lane's query, all its keys in that lane's
// accumulator registers (plus the other half-wave), so
//   * the softmax max / sum are in-lane reductions + ONE cross-half shuffle (instead of 5 shuffles per
//     register in the row-major form);
//   * the probabilities are already the B operand of the next products (O^T = V^T P^T, dQ^T = K^T dS^T):
//     an accumulator tile feeds the next MFMA directly (k index = key, permuted consistently on the
//     LDS-side operand), so P never goes through LDS in the forward pass;
//   * only the two products that contract over queries (dV, dK) need P~^T / dS^T in LDS.
// All MFMAs are v_mfma_f32_32x32x2_f32 (exact fp32).  LDS matrices use ODD row strides: row-indexed and
// column-indexed fragment reads (ds_read_b32) are both bank-conflict-free.
// Layout: qkv [T x 3E] packed q|k|v per token (t = s*B + b), head h = columns h*hd .. h*hd+hd-1.
#include "common.h"

namespace ganffn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// row inside a 32x32 accumulator tile held by (register s, lane half h)
__device__ __forceinline__ int krow(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

struct AttnGeom {
    int S, B, E, H, hd;
    int NT;   // 32-wide tiles covering S (= waves per workgroup)
    int LDH;  // row stride of the [*, hd] matrices (odd)
    int LDP;  // row stride of the S x S matrix (odd)
    int ROWS; // rows allocated per [*, hd] matrix
};

__host__ __device__ inline AttnGeom make_geom(int S, int B, int E, int H) {
    AttnGeom g;
    g.S = S; g.B = B; g.E = E; g.H = H; g.hd = E / H;
    g.NT = (S + 31) / 32;
    g.LDH = g.hd | 1;
    g.LDP = ((S + 1) & ~1) | 1;   // query columns 0 .. S_even-1 are stored; S_even <= LDP
    g.ROWS = g.NT * 32 + 1;
    return g;
}
static inline size_t hd_mat_floats(const AttnGeom& g) { return (size_t)g.ROWS * g.LDH + 64; }
static inline size_t ss_mat_floats(const AttnGeom& g) { return (size_t)g.NT * 32 * g.LDP + 64; }

// Load up to three [S x hd] head slices into LDS (rows >= S zero-filled).  All global loads of a batch of
// U items per thread are issued before the first LDS write, so a workgroup pays ONE memory round trip per
// batch instead of one per item (a plain per-item loop serialises them: that was 60-80 % of this kernel).
struct HeadSrc {
    float* dst;
    const float* src;
    int ld_src;
    float scale;
};

template <int NM>
__device__ __forceinline__ void load_heads_generic(const HeadSrc (&m)[NM], const AttnGeom& g, int b, int tid, int nthreads) {
    constexpr int U = 8;
    const int hd2 = g.hd >> 1;
    const int per = g.NT * 32 * hd2;          // float2 items per matrix
#pragma unroll
    for (int mi = 0; mi < NM; ++mi) {
        for (int base = tid; base < per; base += nthreads * U) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = min(base + u * nthreads, per - 1);
                const int s = i / hd2, d = (i - s * hd2) * 2;
                const float2 q = *reinterpret_cast<const float2*>(m[mi].src + (size_t)(min(s, g.S - 1) * g.B + b) * m[mi].ld_src + d);
                v[u] = s < g.S ? q : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * nthreads;
                if (i < per) {
                    const int s = i / hd2, d = (i - s * hd2) * 2;
                    m[mi].dst[s * g.LDH + d] = v[u].x * m[mi].scale;
                    m[mi].dst[s * g.LDH + d + 1] = v[u].y * m[mi].scale;
                }
            }
        }
    }
}

// Compile-time head_dim / tile count: every global load of all NM matrices is issued (from clamped addresses, zeroed
// by a select for the padding rows) before the first LDS write, so the workgroup pays one memory round trip in all.
template <int HD, int NT, int NM>
__device__ __forceinline__ void load_heads(const HeadSrc (&m)[NM], const AttnGeom& g, int b, int tid) {
    if constexpr (HD == 0) {
        load_heads_generic<NM>(m, g, b, tid, 64 * NT);
    } else {
        constexpr int VEC = (HD % 4 == 0) ? 4 : 2;
        constexpr int PR = HD / VEC;                  // items per row
        constexpr int PER = NT * 32 * PR;             // items per matrix
        constexpr int NTH = 64 * NT;
        constexpr int U = (PER + NTH - 1) / NTH;      // items per thread per matrix
        float v[NM][U][VEC];
        // load phase: every load unconditional, from clamped addresses, and nothing else (round 4: the `ok ? q : 0` selects
        // written here before were compiled into loads under exec-mask branches with an `s_waitcnt vmcnt(0)` behind many of
        // them — 7 load phases in the forward of the head_dim-64 kernel, 16 in its backward instead of one round trip)
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = min(tid + u * NTH, PER - 1);
                const int s = i / PR, d = (i - s * PR) * VEC;
                const float* src = m[mi].src + (size_t)(min(s, g.S - 1) * g.B + b) * m[mi].ld_src + d;
                if constexpr (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    v[mi][u][0] = q.x; v[mi][u][1] = q.y; v[mi][u][2] = q.z; v[mi][u][3] = q.w;
                } else {
                    const float2 q = *reinterpret_cast<const float2*>(src);
                    v[mi][u][0] = q.x; v[mi][u][1] = q.y;
                }
            }
        }
        // fence both ways: the memory clobber keeps every load above, the stores' 0 / scale factor comes out of the asm
        float one = 1.f;
        asm volatile("" : "+v"(one) : : "memory");
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = tid + u * NTH;
                if (PER % NTH == 0 || i < PER) {
                    const int s = i / PR, d = (i - s * PR) * VEC;
                    const float f = (s < g.S ? one : 0.f) * m[mi].scale;      // rows >= S: zero (finite operand x 0)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) m[mi].dst[s * g.LDH + d + e] = v[mi][u][e] * f;
                }
            }
        }
    }
}

// acc[c][s] (key tile c, row krow(s,h)) += sum_d Km[32c + r][d] * Qm[q0 + r][d]   ->  S^T tiles for this wave's queries
// NOTE: tile counts (NT key/query tiles, NTD head-dim tiles) are COMPILE-TIME: an MFMA under a run-time
// condition makes hipcc copy the 16-register accumulator tuples at every control-flow merge (measured:
// ~60 v_accvgpr/v_mov per MFMA, 750 cycles per MFMA instead of 64).
template <int HD, int NT>
__device__ __forceinline__ void scores_T(floatx16 (&acc)[NT], const float* __restrict__ Km,
                                         const float* __restrict__ Qm, int LDH, int hd, int q0, int r, int h) {
    const float* pb = Qm + (q0 + r) * LDH + h;
    const float* pa = Km + r * LDH + h;
    if constexpr (HD != 0) {
        // operands of a whole batch of k-steps are read from LDS before the first MFMA of the batch (left to itself
        // hipcc reuses one register: ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma, a full LDS latency per MFMA)
        constexpr int STEPS = HD >> 1;
        constexpr int BT = STEPS < 8 ? STEPS : (STEPS % 8 == 0 ? 8 : (STEPS % 6 == 0 ? 6 : (STEPS % 5 == 0 ? 5 : 1)));
        static_assert(STEPS % BT == 0 && BT > 1, "head_dim/2 must be a multiple of the batch");
#pragma unroll
        for (int s0 = 0; s0 < STEPS; s0 += BT) {
            float bv[BT], av[BT][NT];
#pragma unroll
            for (int j = 0; j < BT; ++j) {
                bv[j] = pb[2 * (s0 + j)];
#pragma unroll
                for (int c = 0; c < NT; ++c) av[j][c] = pa[c * 32 * LDH + 2 * (s0 + j)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < BT; ++j)
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][c], bv[j], acc[c], 0, 0, 0);
        }
    } else {
        const int steps = hd >> 1;
#pragma unroll 4
        for (int s = 0; s < steps; ++s) {
            const float b = pb[2 * s];
#pragma unroll
            for (int c = 0; c < NT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c * 32 * LDH + 2 * s], b, acc[c], 0, 0, 0);
        }
    }
}

// out[dt][s] (row d = 32dt + krow(s,h), col = query) += sum_keys Mm[key][d] * P[c][s]   (P = accumulator tiles as B operand)
template <int NT, int NTD>
__device__ __forceinline__ void apply_T(floatx16 (&out)[NTD], const floatx16 (&P)[NT],
                                        const float* __restrict__ Mm, int LDH, int r, int h) {
    float av[2][16][NTD];
    auto fetch = [&](int c, float (&dst)[16][NTD]) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* row = Mm + (32 * c + krow(s, h)) * LDH + r;
#pragma unroll
            for (int dt = 0; dt < NTD; ++dt) dst[s][dt] = row[32 * dt];
        }
    };
    fetch(0, av[0]);
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        if (c + 1 < NT) fetch(c + 1, av[(c + 1) & 1]);   // next key tile's operands are in flight under this tile's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float b = P[c][s];
#pragma unroll
            for (int dt = 0; dt < NTD; ++dt)
                out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][s][dt], b, out[dt], 0, 0, 0);
        }
    }
}

// acc[t] (rows 32w + .., cols d) += sum_i A[row][i] * Bm[i][32t + r]  over i = 0 .. 2*ks-1, operands batched like above
template <int NTD>
__device__ __forceinline__ void rows_T(floatx16 (&acc)[NTD], const float* __restrict__ pa, const float* __restrict__ pb,
                                       int LDH, int ks) {
    constexpr int BT = 8;
    int s = 0;
    for (; s + BT <= ks; s += BT) {
        float a[BT], bv[BT][NTD];
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            a[j] = pa[2 * (s + j)];
#pragma unroll
            for (int t = 0; t < NTD; ++t) bv[j][t] = pb[2 * (s + j) * LDH + 32 * t];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < BT; ++j)
#pragma unroll
            for (int t = 0; t < NTD; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bv[j][t], acc[t], 0, 0, 0);
    }
    for (; s < ks; ++s) {
        const float a = pa[2 * s];
#pragma unroll
        for (int t = 0; t < NTD; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[2 * s * LDH + 32 * t], acc[t], 0, 0, 0);
    }
}

// softmax over keys for each lane's query; P^T tiles in place
template <int NT>
__device__ __forceinline__ void softmax_T(floatx16 (&p)[NT], int S, int h) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (32 * c + krow(s, h) >= S) p[c][s] = -INFINITY;
            m = fmaxf(m, p[c][s]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float e = __expf(p[c][s] - m);   // exp(-inf) = 0 for padded keys
            p[c][s] = e;
            sum += e;
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int s = 0; s < 16; ++s) p[c][s] *= inv;
    }
}

// Dropout keep-bits for this lane's query and all its keys.  Contract: one Philox call = 4 consecutive
// queries at one key; the 4 lanes of a quad are 4 consecutive queries, so each lane evaluates the calls of
// the registers s with (s & 3) == (lane & 3) and the quad exchanges 64-bit masks.
// keep(c, s) = (mq[s & 3] >> (((c * 4 + (s >> 2)) * 4) + (lane & 3))) & 1
template <int NT>
__device__ __forceinline__ void keep_masks(unsigned long long (&mq)[4], const DropCtx& dc, int bh, int w, int lane) {
    if (!dc.on) {
        mq[0] = mq[1] = mq[2] = mq[3] = ~0ull;
        return;
    }
    const int r = lane & 31, h = lane >> 5, ql = lane & 3;
    unsigned long long mine = 0ull;
    const uint32_t rowgroup = (uint32_t)(bh * 28 + 8 * w + (r >> 2));
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int j = 32 * c + krow(4 * gq + ql, h);
            uint32_t wd[4];
            philox4(rowgroup * 128u + (uint32_t)j, dc.site, dc.o0, dc.o1, dc.k0, dc.k1, wd);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (wd[q] >= dc.thr) mine |= 1ull << ((c * 4 + gq) * 4 + q);
        }
    }
    const uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int src = (lane & ~3) | q;
        const uint32_t l2 = __shfl(lo, src, 64), h2 = __shfl(hi, src, 64);
        mq[q] = ((unsigned long long)h2 << 32) | l2;
    }
}
__device__ __forceinline__ bool kept(const unsigned long long (&mq)[4], int c, int s, int lane) {
    return (mq[s & 3] >> (((c * 4 + (s >> 2)) * 4) + (lane & 3))) & 1ull;
}

// ------------------------------------------------------------------------------------------
// forward: one wave per 32 queries, blockDim = 64 * NT
// ------------------------------------------------------------------------------------------
template <int HD, int NT>
__global__ __launch_bounds__(64 * NT) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o, AttnGeom g,
                                                                float p, uint32_t site, const uint64_t* __restrict__ rng,
                                                                uint64_t add, int train) {
    constexpr int NTD = HD ? (HD + 31) / 32 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / g.H, head = bh % g.H;
    const size_t HM = (size_t)g.ROWS * g.LDH + 64;
    float* Qs = smem;
    float* Ks = Qs + HM;
    float* Vs = Ks + HM;
    const int ld3 = 3 * g.E;
    const float scale = rsqrtf((float)g.hd);
    {
        const HeadSrc m3[3] = {{Qs, qkv + head * g.hd, ld3, scale}, {Ks, qkv + g.E + head * g.hd, ld3, 1.f},
                               {Vs, qkv + 2 * g.E + head * g.hd, ld3, 1.f}};
        load_heads<HD, NT, 3>(m3, g, b, tid);
    }
    __syncthreads();

    floatx16 pr[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[c][i] = 0.f;
    scores_T<HD, NT>(pr, Ks, Qs, g.LDH, g.hd, 32 * w, r, h);
    softmax_T<NT>(pr, g.S, h);

    const DropCtx dc = make_drop(rng, add, site, p, train);
    if (dc.on) {
        unsigned long long mq[4];
        keep_masks<NT>(mq, dc, bh, w, lane);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int s = 0; s < 16; ++s) pr[c][s] = kept(mq, c, s, lane) ? pr[c][s] * dc.scale : 0.f;
        }
    }

    floatx16 oacc[NTD];
#pragma unroll
    for (int t = 0; t < NTD; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
    apply_T<NT, NTD>(oacc, pr, Vs, g.LDH, r, h);

    const int i = 32 * w + r;
    if (i < g.S) {
        float* orow = o + (size_t)(i * g.B + b) * g.E + head * g.hd;
#pragma unroll
        for (int t = 0; t < NTD; ++t) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int d = 32 * t + krow(s, h);
                if (d < g.hd) orow[d] = oacc[t][s];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward: d_qkv from d_o, recomputing P^T and the dropout mask
// ------------------------------------------------------------------------------------------
template <int HD, int NT>
__global__ __launch_bounds__(64 * NT) void attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                float* __restrict__ d_qkv, AttnGeom g, float p, uint32_t site,
                                                                const uint64_t* __restrict__ rng, uint64_t add, int train) {
    constexpr int NTD = HD ? (HD + 31) / 32 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / g.H, head = bh % g.H;
    const size_t HM = (size_t)g.ROWS * g.LDH + 64;
    // All four operand matrices stay resident in LDS when they fit (one global-load round trip, one barrier fewer);
    // otherwise dO overwrites Q and the re-loaded Q overwrites V (ALIAS).  Same rule on the host: launch_attention_bwd().
    constexpr bool ALIAS = (HD == 0) || (HD * NT > 192);
    float* RA = smem;                        // scaled Q   (ALIAS: later dO)
    float* RB = RA + HM;                     // K
    float* RC = RB + HM;                     // V          (ALIAS: later scaled Q)
    float* RD = ALIAS ? RA : RC + HM;        // dO
    float* SS = (ALIAS ? RC : RD) + HM;      // P~^T [key][query], later dS^T
    float* QS = ALIAS ? RC : RA;             // scaled Q as read by the dK product
    const int ld3 = 3 * g.E;
    const float scale = rsqrtf((float)g.hd);

    if constexpr (ALIAS) {
        const HeadSrc m3[3] = {{RA, qkv + head * g.hd, ld3, scale}, {RB, qkv + g.E + head * g.hd, ld3, 1.f},
                               {RC, qkv + 2 * g.E + head * g.hd, ld3, 1.f}};
        load_heads<HD, NT, 3>(m3, g, b, tid);
    } else {
        const HeadSrc m4[4] = {{RA, qkv + head * g.hd, ld3, scale}, {RB, qkv + g.E + head * g.hd, ld3, 1.f},
                               {RC, qkv + 2 * g.E + head * g.hd, ld3, 1.f}, {RD, d_o + head * g.hd, g.E, 1.f}};
        load_heads<HD, NT, 4>(m4, g, b, tid);
    }
    __syncthreads();

    floatx16 pr[NT];   // P^T (pre-dropout probabilities), later dS^T
    floatx16 dp[NT];   // dP~^T
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { pr[c][i] = 0.f; dp[c][i] = 0.f; }

    scores_T<HD, NT>(pr, RB, RA, g.LDH, g.hd, 32 * w, r, h);
    softmax_T<NT>(pr, g.S, h);
    const DropCtx dc = make_drop(rng, add, site, p, train);
    unsigned long long mq[4];
    keep_masks<NT>(mq, dc, bh, w, lane);
    // P~^T -> LDS [key][query] (this wave's 32 query columns)
    {
        const int i = 32 * w + r;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float v = kept(mq, c, s, lane) ? pr[c][s] * dc.scale : 0.f;
                if (i < g.LDP) SS[(32 * c + krow(s, h)) * g.LDP + i] = v;
            }
        }
    }
    __syncthreads();                                           // all waves are done with Q; P~^T complete
    if constexpr (ALIAS) {
        const HeadSrc m1[1] = {{RA, d_o + head * g.hd, g.E, 1.f}};   // dO over Q
        load_heads<HD, NT, 1>(m1, g, b, tid);
        __syncthreads();
    }

    // dP~^T = V dO^T (same structure as the scores)
    scores_T<HD, NT>(dp, RC, RD, g.LDH, g.hd, 32 * w, r, h);
    // D_i = sum_j dP_ij P_ij with dP = keep*scale*dP~ ;  dS = P * (dP - D)
    {
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                dp[c][s] = kept(mq, c, s, lane) ? dp[c][s] * dc.scale : 0.f;
                dsum += dp[c][s] * pr[c][s];
            }
        }
        dsum += __shfl_xor(dsum, 32, 64);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int s = 0; s < 16; ++s) pr[c][s] = pr[c][s] * (dp[c][s] - dsum);
        }
    }
    // dQ^T = scale * K^T dS^T  (accumulator tiles feed the MFMA directly)
    {
        floatx16 acc[NTD];
#pragma unroll
        for (int t = 0; t < NTD; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        apply_T<NT, NTD>(acc, pr, RB, g.LDH, r, h);
        const int i = 32 * w + r;
        if (i < g.S) {
            float* row = d_qkv + (size_t)(i * g.B + b) * ld3 + head * g.hd;
#pragma unroll
            for (int t = 0; t < NTD; ++t) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int d = 32 * t + krow(s, h);
                    if (d < g.hd) row[d] = acc[t][s] * scale;
                }
            }
        }
    }
    // dV[j][d] = sum_i P~^T[j][i] dO[i][d]   (this wave: key rows 32w .. 32w+31)
    {
        floatx16 acc[NTD];
#pragma unroll
        for (int t = 0; t < NTD; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const float* pa = SS + (32 * w + r) * g.LDP + h;
        const float* pb = RD + h * g.LDH + r;
        const int ks = (g.S + 1) >> 1;   // queries 0 .. S_even-1 (rows beyond S of dO / Q are zero)
        rows_T<NTD>(acc, pa, pb, g.LDH, ks);
#pragma unroll
        for (int t = 0; t < NTD; ++t) {
            const int d = 32 * t + r;
            if (d < g.hd) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int j = 32 * w + krow(s, h);
                    if (j < g.S) d_qkv[(size_t)(j * g.B + b) * ld3 + 2 * g.E + head * g.hd + d] = acc[t][s];
                }
            }
        }
    }
    __syncthreads();                                           // P~^T, V and dO are dead
    {
        const int i = 32 * w + r;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int j = 32 * c + krow(s, h);
                if (i < g.LDP) SS[j * g.LDP + i] = (i < g.S && j < g.S) ? pr[c][s] : 0.f;
            }
        }
    }
    if constexpr (ALIAS) {
        const HeadSrc m1[1] = {{RC, qkv + head * g.hd, ld3, scale}};  // scaled Q over V
        load_heads<HD, NT, 1>(m1, g, b, tid);
    }
    __syncthreads();
    // dK[j][d] = sum_i dS^T[j][i] (scale*Q)[i][d]
    {
        floatx16 acc[NTD];
#pragma unroll
        for (int t = 0; t < NTD; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const float* pa = SS + (32 * w + r) * g.LDP + h;
        const float* pb = QS + h * g.LDH + r;
        const int ks = (g.S + 1) >> 1;   // queries 0 .. S_even-1 (rows beyond S of dO / Q are zero)
        rows_T<NTD>(acc, pa, pb, g.LDH, ks);
#pragma unroll
        for (int t = 0; t < NTD; ++t) {
            const int d = 32 * t + r;
            if (d < g.hd) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int j = 32 * w + krow(s, h);
                    if (j < g.S) d_qkv[(size_t)(j * g.B + b) * ld3 + g.E + head * g.hd + d] = acc[t][s];
                }
            }
        }
    }
}

static int check_attn(int S, int B, int E, int H) {
    GF_CHECK_ARG(S >= 1 && S <= GANFFN_MAX_SEQ, "attention: S=%d out of range [1,%d]", S, GANFFN_MAX_SEQ);
    GF_CHECK_ARG(B >= 1 && H >= 1 && E % H == 0, "attention: bad B=%d E=%d H=%d", B, E, H);
    const int hd = E / H;
    GF_CHECK_ARG((hd & 1) == 0 && hd <= 64, "attention: head_dim=%d must be even and <= 64", hd);
    GF_CHECK_ARG((long)B * H * 28 * 128 < (1l << 32), "attention: B*H too large for the Philox counter");
    return 0;
}

template <int HD, int NT>
static int launch_fwd_t(const float* qkv, float* o, const AttnGeom& g, size_t lds, float p, uint32_t site, const uint64_t* rng,
                        uint64_t add, int train, hipStream_t st) {
    GF_TRY((lds_optin<attention_fwd_kernel<HD, NT>>(lds, "attention_fwd")));
    hipLaunchKernelGGL((attention_fwd_kernel<HD, NT>), dim3(g.B * g.H), dim3(64 * NT), lds, st, qkv, o, g, p, site, rng, add, train);
    GF_LAUNCH_CHECK();
    return 0;
}
template <int HD, int NT>
static int launch_bwd_t(const float* qkv, const float* d_o, float* d_qkv, const AttnGeom& g, size_t lds, float p, uint32_t site,
                        const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_TRY((lds_optin<attention_bwd_kernel<HD, NT>>(lds, "attention_bwd")));
    hipLaunchKernelGGL((attention_bwd_kernel<HD, NT>), dim3(g.B * g.H), dim3(64 * NT), lds, st, qkv, d_o, d_qkv, g, p, site, rng,
                       add, train);
    GF_LAUNCH_CHECK();
    return 0;
}

#define NT_SWITCH(FN, HD, ...)                                  \
    switch (g.NT) {                                             \
        case 1: return FN<HD, 1>(__VA_ARGS__);                  \
        case 2: return FN<HD, 2>(__VA_ARGS__);                  \
        case 3: return FN<HD, 3>(__VA_ARGS__);                  \
        default: return FN<HD, 4>(__VA_ARGS__);                 \
    }

int launch_attention_fwd(const float* qkv, float* o, float* lse, uint32_t* keep, int S, int B, int E, int H, float p, uint32_t site,
                         const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_TRY(check_attn(S, B, E, H));
    GF_CHECK_ARG(qkv && o, "attention_fwd: null pointer");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "attention_fwd: rng required when dropout is active");
    if (attn16_supported(E, H, S)) return launch_attn16_fwd(qkv, o, lse, keep, S, B, E, H, p, site, rng, add, train, st);
    const AttnGeom g = make_geom(S, B, E, H);
    const size_t lds = 3 * hd_mat_floats(g) * sizeof(float);
    GF_CHECK_ARG(lds <= 160 * 1024, "attention_fwd: LDS need %zu > 160 KiB", lds);
    if (g.hd == 64) { NT_SWITCH(launch_fwd_t, 64, qkv, o, g, lds, p, site, rng, add, train, st) }
    if (g.hd == 60) { NT_SWITCH(launch_fwd_t, 60, qkv, o, g, lds, p, site, rng, add, train, st) }
    NT_SWITCH(launch_fwd_t, 0, qkv, o, g, lds, p, site, rng, add, train, st)
}

int launch_attention_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, const uint32_t* keep, float* d_qkv,
                         int S, int B, int E, int H, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train,
                         hipStream_t st) {
    GF_TRY(check_attn(S, B, E, H));
    GF_CHECK_ARG(qkv && d_o && d_qkv, "attention_bwd: null pointer");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "attention_bwd: rng required when dropout is active");
    if (attn16_supported(E, H, S)) return launch_attn16_bwd(qkv, o, lse, d_o, keep, d_qkv, S, B, E, H, p, site, rng, add, train, st);
    const AttnGeom g = make_geom(S, B, E, H);
    const int hdt = (g.hd == 60 || g.hd == 64) ? g.hd : 0;      // kernel template head_dim (0 = generic)
    const bool alias = hdt == 0 || hdt * g.NT > 192;            // == attention_bwd_kernel::ALIAS
    const size_t lds = ((alias ? 3 : 4) * hd_mat_floats(g) + ss_mat_floats(g)) * sizeof(float);
    GF_CHECK_ARG(lds <= 160 * 1024, "attention_bwd: LDS need %zu > 160 KiB", lds);
    if (g.hd == 64) { NT_SWITCH(launch_bwd_t, 64, qkv, d_o, d_qkv, g, lds, p, site, rng, add, train, st) }
    if (g.hd == 60) { NT_SWITCH(launch_bwd_t, 60, qkv, d_o, d_qkv, g, lds, p, site, rng, add, train, st) }
    NT_SWITCH(launch_bwd_t, 0, qkv, d_o, d_qkv, g, lds, p, site, rng, add, train, st)
}

}  // namespace ganffn
