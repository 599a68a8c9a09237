// rowchain.hip — d_model = 100 encoder layers: the token-local chains around the two LayerNorms as single kernels.
//
// In a post-LN nn.TransformerEncoderLayer (/root/reference/model.py:1210 -> torch _sa_block / norm1 / _ff_block / norm2)
// everything between the attention core and the feed-forward GEMMs is local to a token row:
//     forward   A: attn_o --out_proj--> (+bias) --dropout--> (+x) --LayerNorm1--> x1
//               B: linear2 slabs (+x1) --dropout, LayerNorm2--> X[l+1] --in_proj of layer l+1--> qkv[l+1]
//     backward  A: d_qkv[l+1] --in_proj dgrad of layer l+1--> (+dz1[l+1]) --LayerNorm2 backward--> dz2, dyA
//               B: dh.W1 slabs (+dz2) --LayerNorm1 backward--> dz1, dyB --out_proj dgrad--> d_attn
// As separate launches these were a [T x 100] x [100 x 100] GEMM (2 x 47 = 94 workgroups on 256 CUs), a [T x 100] x
// [100 x 300] GEMM (235 workgroups) and a LayerNorm pass each: 5-12 us launch floors, ~18 % of the GPU time of a
// d_model-100 pass for ~3 % of its FLOPs (profiles/r02_bench_streams1_by_launch_shape.txt).  Here a workgroup owns 16
// token rows and runs the whole chain for them: the small GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32) with the weights
// read straight from L2 into the operand layout, the LayerNorm statistics by two shuffles + one LDS exchange between the
// 4 waves, the row tile handed from the LayerNorm stage to the following GEMM through LDS.  4 of the 9 launches of a
// layer disappear in each direction.
//
// These kernels are LATENCY-sized (188-376 workgroups, < 1 wave per SIMD), so they are written for few dependent round
// trips and few memory instructions:
//  * MFMA orientation: D[m][n] with m = output FEATURE, n = TOKEN.  Lane (c, g) (lane = 16 g + c) supplies A[m = c][k = g]
//    (a weight row) and B[k = g][n = c] (a token row) and receives D[m = 4 g + r][n = c], r = 0..3: one token, 4 CONSECUTIVE
//    features — every global access of the element-wise part (slabs, residual, outputs) is one 16-byte access per lane;
//  * k order: lane group g takes k = 16 q + 4 g + j for the j-th MFMA of group q (one 16-byte load per operand and group),
//    the same permutation on both operands and the same for every token, so a dialogue's bits do not depend on its
//    position in the batch (tests/test_hip_properties.py::test_full_size_batch_permutation...);
//  * every load that does not depend on computed data (weights of BOTH GEMM stages, slabs, residual, LayerNorm
//    parameters) is issued before the first wait;
//  * the 4 waves split the 7 feature tiles of a 100-wide row (wave w owns tiles w and w + 4; a tile beyond the last is
//    computed on a clamped copy and discarded, so no MFMA sits under a run-time condition); the K = 300 product of the
//    in-proj dgrad is split over the waves along K instead (each wave: all 7 tiles, a quarter of K; partial tiles are
//    summed through LDS in wave order) so that its 54 operand loads per lane are one round trip;
//  * dropout: the Philox contract is "one call = 4 consecutive TOKENS of one column" (common.h drop_mult4); lane ql of a
//    quad (4 consecutive tokens, same features) evaluates the call of feature f0 + ql and the quad exchanges keep bits
//    by DPP — one call per tile and lane, as in the GEMM epilogues.
// Deterministic: no atomics; the LayerNorm parameter gradients leave as per-workgroup partial rows (summed over the 16
// tokens in token order) that ln_param_reduce_kernel (elementwise.hip) adds in block order.
#include "common.h"

#pragma clang fp contract(off)

namespace ganffn {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RE = 100;              // d_model handled here
constexpr int RT = (RE + 15) / 16;   // 7 feature tiles of 16 over a 100-wide row
constexpr int LDX = 116;             // LDS row stride of a 16 x 100 row tile (zero-padded to 112 columns)

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// one lane's operand fragments over K (a multiple of 4) from a K-contiguous row: full group q = floats 16 q + 4 g .. + 3 (four
// MFMAs, lane group g taking k = 16 q + 4 g + j at the j-th); the K % 16 values behind the last full group go ONE k per lane
// group and MFMA (k = 16 KF + 4 j + g): K = 100 is 6 groups + 1 MFMA = 25 MFMAs per tile, not 7 groups = 28 with three
// quarters of the last four empty (round 4)
template <int K>
struct Frag {
    static_assert(K % 4 == 0, "K must be a multiple of 4");
    static constexpr int KF = K / 16, R = (K % 16) / 4;
    float4 v[KF];
    float t[R > 0 ? R : 1];
    __device__ __forceinline__ void load(const float* __restrict__ row, int g) {
#pragma unroll
        for (int q = 0; q < KF; ++q) v[q] = f4(row + 16 * q + 4 * g);
#pragma unroll
        for (int j = 0; j < R; ++j) t[j] = row[16 * KF + 4 * j + g];
    }
};

// acc[i] += W_i (16 features x K) . X^T (K x 16 tokens) for NI feature tiles; interleaved accumulator chains
template <int K, int NI>
__device__ __forceinline__ void mma(floatx4 (&acc)[NI], const Frag<K> (&wf)[NI], const Frag<K>& xf) {
#pragma unroll
    for (int q = 0; q < Frag<K>::KF; ++q) {
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].v[q].x, xf.v[q].x, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].v[q].y, xf.v[q].y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].v[q].z, xf.v[q].z, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].v[q].w, xf.v[q].w, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < Frag<K>::R; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i].t[j], xf.t[j], acc[i], 0, 0, 0);
}

template <int R>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {          // value of lane R of this lane's quad (DPP quad_perm)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, R * 0x55, 0xF, 0xF, true);
}

// dropout multipliers of (this lane's token, features f0 .. f0 + 3): lane ql of the quad evaluates the Philox call of
// feature f0 + ql (4 consecutive tokens = the quad) and the keep bits are exchanged inside the quad
__device__ __forceinline__ void drop_mult_quad(const DropCtx& dc, uint32_t rowgroup, int f0, int ql, float (&m)[4]) {
    if (!dc.on) {
        m[0] = m[1] = m[2] = m[3] = 1.f;
        return;
    }
    uint32_t wd[4];
    philox4(rowgroup * (uint32_t)RE + (uint32_t)(f0 + ql), dc.site, dc.o0, dc.o1, dc.k0, dc.k1, wd);
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) mine |= (wd[j] >= dc.thr ? 1u : 0u) << j;
    const uint32_t b[4] = {quad_bcast<0>(mine), quad_bcast<1>(mine), quad_bcast<2>(mine), quad_bcast<3>(mine)};
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = ((b[r] >> ql) & 1u) ? dc.scale : 0.f;
}

// per-token sum over the 4 lane groups of a wave (this wave's features), then over the 4 waves through LDS (fixed order)
__device__ __forceinline__ float token_allreduce(float v, float* __restrict__ red, int w, int c, int g) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0) red[w * 16 + c] = v;
    __syncthreads();
    return ((red[c] + red[16 + c]) + red[32 + c]) + red[48 + c];
}

struct RcFwdArgs {
    // leading stage (PRE): 1: y = pre_a . pre_w^T + pre_b (GEMM);  0: y = sum of nslab partial slabs;
    // 2: no LayerNorm at all — out = dropout(x + y[t / B]) with y the positional-encoding table (model.py:1196-1197)
    const float* pre_a; const float* pre_w; const float* pre_b;
    const float* y; int nslab; long slab_stride;
    const float* x;                  // residual input [T x E]
    const float* gamma; const float* beta;
    float* out; float* xhat; float* rstd;     // xhat / rstd may be null (nothing kept for backward)
    // trailing stage (optional): post_out [T x 3E] = out . post_w^T + post_b
    const float* post_w; const float* post_b; float* post_out;
    int T, B; float eps, p; uint32_t site; const uint64_t* rng; uint64_t add; int train;
};

// z = x + dropout(y); out = LayerNorm(z); optionally the next layer's in-proj on the fresh rows
template <int PRE, bool POST_GEMM>
__global__ __launch_bounds__(256) void rc_fwd_kernel(RcFwdArgs a) {
    constexpr int NP = 3 * RE, PT = (NP + 15) / 16;     // in-proj: 300 output features, 19 tiles
    __shared__ __attribute__((aligned(16))) float red[2 * 64];
    __shared__ __attribute__((aligned(16))) float xs[POST_GEMM ? 16 * LDX : 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4, ql = c & 3;
    const int t0 = blockIdx.x * 16, T = a.T;
    const bool tok = t0 + c < T;
    const size_t trow = (size_t)min(t0 + c, T - 1);
    const DropCtx dc = make_drop(a.rng, a.add, a.site, a.p, a.train);

    // this wave's feature tiles: w and w + 4 (tile 7 does not exist: wave 3 recomputes tile 6 and discards it)
    int mt[2], f0[2], f0c[2];
    bool fok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        mt[i] = min(w + 4 * i, RT - 1);
        f0[i] = 16 * mt[i] + 4 * g;
        fok[i] = (w + 4 * i < RT) && f0[i] < RE;
        f0c[i] = min(f0[i], RE - 4);
    }

    // ---------------- every independent load, up front ----------------
    Frag<RE> wpost[3];                   // trailing GEMM, tiles w, w + 4, w + 8 (all < 19)
    if constexpr (POST_GEMM) {
#pragma unroll
        for (int j = 0; j < 3; ++j) wpost[j].load(a.post_w + (size_t)min(16 * (w + 4 * j) + c, NP - 1) * RE, g);
    }
    Frag<RE> xf, wf[2];
    float4 bias[2], ysum[2];
    if constexpr (PRE == 1) {
        xf.load(a.pre_a + trow * RE, g);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wf[i].load(a.pre_w + (size_t)min(16 * mt[i] + c, RE - 1) * RE, g);
            bias[i] = f4(a.pre_b + f0c[i]);
        }
    }
    float4 xv[2], gam[2], bet[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        xv[i] = f4(a.x + trow * RE + f0c[i]);
        if constexpr (PRE == 2) {
            gam[i] = bet[i] = zero4();
            ysum[i] = f4(a.y + (trow / (size_t)a.B) * RE + f0c[i]);
        } else {
            gam[i] = f4(a.gamma + f0c[i]);
            bet[i] = f4(a.beta + f0c[i]);
            ysum[i] = zero4();
        }
    }
    if constexpr (PRE == 0) {
        for (int s0 = 0; s0 < a.nslab; s0 += 8) {            // 8 slabs' loads in flight at a time, added in slab order
            float4 v[8][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* ys = a.y + (size_t)min(s0 + j, a.nslab - 1) * a.slab_stride + trow * RE;
#pragma unroll
                for (int i = 0; i < 2; ++i) v[j][i] = f4(ys + f0c[i]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float m = (s0 + j < a.nslab) ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ysum[i].x = __fmaf_rn(m, v[j][i].x, ysum[i].x); ysum[i].y = __fmaf_rn(m, v[j][i].y, ysum[i].y);
                    ysum[i].z = __fmaf_rn(m, v[j][i].z, ysum[i].z); ysum[i].w = __fmaf_rn(m, v[j][i].w, ysum[i].w);
                }
            }
        }
    }

    __builtin_amdgcn_sched_barrier(0);   // (keep the loads above ahead of everything below: one round trip)

    // ---------------- leading GEMM ----------------
    float y[2][4];
    if constexpr (PRE == 1) {
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        mma<RE, 2>(acc, wf, xf);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            y[i][0] = acc[i][0] + bias[i].x; y[i][1] = acc[i][1] + bias[i].y;
            y[i][2] = acc[i][2] + bias[i].z; y[i][3] = acc[i][3] + bias[i].w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) { y[i][0] = ysum[i].x; y[i][1] = ysum[i].y; y[i][2] = ysum[i].z; y[i][3] = ysum[i].w; }
    }

    // ---------------- residual + dropout + LayerNorm ----------------
    float z[2][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float mult[4];
        drop_mult_quad(dc, (uint32_t)(t0 / 4 + (c >> 2)), f0c[i], ql, mult);
        const float xr[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (PRE == 2) z[i][r] = __fmul_rn(__fadd_rn(xr[r], y[i][r]), mult[r]);
            else z[i][r] = fok[i] ? __fmaf_rn(y[i][r], mult[r], xr[r]) : 0.f;
            sum += z[i][r];
        }
    }
    const float invE = 1.0f / (float)RE;
    float mean = 0.f, rs = 1.f;
    if constexpr (PRE != 2) {
        mean = token_allreduce(sum, red, w, c, g) * invE;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = fok[i] ? __fsub_rn(z[i][r], mean) : 0.f;
                var = __fmaf_rn(d, d, var);
            }
        rs = rsqrtf(token_allreduce(var, red + 64, w, c, g) * invE + a.eps);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float gm[4] = {gam[i].x, gam[i].y, gam[i].z, gam[i].w}, bt[4] = {bet[i].x, bet[i].y, bet[i].z, bet[i].w};
        float xh[4], o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[r] = __fmul_rn(__fsub_rn(z[i][r], mean), rs);
            o[r] = PRE == 2 ? z[i][r] : __fmaf_rn(xh[r], gm[r], bt[r]);
        }
        if (fok[i] && tok) {
            const size_t off = trow * RE + f0[i];
            if (a.xhat) *reinterpret_cast<float4*>(a.xhat + off) = make_float4(xh[0], xh[1], xh[2], xh[3]);
            *reinterpret_cast<float4*>(a.out + off) = make_float4(o[0], o[1], o[2], o[3]);
        }
        if constexpr (POST_GEMM) {
            // row tile -> LDS for the trailing GEMM; columns 100 .. 111 zero (tile 6's lane groups g >= 1)
            if (w + 4 * i < RT)
                *reinterpret_cast<float4*>(xs + c * LDX + f0[i]) = fok[i] ? make_float4(o[0], o[1], o[2], o[3]) : zero4();
        }
    }
    if (a.rstd && w == 0 && g == 0 && tok) a.rstd[t0 + c] = rs;

    // ---------------- trailing GEMM: in-proj of the next layer ----------------
    if constexpr (POST_GEMM) {
        // second tile group (w + 12, w + 16) in flight while the first one multiplies
        Frag<RE> wpost2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) wpost2[j].load(a.post_w + (size_t)min(16 * (w + 12 + 4 * j) + c, NP - 1) * RE, g);
        float4 pb[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) pb[j] = f4(a.post_b + min(16 * (w + 4 * j) + 4 * g, NP - 4));
        __syncthreads();
        Frag<RE> af;
        af.load(xs + c * LDX, g);
        floatx4 acc[3] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        mma<RE, 3>(acc, wpost, af);
        floatx4 acc2[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        mma<RE, 2>(acc2, wpost2, af);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int pf = 16 * (w + 4 * j) + 4 * g;        // (tile 19 = wave 3's fifth: beyond the matrix, discarded)
            const floatx4 r4 = j < 3 ? acc[j] : acc2[j - 3];
            if (tok && (w + 4 * j) < PT && pf < NP)
                *reinterpret_cast<float4*>(a.post_out + trow * NP + pf) =
                    make_float4(r4[0] + pb[j].x, r4[1] + pb[j].y, r4[2] + pb[j].z, r4[3] + pb[j].w);
        }
    }
}

struct RcBwdArgs {
    // incoming gradient d = (pre_a . pre_wt^T | sum of slabs | one tensor) + addend
    const float* pre_a; const float* pre_wt;     // PRE = 2: d_qkv of the layer above [T x 3E], its in-proj weight TRANSPOSED [E x 3E]
    const float* d_out; int nslab; long slab_stride;
    const float* addend;                         // may be null
    const float* xhat; const float* rstd; const float* gamma;
    float* dz; float* dy;                        // dy = dz * dropout multiplier (the branch through the dropout)
    float* gpart;                                // [gridDim.x][2][E] partial sums of the LayerNorm weight / bias gradients (may be null)
    const float* post_wt; float* post_out;       // POST: post_out [T x E] = dy . post_wt^T   (post_wt = out_proj weight transposed)
    int T; float p; uint32_t site; const uint64_t* rng; uint64_t add; int train;
};

// PRE: 0 = d_out is one tensor, 1 = d_out is nslab partial slabs, 2 = d = pre_a . pre_wt^T (K = 3E)
template <int PRE, bool POST_GEMM>
__global__ __launch_bounds__(256) void rc_bwd_kernel(RcBwdArgs a) {
    constexpr int K3 = 3 * RE, KQ3 = (K3 + 15) / 16, NQW = (KQ3 + 3) / 4;      // 19 k groups, <= 5 per wave
    constexpr int LDP = 112;
    // pp: PRE = 2: the waves' partial tiles [4][7][64] float4; afterwards (all kernels) the two 16 x 112 tiles whose column
    // sums are the LayerNorm parameter gradients
    __shared__ __attribute__((aligned(16))) float pp[PRE == 2 ? 4 * RT * 64 * 4 : 2 * 16 * LDP];
    __shared__ __attribute__((aligned(16))) float red[2 * 64];
    __shared__ __attribute__((aligned(16))) float xs[POST_GEMM ? 16 * LDX : 4];
    static_assert(4 * RT * 64 * 4 >= 2 * 16 * LDP, "pp must hold the two gradient tiles");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4, ql = c & 3;
    const int t0 = blockIdx.x * 16, T = a.T;
    const bool tok = t0 + c < T;
    const size_t trow = (size_t)min(t0 + c, T - 1);
    const DropCtx dc = make_drop(a.rng, a.add, a.site, a.p, a.train);
    int mt[2], f0[2], f0c[2];
    bool fok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        mt[i] = min(w + 4 * i, RT - 1);
        f0[i] = 16 * mt[i] + 4 * g;
        fok[i] = (w + 4 * i < RT) && f0[i] < RE;
        f0c[i] = min(f0[i], RE - 4);
    }

    // ---------------- every independent load, up front ----------------
    Frag<RE> wpost[2];
    if constexpr (POST_GEMM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) wpost[i].load(a.post_wt + (size_t)min(16 * mt[i] + c, RE - 1) * RE, g);
    }
    float4 xh4[2], d4[2], gam[2];
    const float rs = a.rstd[trow];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        xh4[i] = f4(a.xhat + trow * RE + f0c[i]);
        gam[i] = f4(a.gamma + f0c[i]);
        d4[i] = a.addend ? f4(a.addend + trow * RE + f0c[i]) : zero4();
    }
    if constexpr (PRE == 2) {
        // K split over the waves: wave w takes k groups w, w + 4, ... of the 19 for ALL 7 feature tiles
        float4 xq[NQW], wq[RT][NQW];
#pragma unroll
        for (int u = 0; u < NQW; ++u) {
            const int q = w + 4 * u, col = 16 * q + 4 * g;
            const bool ok = q < KQ3 && col < K3;
            const int colc = min(col, K3 - 4);
            const float4 tx = f4(a.pre_a + trow * K3 + colc);
            xq[u] = ok ? tx : zero4();
#pragma unroll
            for (int m = 0; m < RT; ++m) {
                const float4 tw = f4(a.pre_wt + (size_t)min(16 * m + c, RE - 1) * K3 + colc);
                wq[m][u] = ok ? tw : zero4();
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // all 40 operand loads in flight before the first MFMA waits for one
        floatx4 acc[RT];
#pragma unroll
        for (int m = 0; m < RT; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NQW; ++u) {
#pragma unroll
            for (int m = 0; m < RT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[m][u].x, xq[u].x, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[m][u].y, xq[u].y, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[m][u].z, xq[u].z, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[m][u].w, xq[u].w, acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < RT; ++m)
            *reinterpret_cast<float4*>(pp + ((w * RT + m) * 64 + lane) * 4) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {                 // partial tiles in wave order
                const float4 v = f4(pp + ((ww * RT + mt[i]) * 64 + lane) * 4);
                d4[i].x += v.x; d4[i].y += v.y; d4[i].z += v.z; d4[i].w += v.w;
            }
    } else {
        const int ns = PRE == 1 ? a.nslab : 1;
        constexpr int NB = PRE == 1 ? 8 : 1;                // slabs in flight together; added in slab order
        for (int s0 = 0; s0 < ns; s0 += NB) {
            float4 v[NB][2];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float* ds = a.d_out + (size_t)min(s0 + j, ns - 1) * a.slab_stride + trow * RE;
#pragma unroll
                for (int i = 0; i < 2; ++i) v[j][i] = f4(ds + f0c[i]);
            }
            // (hipcc otherwise sinks the later slabs' loads to their adds — one register set re-used, one global round trip
            // per slab: the ISA of rc_bwd_kernel<1, true> had four `global_load, s_waitcnt vmcnt(0)` pairs in a row.  The
            // empty asm is a fence both ways: its memory clobber keeps every load above it, and the adds below take their
            // 0 / 1 factor from its output, so none of them can move up between the loads.)
            float one = 1.f;
            if (NB > 1) asm volatile("" : "+v"(one) : : "memory");
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float m = (s0 + j < ns) ? one : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    d4[i].x += m * v[j][i].x; d4[i].y += m * v[j][i].y; d4[i].z += m * v[j][i].z; d4[i].w += m * v[j][i].w;
                }
            }
        }
    }

    // g = d * gamma; dz = rstd * (g - mean(g) - xhat * mean(g * xhat))
    float dd[2][4], xh[2][4], gv[2][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float dr[4] = {d4[i].x, d4[i].y, d4[i].z, d4[i].w}, hr[4] = {xh4[i].x, xh4[i].y, xh4[i].z, xh4[i].w};
        const float gm[4] = {gam[i].x, gam[i].y, gam[i].z, gam[i].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = fok[i] && tok;
            dd[i][r] = ok ? dr[r] : 0.f;
            xh[i][r] = ok ? hr[r] : 0.f;
            gv[i][r] = dd[i][r] * gm[r];
            s1 += gv[i][r];
            s2 += gv[i][r] * xh[i][r];
        }
    }
    if (PRE == 2) __syncthreads();                           // pp is about to be reused for the gradient tiles
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (g == 0) { red[w * 16 + c] = s1; red[64 + w * 16 + c] = s2; }
    // LayerNorm weight / bias gradient terms of this lane's elements -> LDS tiles [16 tokens][112]
    if (a.gpart) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (w + 4 * i < RT) {
                *reinterpret_cast<float4*>(pp + c * LDP + f0[i]) =
                    make_float4(dd[i][0] * xh[i][0], dd[i][1] * xh[i][1], dd[i][2] * xh[i][2], dd[i][3] * xh[i][3]);
                *reinterpret_cast<float4*>(pp + 16 * LDP + c * LDP + f0[i]) = make_float4(dd[i][0], dd[i][1], dd[i][2], dd[i][3]);
            }
    }
    __syncthreads();
    const float invE = 1.0f / (float)RE;
    const float c1 = (((red[c] + red[16 + c]) + red[32 + c]) + red[48 + c]) * invE;
    const float c2 = (((red[64 + c] + red[80 + c]) + red[96 + c]) + red[112 + c]) * invE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float mult[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.dy) drop_mult_quad(dc, (uint32_t)(t0 / 4 + (c >> 2)), f0c[i], ql, mult);
        float v[4], vy[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = rs * (gv[i][r] - c1 - xh[i][r] * c2);
            vy[r] = v[r] * mult[r];
        }
        if (fok[i] && tok) {
            const size_t off = trow * RE + f0[i];
            *reinterpret_cast<float4*>(a.dz + off) = make_float4(v[0], v[1], v[2], v[3]);
            if (a.dy) *reinterpret_cast<float4*>(a.dy + off) = make_float4(vy[0], vy[1], vy[2], vy[3]);
        }
        if constexpr (POST_GEMM) {
            if (w + 4 * i < RT)
                *reinterpret_cast<float4*>(xs + c * LDX + f0[i]) = (fok[i] && tok) ? make_float4(vy[0], vy[1], vy[2], vy[3]) : zero4();
        }
    }
    // column sums over the 16 tokens, in token order: this workgroup's partial row of the parameter gradients
    if (a.gpart && tid < RE) {
        float sw = 0.f, sb = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sw += pp[r * LDP + tid];
            sb += pp[16 * LDP + r * LDP + tid];
        }
        a.gpart[((size_t)blockIdx.x * 2 + 0) * RE + tid] = sw;
        a.gpart[((size_t)blockIdx.x * 2 + 1) * RE + tid] = sb;
    }
    if constexpr (POST_GEMM) {
        __syncthreads();
        Frag<RE> af;
        af.load(xs + c * LDX, g);
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        mma<RE, 2>(acc, wpost, af);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (fok[i] && tok)
                *reinterpret_cast<float4*>(a.post_out + trow * RE + f0[i]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
}

// transposed copies of the in-proj [3E x E] and out-proj [E x E] weights of `nl` consecutive layers (backward runs the
// dgrad products in the same K-contiguous-row form as the forward): wt[l] = { in_w^T [E x 3E] | out_w^T [E x E] }
__global__ __launch_bounds__(256) void rc_pack_kernel(const float* __restrict__ params, long layer_stride, long off_in, long off_out,
                                                      float* __restrict__ wt) {
    constexpr int N1 = 3 * RE * RE, N2 = RE * RE;
    const int l = blockIdx.y;
    const float* P = params + (size_t)l * layer_stride;
    float* o = wt + (size_t)l * (N1 + N2);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N1 + N2; i += gridDim.x * 256) {
        if (i < N1) {
            const int n = i / (3 * RE), k = i - n * 3 * RE;          // out[n][k] = in_w[k][n]
            o[i] = P[off_in + (size_t)k * RE + n];
        } else {
            const int j = i - N1, n = j / RE, k = j - n * RE;        // out[n][k] = out_w[k][n]
            o[i] = P[off_out + (size_t)k * RE + n];
        }
    }
}

}  // namespace

bool rc_supported(int E) { return E == RE; }
long rc_pack_floats() { return (long)3 * RE * RE + (long)RE * RE; }
int rc_blocks(int T) { return (T + 15) / 16; }

int launch_rc_pack(const float* params, long layer_stride, long off_in, long off_out, float* wt, int nl, hipStream_t st) {
    GF_CHECK_ARG(params && wt && nl >= 1 && aligned16(wt), "rc_pack: bad arguments");
    hipLaunchKernelGGL(rc_pack_kernel, dim3(40, nl), dim3(256), 0, st, params, layer_stride, off_in, off_out, wt);
    GF_LAUNCH_CHECK();
    return 0;
}

// forward A: x1 = LayerNorm1(x + dropout(attn_o . Wo^T + bo))
int launch_rc_outproj_ln_fwd(const float* attn_o, const float* wo, const float* bo, const float* x, const float* gamma,
                             const float* beta, float* out, float* xhat, float* rstd, int T, float eps, float p, uint32_t site,
                             const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_CHECK_ARG(attn_o && wo && bo && x && gamma && beta && out && T > 0, "rc_outproj_ln_fwd: bad arguments");
    GF_CHECK_ARG(aligned16(attn_o) && aligned16(wo) && aligned16(bo) && aligned16(x) && aligned16(gamma) && aligned16(beta) &&
                     aligned16(out) && (!xhat || aligned16(xhat)), "rc_outproj_ln_fwd: operands must be 16-byte aligned");
    RcFwdArgs a{};
    a.pre_a = attn_o; a.pre_w = wo; a.pre_b = bo; a.x = x; a.gamma = gamma; a.beta = beta; a.out = out; a.xhat = xhat; a.rstd = rstd;
    a.T = T; a.eps = eps; a.p = p; a.site = site; a.rng = rng; a.add = add; a.train = train;
    hipLaunchKernelGGL((rc_fwd_kernel<1, false>), dim3(rc_blocks(T)), dim3(256), 0, st, a);
    GF_LAUNCH_CHECK();
    return 0;
}

// forward B: out = LayerNorm2(x + dropout(sum of slabs)); with w_in: qkv = out . w_in^T + b_in (the next layer's in-proj)
int launch_rc_ln_inproj_fwd(const float* y, int nslab, long slab_stride, const float* x, const float* gamma, const float* beta,
                            float* out, float* xhat, float* rstd, const float* w_in, const float* b_in, float* qkv, int T,
                            float eps, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_CHECK_ARG(y && x && gamma && beta && out && T > 0 && nslab >= 1, "rc_ln_inproj_fwd: bad arguments");
    GF_CHECK_ARG(aligned16(y) && (slab_stride & 3) == 0 && aligned16(x) && aligned16(gamma) && aligned16(beta) && aligned16(out) &&
                     (!xhat || aligned16(xhat)), "rc_ln_inproj_fwd: operands must be 16-byte aligned");
    GF_CHECK_ARG(!w_in || (b_in && qkv && aligned16(w_in) && aligned16(b_in) && aligned16(qkv)),
                 "rc_ln_inproj_fwd: in-proj needs bias, output and 16-byte aligned operands");
    RcFwdArgs a{};
    a.y = y; a.nslab = nslab; a.slab_stride = slab_stride; a.x = x; a.gamma = gamma; a.beta = beta; a.out = out; a.xhat = xhat;
    a.rstd = rstd; a.post_w = w_in; a.post_b = b_in; a.post_out = qkv;
    a.T = T; a.eps = eps; a.p = p; a.site = site; a.rng = rng; a.add = add; a.train = train;
    if (w_in) hipLaunchKernelGGL((rc_fwd_kernel<0, true>), dim3(rc_blocks(T)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((rc_fwd_kernel<0, false>), dim3(rc_blocks(T)), dim3(256), 0, st, a);
    GF_LAUNCH_CHECK();
    return 0;
}

// forward 0: x0 = dropout(x_in + pe[s]) and qkv of layer 0 = x0 . w_in^T + b_in — the head of an encoder stack as one launch
int launch_rc_pe_inproj_fwd(const float* x_in, const float* pe, float* out, const float* w_in, const float* b_in, float* qkv, int T,
                            int B, float p, const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_CHECK_ARG(x_in && pe && out && w_in && b_in && qkv && T > 0 && B > 0, "rc_pe_inproj_fwd: bad arguments");
    GF_CHECK_ARG(aligned16(x_in) && aligned16(pe) && aligned16(out) && aligned16(w_in) && aligned16(b_in) && aligned16(qkv),
                 "rc_pe_inproj_fwd: operands must be 16-byte aligned");
    RcFwdArgs a{};
    a.y = pe; a.x = x_in; a.out = out; a.post_w = w_in; a.post_b = b_in; a.post_out = qkv;
    a.T = T; a.B = B; a.p = p; a.site = SITE_PE; a.rng = rng; a.add = add; a.train = train;
    hipLaunchKernelGGL((rc_fwd_kernel<2, true>), dim3(rc_blocks(T)), dim3(256), 0, st, a);
    GF_LAUNCH_CHECK();
    return 0;
}

// backward of a LayerNorm with its surroundings.  Incoming gradient: d_qkv . w_in_t^T (w_in_t = in-proj weight transposed,
// [E x 3E]) when d_qkv is given, else the nslab slabs of d_out; + addend.  Outgoing: dz, dy = dz * dropout multiplier, the
// per-workgroup partial sums of the LayerNorm parameter gradients (gpart: rc_blocks(T) * 2 * E floats), and with wo_t
// (out-proj weight transposed) d_attn = dy . wo_t^T.
int launch_rc_ln_bwd(const float* d_qkv, const float* w_in_t, const float* d_out, int nslab, long slab_stride, const float* addend,
                     const float* xhat, const float* rstd, const float* gamma, float* dz, float* dy, float* gpart,
                     const float* wo_t, float* d_attn, int T, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train,
                     hipStream_t st) {
    GF_CHECK_ARG((d_qkv || d_out) && xhat && rstd && gamma && dz && T > 0, "rc_ln_bwd: bad arguments");
    GF_CHECK_ARG(!d_qkv || (w_in_t && aligned16(d_qkv) && aligned16(w_in_t)), "rc_ln_bwd: in-proj dgrad needs the transposed weight");
    GF_CHECK_ARG(!wo_t || (d_attn && dy && aligned16(wo_t) && aligned16(d_attn)), "rc_ln_bwd: out-proj dgrad needs dy and an output");
    GF_CHECK_ARG((!d_out || (aligned16(d_out) && (slab_stride & 3) == 0)) && (!addend || aligned16(addend)) && aligned16(xhat) &&
                     aligned16(gamma) && aligned16(dz) && (!dy || aligned16(dy)), "rc_ln_bwd: operands must be 16-byte aligned");
    RcBwdArgs a{};
    a.pre_a = d_qkv; a.pre_wt = w_in_t; a.d_out = d_out; a.nslab = nslab; a.slab_stride = slab_stride; a.addend = addend;
    a.xhat = xhat; a.rstd = rstd; a.gamma = gamma; a.dz = dz; a.dy = dy; a.gpart = gpart; a.post_wt = wo_t; a.post_out = d_attn;
    a.T = T; a.p = p; a.site = site; a.rng = rng; a.add = add; a.train = train;
    const dim3 grid(rc_blocks(T)), blk(256);
    if (d_qkv) {
        if (wo_t) hipLaunchKernelGGL((rc_bwd_kernel<2, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((rc_bwd_kernel<2, false>), grid, blk, 0, st, a);
    } else if (nslab > 1) {
        if (wo_t) hipLaunchKernelGGL((rc_bwd_kernel<1, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((rc_bwd_kernel<1, false>), grid, blk, 0, st, a);
    } else {
        if (wo_t) hipLaunchKernelGGL((rc_bwd_kernel<0, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((rc_bwd_kernel<0, false>), grid, blk, 0, st, a);
    }
    GF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganffn
