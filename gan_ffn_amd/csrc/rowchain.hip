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
// token rows and runs the whole chain for them: the small GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32) with the
// weights read straight from L2 into the B-operand layout, the LayerNorm statistics by a 16-lane reduce + one LDS
// exchange between the 4 waves, the row tile handed from the LayerNorm stage to the following GEMM through LDS.
// 4 of the 9 (forward) / 9 (backward) launches of a layer disappear.
//
// MFMA layout (v_mfma_f32_16x16x4_f32; lane = 16 g + c):  A: lane holds A[row c][k = g], B: lane holds B[k = g][col c],
// D register r of the lane = D[row 4 g + r][col c].  Rows = the 16 tokens of the workgroup, columns = 16 output
// features of one column tile: a lane owns 4 CONSECUTIVE TOKENS of one column — exactly one Philox call (common.h
// drop_mult4: 4 consecutive rows of a column), and the layout of the 32x32 GEMM epilogues elsewhere.
// k order: lane group g takes k = 16 q + 4 g + j for the j-th MFMA of group q (one 16-byte load per operand and group);
// the same permutation on A and B, and the same for every row, so a dialogue's bits do not depend on its position in
// the batch (tests/test_hip_properties.py::test_full_size_batch_permutation...).
// The 4 waves split the column tiles (wave w owns tiles w, w + 4, ...); a tile beyond the last one is computed on a
// clamped copy and discarded, so no MFMA sits under a run-time condition.
// Deterministic: no atomics; the LayerNorm parameter gradients leave as per-workgroup partial rows that
// ln_param_reduce_kernel (elementwise.hip) adds in block order.
#include "common.h"

#pragma clang fp contract(off)

namespace ganffn {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RE = 100;              // d_model handled here
constexpr int RT = (RE + 15) / 16;   // 7 column tiles of 16 over a 100-wide row
constexpr int LDX = 116;             // LDS row stride of the 16 x 100 row tile handed to the trailing GEMM (zero-padded to 112)

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// one lane's A or B fragments for K (a multiple of 4) from a K-contiguous row: group q = floats 16 q + 4 g .. + 3
template <int K>
struct Frag {
    static constexpr int KQ = (K + 15) / 16;
    float4 v[KQ];
    __device__ __forceinline__ void load(const float* __restrict__ row, int g) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int col = 16 * q + 4 * g;
            const float4 t = *reinterpret_cast<const float4*>(row + min(col, K - 4));
            v[q] = (16 * q + 12 < K || col < K) ? t : zero4();     // compile-time true except in the last group
        }
    }
};

// acc[i] += (this workgroup's 16 rows) x (W rows 16 nt_i + c)^T over K; wrow[i] = the lane's row of W for tile i.
// B fragments are loaded CH groups at a time for all NI tiles, then the MFMAs of those groups run interleaved over the
// tiles (independent accumulator chains).
template <int K, int NI>
__device__ __forceinline__ void gemm16(floatx4 (&acc)[NI], const Frag<K>& a, const float* const* wrow, int g) {
    constexpr int KQ = Frag<K>::KQ;
    constexpr int CH = (KQ * NI <= 16) ? KQ : (NI >= 3 ? 4 : 5);
#pragma unroll
    for (int q0 = 0; q0 < KQ; q0 += CH) {
        float4 b[NI][CH];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int q = q0 + u;
                if (q < KQ) {
                    const int col = 16 * q + 4 * g;
                    const float4 t = *reinterpret_cast<const float4*>(wrow[i] + min(col, K - 4));
                    b[i][u] = (16 * q + 12 < K || col < K) ? t : zero4();
                } else {
                    b[i][u] = zero4();
                }
            }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (q0 + u < KQ) {
                const float4 av = a.v[q0 + u];
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[i][u].x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[i][u].y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[i][u].z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[i][u].w, acc[i], 0, 0, 0);
            }
        }
    }
}

// sum over the 16 lanes of a lane group (same g): the 16 columns of a tile
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// per-token sums of the 4 waves' partial row sums: part[r] (token 4 g + r, this wave's columns) -> the full row sum
// (fixed order w = 0..3).  red: [4 waves][16 tokens], one buffer per exchange of a kernel (no barrier needed before the
// write: a buffer is written once); one barrier inside.
__device__ __forceinline__ void rows_allreduce(float (&part)[4], float* __restrict__ red, int w, int c, int g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) part[r] = group16_sum(part[r]);
    if (c == 0) *reinterpret_cast<float4*>(red + w * 16 + 4 * g) = make_float4(part[0], part[1], part[2], part[3]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
        part[r] = ((red[0 * 16 + 4 * g + r] + red[1 * 16 + 4 * g + r]) + red[2 * 16 + 4 * g + r]) + red[3 * 16 + 4 * g + r];
}
// two row sums in one exchange (red: [2][4 waves][16 tokens])
__device__ __forceinline__ void rows_allreduce2(float (&pa)[4], float (&pb)[4], float* __restrict__ red, int w, int c, int g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { pa[r] = group16_sum(pa[r]); pb[r] = group16_sum(pb[r]); }
    if (c == 0) {
        *reinterpret_cast<float4*>(red + w * 16 + 4 * g) = make_float4(pa[0], pa[1], pa[2], pa[3]);
        *reinterpret_cast<float4*>(red + 64 + w * 16 + 4 * g) = make_float4(pb[0], pb[1], pb[2], pb[3]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pa[r] = ((red[0 * 16 + 4 * g + r] + red[1 * 16 + 4 * g + r]) + red[2 * 16 + 4 * g + r]) + red[3 * 16 + 4 * g + r];
        pb[r] = ((red[64 + 0 * 16 + 4 * g + r] + red[64 + 1 * 16 + 4 * g + r]) + red[64 + 2 * 16 + 4 * g + r]) + red[64 + 3 * 16 + 4 * g + r];
    }
}

struct RcFwdArgs {
    // leading stage: y = pre_a . pre_w^T + pre_b (GEMM)  or  y = sum of nslab partial slabs
    const float* pre_a; const float* pre_w; const float* pre_b;
    const float* y; int nslab; long slab_stride;
    const float* x;                  // residual input [T x E]
    const float* gamma; const float* beta;
    float* out; float* xhat; float* rstd;     // xhat / rstd may be null (nothing kept for backward)
    // trailing stage (optional): post_out [T x 3E] = out . post_w^T + post_b
    const float* post_w; const float* post_b; float* post_out;
    int T; float eps, p; uint32_t site; const uint64_t* rng; uint64_t add; int train;
};

// z = x + dropout(y); out = LayerNorm(z); optionally the next layer's in-proj on the fresh rows
template <bool PRE_GEMM, bool POST_GEMM>
__global__ __launch_bounds__(256) void rc_fwd_kernel(RcFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float red[2 * 4 * 16];
    __shared__ __attribute__((aligned(16))) float xs[POST_GEMM ? 16 * LDX : 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    const int t0 = blockIdx.x * 16;
    const int T = a.T;
    const DropCtx dc = make_drop(a.rng, a.add, a.site, a.p, a.train);

    // this wave's column tiles: w and w + 4 (tile 7 does not exist: wave 3 recomputes tile 6 and discards it)
    int nt[2], col[2], colc[2];
    bool cok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        nt[i] = w + 4 * i;
        const bool tile_ok = nt[i] < RT;
        col[i] = 16 * min(nt[i], RT - 1) + c;
        cok[i] = tile_ok && col[i] < RE;
        colc[i] = min(col[i], RE - 1);
    }
    size_t roff[4];                      // clamped row offsets (in rows) of this lane's 4 tokens
    bool rok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rok[r] = t0 + 4 * g + r < T;
        roff[r] = (size_t)min(t0 + 4 * g + r, T - 1);
    }

    // ---------------- leading stage ----------------
    float y[2][4];
    if constexpr (PRE_GEMM) {
        Frag<RE> af;
        af.load(a.pre_a + (size_t)min(t0 + c, T - 1) * RE, g);
        const float* const wrow[2] = {a.pre_w + (size_t)colc[0] * RE, a.pre_w + (size_t)colc[1] * RE};
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        gemm16<RE, 2>(acc, af, wrow, g);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float bias = a.pre_b[colc[i]];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[i][r] = acc[i][r] + bias;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[i][r] = 0.f;
        for (int s0 = 0; s0 < a.nslab; s0 += 4) {           // 4 slabs' loads in flight at a time, added in slab order
            float v[4][2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* ys = a.y + (size_t)min(s0 + j, a.nslab - 1) * a.slab_stride;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[j][i][r] = ys[roff[r] * RE + colc[i]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m = (s0 + j < a.nslab) ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[i][r] = __fmaf_rn(m, v[j][i][r], y[i][r]);
            }
        }
    }

    // ---------------- residual + dropout + LayerNorm ----------------
    float z[2][4], gam[2], bet[2];
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        gam[i] = a.gamma[colc[i]];
        bet[i] = a.beta[colc[i]];
        float xv[4], mult[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = a.x[roff[r] * RE + colc[i]];
        drop_mult4(dc, (uint32_t)(t0 / 4 + g), (uint32_t)RE, (uint32_t)colc[i], mult);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[i][r] = cok[i] ? __fmaf_rn(y[i][r], mult[r], xv[r]) : 0.f;
            sum[r] += z[i][r];
        }
    }
    rows_allreduce(sum, red, w, c, g);
    const float invE = 1.0f / (float)RE;
    float mean[4], var[4] = {0.f, 0.f, 0.f, 0.f}, rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mean[r] = sum[r] * invE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = cok[i] ? __fsub_rn(z[i][r], mean[r]) : 0.f;
            var[r] = __fmaf_rn(d, d, var[r]);
        }
    rows_allreduce(var, red + 64, w, c, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) rs[r] = rsqrtf(var[r] * invE + a.eps);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = __fmul_rn(__fsub_rn(z[i][r], mean[r]), rs[r]);
            const float o = __fmaf_rn(xh, gam[i], bet[i]);
            if (cok[i] && rok[r]) {
                const size_t off = roff[r] * RE + col[i];
                if (a.xhat) a.xhat[off] = xh;
                a.out[off] = o;
            }
            if constexpr (POST_GEMM) {
                // row tile -> LDS for the trailing GEMM; columns 100 .. 111 zero (tile 6's lanes c >= 4)
                if (nt[i] < RT) xs[(4 * g + r) * LDX + 16 * nt[i] + c] = cok[i] ? o : 0.f;
            }
        }
    if (a.rstd && w == 0 && c == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rok[r]) a.rstd[t0 + 4 * g + r] = rs[r];
    }

    // ---------------- trailing stage: in-proj of the next layer ----------------
    if constexpr (POST_GEMM) {
        constexpr int NP = 3 * RE, PT = (NP + 15) / 16;     // 300 output features, 19 tiles
        __syncthreads();
        Frag<RE> af;                                        // rows beyond 100 columns read the zero padding
#pragma unroll
        for (int q = 0; q < Frag<RE>::KQ; ++q) af.v[q] = *reinterpret_cast<const float4*>(xs + c * LDX + 16 * q + 4 * g);
        // wave w: tiles w, w + 4, w + 8 then w + 12, w + 16 (clamped duplicates beyond tile 18 are discarded)
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            constexpr int NI0 = 3;
            const int ni = grp == 0 ? 3 : 2;
            int pn[NI0];
            const float* wrow[NI0];
            floatx4 acc[NI0];
#pragma unroll
            for (int i = 0; i < NI0; ++i) {
                const int tile = w + 4 * (3 * grp + i);
                pn[i] = (i < ni && tile < PT) ? 16 * tile + c : -1;
                wrow[i] = a.post_w + (size_t)min(max(pn[i], 0), NP - 1) * RE;
                acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
            gemm16<RE, NI0>(acc, af, wrow, g);
#pragma unroll
            for (int i = 0; i < NI0; ++i) {
                const bool ok = pn[i] >= 0 && pn[i] < NP;
                const float bias = a.post_b[min(max(pn[i], 0), NP - 1)];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ok && rok[r]) a.post_out[roff[r] * NP + pn[i]] = acc[i][r] + bias;
            }
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
    __shared__ __attribute__((aligned(16))) float red[2 * 4 * 16];
    __shared__ __attribute__((aligned(16))) float xs[POST_GEMM ? 16 * LDX : 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    const int t0 = blockIdx.x * 16;
    const int T = a.T;
    const DropCtx dc = make_drop(a.rng, a.add, a.site, a.p, a.train);
    int nt[2], col[2], colc[2];
    bool cok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        nt[i] = w + 4 * i;
        col[i] = 16 * min(nt[i], RT - 1) + c;
        cok[i] = nt[i] < RT && col[i] < RE;
        colc[i] = min(col[i], RE - 1);
    }
    size_t roff[4];
    bool rok[4];
    float rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rok[r] = t0 + 4 * g + r < T;
        roff[r] = (size_t)min(t0 + 4 * g + r, T - 1);
        rs[r] = a.rstd[roff[r]];
    }

    float d[2][4], xh[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[i][r] = a.xhat[roff[r] * RE + colc[i]];
            d[i][r] = a.addend ? a.addend[roff[r] * RE + colc[i]] : 0.f;
        }
    if constexpr (PRE == 2) {
        constexpr int K3 = 3 * RE;
        Frag<K3> af;
        af.load(a.pre_a + (size_t)min(t0 + c, T - 1) * K3, g);
        const float* const wrow[2] = {a.pre_wt + (size_t)colc[0] * K3, a.pre_wt + (size_t)colc[1] * K3};
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        gemm16<K3, 2>(acc, af, wrow, g);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) d[i][r] += acc[i][r];
    } else {
        const int ns = PRE == 1 ? a.nslab : 1;
        constexpr int NB = PRE == 1 ? 4 : 1;                // slabs in flight together; added in slab order
        for (int s0 = 0; s0 < ns; s0 += NB) {
            float v[NB][2][4];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float* ds = a.d_out + (size_t)min(s0 + j, ns - 1) * a.slab_stride;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[j][i][r] = ds[roff[r] * RE + colc[i]];
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float m = (s0 + j < ns) ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[i][r] += m * v[j][i][r];
            }
        }
    }

    // g = d * gamma; dz = rstd * (g - mean(g) - xhat * mean(g * xhat))
    float gv[2][4], aw[2], ab[2];
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float gam = a.gamma[colc[i]];
        aw[i] = 0.f; ab[i] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = cok[i] && rok[r];
            const float dd = ok ? d[i][r] : 0.f, h = ok ? xh[i][r] : 0.f;
            xh[i][r] = h;
            aw[i] += dd * h;
            ab[i] += dd;
            gv[i][r] = dd * gam;
            s1[r] += gv[i][r];
            s2[r] += gv[i][r] * h;
        }
    }
    rows_allreduce2(s1, s2, red, w, c, g);
    const float invE = 1.0f / (float)RE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float mult[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.dy) drop_mult4(dc, (uint32_t)(t0 / 4 + g), (uint32_t)RE, (uint32_t)colc[i], mult);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = rs[r] * (gv[i][r] - s1[r] * invE - xh[i][r] * (s2[r] * invE));
            const float vy = v * mult[r];
            if (cok[i] && rok[r]) {
                const size_t off = roff[r] * RE + col[i];
                a.dz[off] = v;
                if (a.dy) a.dy[off] = vy;
            }
            if constexpr (POST_GEMM) {
                if (nt[i] < RT) xs[(4 * g + r) * LDX + 16 * nt[i] + c] = (cok[i] && rok[r]) ? vy : 0.f;
            }
        }
    }
    // LayerNorm weight / bias gradient: this workgroup's 16-token partial sums, one row each
    if (a.gpart) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float sw = aw[i], sb = ab[i];
            sw += __shfl_xor(sw, 16, 64); sb += __shfl_xor(sb, 16, 64);
            sw += __shfl_xor(sw, 32, 64); sb += __shfl_xor(sb, 32, 64);
            if (g == 0 && cok[i]) {
                a.gpart[((size_t)blockIdx.x * 2 + 0) * RE + col[i]] = sw;
                a.gpart[((size_t)blockIdx.x * 2 + 1) * RE + col[i]] = sb;
            }
        }
    }
    if constexpr (POST_GEMM) {
        __syncthreads();
        Frag<RE> af;
#pragma unroll
        for (int q = 0; q < Frag<RE>::KQ; ++q) af.v[q] = *reinterpret_cast<const float4*>(xs + c * LDX + 16 * q + 4 * g);
        const float* const wrow[2] = {a.post_wt + (size_t)colc[0] * RE, a.post_wt + (size_t)colc[1] * RE};
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        gemm16<RE, 2>(acc, af, wrow, g);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (cok[i] && rok[r]) a.post_out[roff[r] * RE + col[i]] = acc[i][r];
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
    GF_CHECK_ARG(aligned16(attn_o) && aligned16(wo), "rc_outproj_ln_fwd: operands must be 16-byte aligned");
    RcFwdArgs a{};
    a.pre_a = attn_o; a.pre_w = wo; a.pre_b = bo; a.x = x; a.gamma = gamma; a.beta = beta; a.out = out; a.xhat = xhat; a.rstd = rstd;
    a.T = T; a.eps = eps; a.p = p; a.site = site; a.rng = rng; a.add = add; a.train = train;
    hipLaunchKernelGGL((rc_fwd_kernel<true, false>), dim3(rc_blocks(T)), dim3(256), 0, st, a);
    GF_LAUNCH_CHECK();
    return 0;
}

// forward B: out = LayerNorm2(x + dropout(sum of slabs)); with w_in: qkv = out . w_in^T + b_in (the next layer's in-proj)
int launch_rc_ln_inproj_fwd(const float* y, int nslab, long slab_stride, const float* x, const float* gamma, const float* beta,
                            float* out, float* xhat, float* rstd, const float* w_in, const float* b_in, float* qkv, int T,
                            float eps, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_CHECK_ARG(y && x && gamma && beta && out && T > 0 && nslab >= 1, "rc_ln_inproj_fwd: bad arguments");
    GF_CHECK_ARG(!w_in || (b_in && qkv && aligned16(w_in)), "rc_ln_inproj_fwd: in-proj needs bias, output and a 16-byte aligned weight");
    RcFwdArgs a{};
    a.y = y; a.nslab = nslab; a.slab_stride = slab_stride; a.x = x; a.gamma = gamma; a.beta = beta; a.out = out; a.xhat = xhat;
    a.rstd = rstd; a.post_w = w_in; a.post_b = b_in; a.post_out = qkv;
    a.T = T; a.eps = eps; a.p = p; a.site = site; a.rng = rng; a.add = add; a.train = train;
    if (w_in) hipLaunchKernelGGL((rc_fwd_kernel<false, true>), dim3(rc_blocks(T)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((rc_fwd_kernel<false, false>), dim3(rc_blocks(T)), dim3(256), 0, st, a);
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
    GF_CHECK_ARG(!wo_t || (d_attn && dy && aligned16(wo_t)), "rc_ln_bwd: out-proj dgrad needs dy and an output");
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
