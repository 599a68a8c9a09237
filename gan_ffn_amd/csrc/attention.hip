// attention.hip — per-(dialogue, head) self-attention core, forward and backward, fp32 MFMA over LDS.
//
// Replaces the attention core of torch's nn.MultiheadAttention as used by the reference's
// nn.TransformerEncoderLayer stacks (/root/reference/model.py:1210,1244,1276,1307,1340,1377):
//   P = softmax(q k^T / sqrt(hd)) over keys, NO masks (padded utterances attend and are attended to),
//   dropout(0.1) on P in train mode, O = P v.
// The sequence is one dialogue (S <= 110 utterances), so one workgroup owns one (b, h) problem with
// q, k, v and the S x S probabilities resident in LDS; nothing S x S ever touches HBM, and the
// backward recomputes P (and the Philox dropout mask) instead of saving it.
//
// Layout: qkv [T x 3E] packed q|k|v per token (t = s*B + b), head h = columns h*hd .. h*hd+hd-1.
// All small products run on v_mfma_f32_32x32x2_f32 with both operands read from LDS by ds_read_b32;
// every LDS matrix has an ODD row stride, which makes row-indexed and column-indexed fragment reads
// bank-conflict-free alike.  4 waves; wave w owns query (or key) rows 32w .. 32w+31.
#include "common.h"

namespace ganffn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// acc[t] += sum_k A(m, k) * B(k, n_t),  m = lane&31 of this wave's row block, n_t = 32*t + lane&31.
// A(m,k) at A[m*sam + k*sak], B(k,n) at Bp[n*sbn + k*sbk]; K = 2*ksteps.
template <int NT>
__device__ __forceinline__ void mma_lds(floatx16 (&acc)[NT], int ntiles, const float* __restrict__ A, int sam, int sak,
                                        const float* __restrict__ Bp, int sbn, int sbk, int ksteps, int r, int h) {
    const float* pa = A + r * sam + h * sak;
    const float* pb = Bp + r * sbn + h * sbk;
    for (int ks = 0; ks < ksteps; ++ks) {
        const float a = pa[2 * ks * sak];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < ntiles) {
                const float b = pb[t * 32 * sbn + 2 * ks * sbk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ float half_max(float v) {  // reduce over the 32 lanes of a wave half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct AttnGeom {
    int S, B, E, H, hd;
    int NTr;  // 32-row tiles covering S
    int SK;   // S rounded up to even (K extent of P*V)
    int LDH;  // row stride of the [*, hd] matrices (odd)
    int LDP;  // row stride of the S x S matrix (odd)
    int ROWS; // rows allocated per [*, hd] matrix (NTr*32 + 1 spare row for over-reads)
};

__host__ __device__ inline AttnGeom make_geom(int S, int B, int E, int H) {
    AttnGeom g;
    g.S = S; g.B = B; g.E = E; g.H = H; g.hd = E / H;
    g.NTr = (S + 31) / 32;
    g.SK = (S + 1) & ~1;
    g.LDH = g.hd | 1;
    g.LDP = g.SK | 1;
    g.ROWS = g.NTr * 32 + 1;
    return g;
}
static inline size_t hd_mat_floats(const AttnGeom& g) { return (size_t)g.ROWS * g.LDH + 64; }
static inline size_t ss_mat_floats(const AttnGeom& g) { return (size_t)g.NTr * 32 * g.LDP + 64; }

// load one [S x hd] head slice of qkv (which = 0 q, 1 k, 2 v) into LDS, zero-padding rows >= S
__device__ __forceinline__ void load_head(float* __restrict__ dst, const float* __restrict__ src, int ld_src,
                                          const AttnGeom& g, int b, float scale, int tid) {
    const int hd2 = g.hd >> 1;
    const int total = g.NTr * 32 * hd2;
    for (int i = tid; i < total; i += 256) {
        const int s = i / hd2, d = (i - s * hd2) * 2;
        float2 v = make_float2(0.f, 0.f);
        if (s < g.S) v = *reinterpret_cast<const float2*>(src + (size_t)(s * g.B + b) * ld_src + d);
        dst[s * g.LDH + d] = v.x * scale;
        dst[s * g.LDH + d + 1] = v.y * scale;
    }
}

// scores (this wave's 32 query rows x all keys) -> probabilities in registers; returns keep bits
template <int NTC>
__device__ __forceinline__ void softmax_rows(floatx16 (&p)[NTC], int ntc, int S, int r, int h) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < NTC; ++c)
            if (c < ntc) {
                const int j = 32 * c + r;
                if (j >= S) p[c][i] = -INFINITY;
                m = fmaxf(m, p[c][i]);
            }
        m = half_max(m);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NTC; ++c)
            if (c < ntc) {
                const float e = expf(p[c][i] - m);  // exp(-inf) = 0 for padded keys
                p[c][i] = e;
                sum += e;
            }
        sum = half_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int c = 0; c < NTC; ++c)
            if (c < ntc) p[c][i] *= inv;
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                            AttnGeom g, float p, uint32_t site,
                                                            const uint64_t* __restrict__ rng, uint64_t add, int train) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / g.H, head = bh % g.H;
    const size_t HM = (size_t)g.ROWS * g.LDH + 64;
    float* Qs = smem;
    float* Ks = Qs + HM;
    float* Vs = Ks + HM;
    float* Ps = Vs + HM;
    const int ld3 = 3 * g.E;
    const float scale = rsqrtf((float)g.hd);

    load_head(Qs, qkv + head * g.hd, ld3, g, b, scale, tid);
    load_head(Ks, qkv + g.E + head * g.hd, ld3, g, b, 1.f, tid);
    load_head(Vs, qkv + 2 * g.E + head * g.hd, ld3, g, b, 1.f, tid);
    __syncthreads();

    const bool active = w < g.NTr;
    floatx16 pr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[c][i] = 0.f;

    if (active) {
        mma_lds<4>(pr, g.NTr, Qs + 32 * w * g.LDH, g.LDH, 1, Ks, g.LDH, 1, g.hd >> 1, r, h);
        softmax_rows<4>(pr, g.NTr, g.S, r, h);
        const DropCtx dc = make_drop(rng, add, site, p, train);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < g.NTr) {
                const int j = 32 * c + r;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float mult[4];
                    drop_mult4(dc, (uint32_t)(bh * 28 + 8 * w + 2 * gq + h), 128u, (uint32_t)j, mult);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 32 * w + 8 * gq + 4 * h + q;
                        if (j < g.SK) Ps[i * g.LDP + j] = (j < g.S) ? pr[c][gq * 4 + q] * mult[q] : 0.f;
                    }
                }
            }
        }
    }
    __syncthreads();

    if (active) {
        floatx16 oacc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[t][i] = 0.f;
        const int ntd = (g.hd + 31) / 32;
        mma_lds<2>(oacc, ntd, Ps + 32 * w * g.LDP, g.LDP, 1, Vs, 1, g.LDH, g.SK >> 1, r, h);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int d = 32 * t + r;
            if (t < ntd && d < g.hd) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int s = 32 * w + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (s < g.S) o[(size_t)(s * g.B + b) * g.E + head * g.hd + d] = oacc[t][i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward: d_qkv from d_o, recomputing P and the dropout mask
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                            float* __restrict__ d_qkv, AttnGeom g, float p,
                                                            uint32_t site, const uint64_t* __restrict__ rng,
                                                            uint64_t add, int train) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / g.H, head = bh % g.H;
    const size_t HM = (size_t)g.ROWS * g.LDH + 64;
    float* RA = smem;        // Q, later dO
    float* RB = RA + HM;     // K
    float* RC = RB + HM;     // V, later Q
    float* SS = RC + HM;     // Pdrop, later dS
    const int ld3 = 3 * g.E;
    const float scale = rsqrtf((float)g.hd);
    const bool active = w < g.NTr;
    const int ntd = (g.hd + 31) / 32;

    load_head(RA, qkv + head * g.hd, ld3, g, b, scale, tid);
    load_head(RB, qkv + g.E + head * g.hd, ld3, g, b, 1.f, tid);
    load_head(RC, qkv + 2 * g.E + head * g.hd, ld3, g, b, 1.f, tid);
    __syncthreads();

    floatx16 pr[4];   // P (pre-dropout probabilities), later dS
    floatx16 dp[4];   // dPdrop
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { pr[c][i] = 0.f; dp[c][i] = 0.f; }
    unsigned long long keep = ~0ull;  // bit (c*16 + i): element kept by dropout
    const DropCtx dc = make_drop(rng, add, site, p, train);

    if (active) {
        mma_lds<4>(pr, g.NTr, RA + 32 * w * g.LDH, g.LDH, 1, RB, g.LDH, 1, g.hd >> 1, r, h);
        softmax_rows<4>(pr, g.NTr, g.S, r, h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < g.NTr) {
                const int j = 32 * c + r;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float mult[4];
                    drop_mult4(dc, (uint32_t)(bh * 28 + 8 * w + 2 * gq + h), 128u, (uint32_t)j, mult);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 32 * w + 8 * gq + 4 * h + q;
                        if (mult[q] == 0.f) keep &= ~(1ull << (c * 16 + gq * 4 + q));
                        if (j < g.SK) SS[i * g.LDP + j] = (j < g.S) ? pr[c][gq * 4 + q] * mult[q] : 0.f;
                    }
                }
            }
        }
    }
    __syncthreads();                                    // everyone is done with Q; Pdrop is visible
    load_head(RA, d_o + head * g.hd, g.E, g, b, 1.f, tid);  // dO over Q
    __syncthreads();

    if (active) {
        // dPdrop = dO V^T
        mma_lds<4>(dp, g.NTr, RA + 32 * w * g.LDH, g.LDH, 1, RC, g.LDH, 1, g.hd >> 1, r, h);
        // D_i = sum_j dPdrop_ij * Pdrop_ij ;  dS_ij = P_ij * (mult_ij * dPdrop_ij - D_i)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < g.NTr) {
                    const float mm = ((keep >> (c * 16 + i)) & 1ull) ? dc.scale : 0.f;
                    dp[c][i] *= mm;
                    d += dp[c][i] * pr[c][i];
                }
            d = half_sum(d);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < g.NTr) pr[c][i] = pr[c][i] * (dp[c][i] - d);
        }
        // dV = Pdrop^T dO   (rows = keys 32w.., K = queries)
        floatx16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        mma_lds<2>(acc, ntd, SS + 32 * w, 1, g.LDP, RA, 1, g.LDH, g.NTr * 16, r, h);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int d = 32 * t + r;
            if (t < ntd && d < g.hd) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int s = 32 * w + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (s < g.S) d_qkv[(size_t)(s * g.B + b) * ld3 + 2 * g.E + head * g.hd + d] = acc[t][i];
                }
            }
        }
    }
    __syncthreads();                                    // Pdrop, dO and V are dead
    if (active) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < g.NTr) {
                const int j = 32 * c + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = 32 * w + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (j < g.SK) SS[row * g.LDP + j] = (row < g.S && j < g.S) ? pr[c][i] : 0.f;
                }
            }
        }
    }
    load_head(RC, qkv + head * g.hd, ld3, g, b, scale, tid);  // scaled Q over V
    __syncthreads();

    if (active) {
        floatx16 acc[2];
        // dQ = scale * dS K        (rows = queries, K = keys)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        mma_lds<2>(acc, ntd, SS + 32 * w * g.LDP, g.LDP, 1, RB, 1, g.LDH, g.SK >> 1, r, h);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int d = 32 * t + r;
            if (t < ntd && d < g.hd) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int s = 32 * w + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (s < g.S) d_qkv[(size_t)(s * g.B + b) * ld3 + head * g.hd + d] = acc[t][i] * scale;
                }
            }
        }
        // dK = dS^T (scale*Q)      (rows = keys, K = queries)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        mma_lds<2>(acc, ntd, SS + 32 * w, 1, g.LDP, RC, 1, g.LDH, g.NTr * 16, r, h);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int d = 32 * t + r;
            if (t < ntd && d < g.hd) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int s = 32 * w + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (s < g.S) d_qkv[(size_t)(s * g.B + b) * ld3 + g.E + head * g.hd + d] = acc[t][i];
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

int launch_attention_fwd(const float* qkv, float* o, int S, int B, int E, int H, float p, uint32_t site,
                         const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_TRY(check_attn(S, B, E, H));
    GF_CHECK_ARG(qkv && o, "attention_fwd: null pointer");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "attention_fwd: rng required when dropout is active");
    const AttnGeom g = make_geom(S, B, E, H);
    const size_t lds = (3 * hd_mat_floats(g) + ss_mat_floats(g)) * sizeof(float);
    GF_CHECK_ARG(lds <= 160 * 1024, "attention_fwd: LDS need %zu > 160 KiB", lds);
    if (lds > 48 * 1024) {  // opt in to large dynamic LDS (per device, so done per launch; host-only call)
        hipError_t e = hipFuncSetAttribute((const void*)attention_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail((int)e, "attention_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(B * H), dim3(256), lds, st, qkv, o, g, p, site, rng, add, train);
    GF_LAUNCH_CHECK();
    return 0;
}

int launch_attention_bwd(const float* qkv, const float* d_o, float* d_qkv, int S, int B, int E, int H, float p,
                         uint32_t site, const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    GF_TRY(check_attn(S, B, E, H));
    GF_CHECK_ARG(qkv && d_o && d_qkv, "attention_bwd: null pointer");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "attention_bwd: rng required when dropout is active");
    const AttnGeom g = make_geom(S, B, E, H);
    const size_t lds = (3 * hd_mat_floats(g) + ss_mat_floats(g)) * sizeof(float);
    GF_CHECK_ARG(lds <= 160 * 1024, "attention_bwd: LDS need %zu > 160 KiB", lds);
    if (lds > 48 * 1024) {  // opt in to large dynamic LDS (per device, so done per launch; host-only call)
        hipError_t e = hipFuncSetAttribute((const void*)attention_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail((int)e, "attention_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * H), dim3(256), lds, st, qkv, d_o, d_qkv, g, p, site, rng, add, train);
    GF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganffn
