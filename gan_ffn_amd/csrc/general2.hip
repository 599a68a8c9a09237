// general2.hip — masked "general2" matching attention, every time step of a dialogue as the query (gfx950).
//
// Replaces the loop of BiModel.forward over MatchingAttention(att_type="general2") calls
// (/root/reference/model.py:1043-1049 calling :169-182, :193): for dialogue b and query step t
//     u_j = m_j * < x_t , m_j * M_j >            x = transform(M) (done by the caller's GEMM), m = umask[b]
//     s_j = tanh(u_j)
//     a_j = softmax_j(s)_j * m_j / sum_k softmax_k(s)_k * m_k  =  e^{s_j} m_j / sum_k e^{s_k} m_k
//     att_t = sum_j a_j M_j
// s is bounded by tanh, so no max-subtraction is needed; a dialogue with no valid step yields 0/0 = NaN as the
// reference does.  Fallback kernels (dialogue too large for the LDS): one workgroup per (query step, dialogue); S <= 128, D <= 1024 (D = 2 D_e = 200 in config 5, 600 in the MELD classifier).
// HBM-bound: M[b] (S x D floats) is re-read by the S workgroups of a dialogue out of L2.
#include "common.h"

namespace ganffn {

constexpr int G2_MAXS = 128;
constexpr int G2_MAXD = 1024;

__global__ __launch_bounds__(256) void general2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mem,
                                                           const float* __restrict__ mask, float* __restrict__ att,
                                                           float* __restrict__ alpha, float* __restrict__ tanh_s, int S,
                                                           int B, int D) {
    __shared__ float xs[G2_MAXD];
    __shared__ float a_s[G2_MAXS];
    __shared__ float red[4];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t row = (size_t)B * D;                     // stride between time steps
    for (int d = tid; d < D; d += 256) xs[d] = x[(size_t)t * row + (size_t)b * D + d];
    __syncthreads();
    float part = 0.f;                                     // this wave's share of sum_k e^{s_k} m_k
    for (int j = w; j < S; j += 4) {
        const float* mj = mem + (size_t)j * row + (size_t)b * D;
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += xs[d] * mj[d];
        dot = wave_sum(dot);
        const float m = mask[(size_t)b * S + j];
        const float s = tanhf(m * (m * dot));
        const float e = __expf(s) * m;
        if (lane == 0) {
            a_s[j] = e;
            tanh_s[((size_t)b * S + t) * S + j] = s;
        }
        part += e;                                        // identical in every lane
    }
    if (lane == 0) red[w] = part;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int j = tid; j < S; j += 256) {
        const float a = a_s[j] * inv;
        a_s[j] = a;
        alpha[((size_t)b * S + t) * S + j] = a;
    }
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc += a_s[j] * mem[(size_t)j * row + (size_t)b * D + d];
        att[(size_t)t * row + (size_t)b * D + d] = acc;
    }
}

// per (query step, dialogue): du_j = a_j (da_j - sum_k a_k da_k) (1 - s_j^2) m_j^2 with da_j = <d_att_t, M_j>;
// dx_t = sum_j du_j M_j.  du is kept for the memory-gradient kernel.
__global__ __launch_bounds__(256) void general2_bwd_q_kernel(const float* __restrict__ d_att, const float* __restrict__ mem,
                                                             const float* __restrict__ mask, const float* __restrict__ alpha,
                                                             const float* __restrict__ tanh_s, float* __restrict__ du,
                                                             float* __restrict__ dx, int S, int B, int D) {
    __shared__ float gs[G2_MAXD];
    __shared__ float da_s[G2_MAXS];
    __shared__ float red[4];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t row = (size_t)B * D;
    const size_t arow = ((size_t)b * S + t) * S;
    for (int d = tid; d < D; d += 256) gs[d] = d_att[(size_t)t * row + (size_t)b * D + d];
    __syncthreads();
    float part = 0.f;
    for (int j = w; j < S; j += 4) {
        const float* mj = mem + (size_t)j * row + (size_t)b * D;
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += gs[d] * mj[d];
        dot = wave_sum(dot);
        if (lane == 0) da_s[j] = dot;
        part += alpha[arow + j] * dot;
    }
    if (lane == 0) red[w] = part;
    __syncthreads();
    const float dsum = red[0] + red[1] + red[2] + red[3];
    for (int j = tid; j < S; j += 256) {
        const float a = alpha[arow + j], s = tanh_s[arow + j], m = mask[(size_t)b * S + j];
        const float v = a * (da_s[j] - dsum) * (1.0f - s * s) * (m * m);
        da_s[j] = v;
        du[arow + j] = v;
    }
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc += da_s[j] * mem[(size_t)j * row + (size_t)b * D + d];
        dx[(size_t)t * row + (size_t)b * D + d] = acc;
    }
}

// per (memory step j, dialogue): dM_j = sum_t ( a_tj d_att_t + du_tj x_t )   — no atomics, fixed order
__global__ __launch_bounds__(256) void general2_bwd_m_kernel(const float* __restrict__ d_att, const float* __restrict__ x,
                                                             const float* __restrict__ alpha, const float* __restrict__ du,
                                                             float* __restrict__ dmem, int S, int B, int D) {
    __shared__ float a_s[G2_MAXS];
    __shared__ float u_s[G2_MAXS];
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t row = (size_t)B * D;
    for (int t = tid; t < S; t += 256) {
        a_s[t] = alpha[((size_t)b * S + t) * S + j];
        u_s[t] = du[((size_t)b * S + t) * S + j];
    }
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        float acc = 0.f;
        for (int t = 0; t < S; ++t) {
            const size_t o = (size_t)t * row + (size_t)b * D + d;
            acc += a_s[t] * d_att[o] + u_s[t] * x[o];
        }
        dmem[(size_t)j * row + (size_t)b * D + d] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Dialogue-resident versions (used whenever the dialogue's S x D memory fits the LDS): one 512-thread workgroup per (block of
// QB = 16 query steps, dialogue) — the S x D image takes half a CU's LDS, so a CU holds one workgroup and the grid is sized to
// fit the chip in one round at the configuration-5 shape (6 x 30 = 180 workgroups).  The kernels above re-read a dialogue's memory from L2 once per query step (424 MB per launch
// at (94, 30, 200): 78 / 66 / 28 us, profiles/r02_general2_line.txt); here M[b] is staged ONCE per 8 queries, so a
// launch pulls ~27 MB through L2.  Same arithmetic, same (fixed) summation orders per output element as above except that
// dot products run serially over d per (key, query) instead of as a 64-lane tree.
constexpr int G2_QB = 16;    // query (or key) steps per workgroup
constexpr int G2_NT = 512;   // threads per workgroup

__host__ __device__ __forceinline__ int g2_ldm(int D) { return D | 1; }          // odd row stride: key-on-lane reads conflict-free

// stage M[b] ([S x D], row stride B*D in global) into Ms[S][ldm] and V[t0 .. t0+7] (same layout) transposed into vT[d][QB]
__device__ __forceinline__ void g2_stage(float* __restrict__ Ms, float* __restrict__ vT, const float* __restrict__ mem,
                                         const float* __restrict__ v, int S, int B, int D, int b, int t0, int tid) {
    const int ldm = g2_ldm(D);
    const size_t row = (size_t)B * D;
    const float* mb = mem + (size_t)b * D;
    if ((D & 3) == 0) {
        // coalesced 16-byte loads over the flat (step, column/4) index, 8 independent loads in flight per thread
        const int kv = D >> 2, total = S * kv;
        for (int i0 = tid; i0 < total; i0 += G2_NT * 8) {
            float4 r[8];
            int jj[8], dd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * G2_NT, total - 1);
                jj[u] = i / kv;
                dd[u] = (i - jj[u] * kv) << 2;
                r[u] = *reinterpret_cast<const float4*>(mb + (size_t)jj[u] * row + dd[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u * G2_NT < total) {
                    float* dst = Ms + jj[u] * ldm + dd[u];
                    dst[0] = r[u].x; dst[1] = r[u].y; dst[2] = r[u].z; dst[3] = r[u].w;
                }
            }
        }
    } else {
        for (int d = tid; d < D; d += G2_NT)
            for (int j0 = 0; j0 < S; j0 += 16) {
                float r[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) r[u] = mb[(size_t)min(j0 + u, S - 1) * row + d];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (j0 + u < S) Ms[(j0 + u) * ldm + d] = r[u];
            }
    }
    for (int d = tid; d < D; d += G2_NT) {
        float r[G2_QB];
#pragma unroll
        for (int q = 0; q < G2_QB; ++q) r[q] = v[(size_t)min(t0 + q, S - 1) * row + (size_t)b * D + d];   // clamped: rows >= S are never stored
#pragma unroll
        for (int q = 0; q < G2_QB; ++q) vT[d * G2_QB + q] = r[q];
    }
}

// dots[q] = < V[t0 + 4 qh + q], M_j > for this thread's key j and its 4 queries
__device__ __forceinline__ void g2_dots(float (&acc)[4], const float* __restrict__ Ms, const float* __restrict__ vT, int D, int j,
                                        int qh) {
    const int ldm = g2_ldm(D);
    acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
    const float* mrow = Ms + j * ldm;
#pragma unroll 8
    for (int d = 0; d < D; ++d) {
        const float m = mrow[d];
        const float4 xv = *reinterpret_cast<const float4*>(vT + d * G2_QB + 4 * qh);   // broadcast read
        acc[0] += m * xv.x; acc[1] += m * xv.y; acc[2] += m * xv.z; acc[3] += m * xv.w;
    }
}

// out[t0 + q][d] = sum_j cT[j][q] * M[j][d] for q < QB (rows < S only); thread = (column d, half of the 16 outputs)
__device__ __forceinline__ void g2_combine(float* __restrict__ out, const float* __restrict__ Ms, const float* __restrict__ cT,
                                           int S, int B, int D, int b, int t0, int tid) {
    const int ldm = g2_ldm(D);
    const size_t row = (size_t)B * D;
    const int qh = tid >> 8;
    for (int d = tid & 255; d < D; d += 256) {
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll 4
        for (int j = 0; j < S; ++j) {
            const float m = Ms[j * ldm + d];
            const float4 c0 = *reinterpret_cast<const float4*>(cT + j * G2_QB + 8 * qh), c1 = *reinterpret_cast<const float4*>(cT + j * G2_QB + 8 * qh + 4);
            acc[0] += c0.x * m; acc[1] += c0.y * m; acc[2] += c0.z * m; acc[3] += c0.w * m;
            acc[4] += c1.x * m; acc[5] += c1.y * m; acc[6] += c1.z * m; acc[7] += c1.w * m;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (t0 + 8 * qh + q < S) out[(size_t)(t0 + 8 * qh + q) * row + (size_t)b * D + d] = acc[q];
    }
}

__global__ __launch_bounds__(G2_NT) void general2_fwd_lds_kernel(const float* __restrict__ x, const float* __restrict__ mem,
                                                               const float* __restrict__ mask, float* __restrict__ att,
                                                               float* __restrict__ alpha, float* __restrict__ tanh_s, int S,
                                                               int B, int D) {
    extern __shared__ __attribute__((aligned(16))) float g2s[];
    float* Ms = g2s;                                   // [S][ldm]
    float* xT = Ms + ((S * g2_ldm(D) + 3) & ~3);       // [D][QB]
    float* aT = xT + D * G2_QB;                        // [S][QB]
    float* inv_s = aT + S * G2_QB;                     // [QB]  (no static LDS: the dynamic opt-in covers the whole 160 KB)
    const int t0 = blockIdx.x * G2_QB, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    g2_stage(Ms, xT, mem, x, S, B, D, b, t0, tid);
    __syncthreads();
    {
        const int j = tid & 127, qh = tid >> 7;
        if (j < S) {
            float acc[4];
            g2_dots(acc, Ms, xT, D, j, qh);
            const float m = mask[(size_t)b * S + j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sv = tanhf(m * (m * acc[q]));
                aT[j * G2_QB + 4 * qh + q] = __expf(sv) * m;
                const int t = t0 + 4 * qh + q;
                if (t < S) tanh_s[((size_t)b * S + t) * S + j] = sv;
            }
        }
    }
    __syncthreads();
    for (int q = 2 * w; q < 2 * w + 2; ++q) {          // wave w normalises queries 2w, 2w + 1
        float part = 0.f;
        for (int j = lane; j < S; j += 64) part += aT[j * G2_QB + q];
        part = wave_sum(part);
        if (lane == 0) inv_s[q] = 1.0f / part;
    }
    __syncthreads();
    for (int i = tid; i < S * G2_QB; i += G2_NT) {
        const int j = i / G2_QB, q = i - j * G2_QB;
        const float a = aT[i] * inv_s[q];
        aT[i] = a;
        if (t0 + q < S) alpha[((size_t)b * S + t0 + q) * S + j] = a;
    }
    __syncthreads();
    g2_combine(att, Ms, aT, S, B, D, b, t0, tid);
}

__global__ __launch_bounds__(G2_NT) void general2_bwd_q_lds_kernel(const float* __restrict__ d_att, const float* __restrict__ mem,
                                                                 const float* __restrict__ mask, const float* __restrict__ alpha,
                                                                 const float* __restrict__ tanh_s, float* __restrict__ du,
                                                                 float* __restrict__ dx, int S, int B, int D) {
    extern __shared__ __attribute__((aligned(16))) float g2s[];
    float* Ms = g2s;
    float* gT = Ms + ((S * g2_ldm(D) + 3) & ~3);       // d_att rows of the block, transposed [D][QB]
    float* uT = gT + D * G2_QB;                        // da, then du: [S][QB]
    float* dsum_s = uT + S * G2_QB;                    // [QB]
    const int t0 = blockIdx.x * G2_QB, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    g2_stage(Ms, gT, mem, d_att, S, B, D, b, t0, tid);
    __syncthreads();
    {
        const int j = tid & 127, qh = tid >> 7;
        if (j < S) {
            float acc[4];
            g2_dots(acc, Ms, gT, D, j, qh);
#pragma unroll
            for (int q = 0; q < 4; ++q) uT[j * G2_QB + 4 * qh + q] = acc[q];
        }
    }
    __syncthreads();
    for (int q = 2 * w; q < 2 * w + 2; ++q) {          // dsum_q = sum_j alpha[q][j] da[q][j]
        const int t = min(t0 + q, S - 1);
        float part = 0.f;
        for (int j = lane; j < S; j += 64) part += alpha[((size_t)b * S + t) * S + j] * uT[j * G2_QB + q];
        part = wave_sum(part);
        if (lane == 0) dsum_s[q] = part;
    }
    __syncthreads();
    for (int i = tid; i < S * G2_QB; i += G2_NT) {
        const int j = i / G2_QB, q = i - j * G2_QB;
        const int t = min(t0 + q, S - 1);
        const size_t ar = ((size_t)b * S + t) * S + j;
        const float a = alpha[ar], sv = tanh_s[ar], m = mask[(size_t)b * S + j];
        const float v = a * (uT[i] - dsum_s[q]) * (1.0f - sv * sv) * (m * m);
        uT[i] = v;
        if (t0 + q < S) du[ar] = v;
    }
    __syncthreads();
    g2_combine(dx, Ms, uT, S, B, D, b, t0, tid);
}

// per (block of 8 memory steps j0 .., dialogue): dM_j = sum_t ( a_tj d_att_t + du_tj x_t ), t ascending
__global__ __launch_bounds__(G2_NT) void general2_bwd_m_lds_kernel(const float* __restrict__ d_att, const float* __restrict__ x,
                                                                 const float* __restrict__ alpha, const float* __restrict__ du,
                                                                 float* __restrict__ dmem, int S, int B, int D) {
    extern __shared__ __attribute__((aligned(16))) float g2s[];
    float* ca = g2s;                                   // [S][QB] alpha[t][j0 + q]
    float* cu = ca + S * G2_QB;                        // [S][QB] du[t][j0 + q]
    const int j0 = blockIdx.x * G2_QB, b = blockIdx.y, tid = threadIdx.x;
    const size_t row = (size_t)B * D;
    for (int i = tid; i < S * G2_QB; i += G2_NT) {
        const int t = i / G2_QB, q = i - t * G2_QB;
        const size_t o = ((size_t)b * S + t) * S + min(j0 + q, S - 1);
        ca[i] = alpha[o];
        cu[i] = du[o];
    }
    __syncthreads();
    const int qh = tid >> 8;
    for (int d = tid & 255; d < D; d += 256) {
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        const float* gp = d_att + (size_t)b * D + d;
        const float* xp = x + (size_t)b * D + d;
        for (int t0 = 0; t0 < S; t0 += 8) {            // 16 loads in flight
            float gv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = min(t0 + u, S - 1);
                gv[u] = gp[(size_t)t * row];
                xv[u] = xp[(size_t)t * row];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < S) {
                    const float* cap = ca + (t0 + u) * G2_QB + 8 * qh;
                    const float* cup = cu + (t0 + u) * G2_QB + 8 * qh;
                    const float4 a0 = *reinterpret_cast<const float4*>(cap), a1 = *reinterpret_cast<const float4*>(cap + 4);
                    const float4 c0 = *reinterpret_cast<const float4*>(cup), c1 = *reinterpret_cast<const float4*>(cup + 4);
                    acc[0] += a0.x * gv[u] + c0.x * xv[u]; acc[1] += a0.y * gv[u] + c0.y * xv[u];
                    acc[2] += a0.z * gv[u] + c0.z * xv[u]; acc[3] += a0.w * gv[u] + c0.w * xv[u];
                    acc[4] += a1.x * gv[u] + c1.x * xv[u]; acc[5] += a1.y * gv[u] + c1.y * xv[u];
                    acc[6] += a1.z * gv[u] + c1.z * xv[u]; acc[7] += a1.w * gv[u] + c1.w * xv[u];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + 8 * qh + q < S) dmem[(size_t)(j0 + 8 * qh + q) * row + (size_t)b * D + d] = acc[q];
    }
}

static size_t g2_lds_bytes(int S, int D) {
    return (size_t)(((S * g2_ldm(D) + 3) & ~3) + D * G2_QB + S * G2_QB + G2_QB) * sizeof(float);
}
static bool g2_fits_lds(int S, int D) { return g2_lds_bytes(S, D) <= 150 * 1024; }

static int check_g2(int S, int B, int D) {
    GF_CHECK_ARG(S >= 1 && S <= G2_MAXS, "general2_attention: S=%d out of [1,%d]", S, G2_MAXS);
    GF_CHECK_ARG(D >= 1 && D <= G2_MAXD, "general2_attention: D=%d out of [1,%d]", D, G2_MAXD);
    GF_CHECK_ARG(B >= 1 && B <= 65535, "general2_attention: B=%d", B);
    return 0;
}

}  // namespace ganffn

using namespace ganffn;

extern "C" int ganffn_general2_attention_fwd(const float* x, const float* mem, const float* mask, float* att, float* alpha,
                                             float* tanh_s, int S, int B, int D, void* stream) {
    GF_TRY(check_g2(S, B, D));
    GF_CHECK_ARG(x && mem && mask && att && alpha && tanh_s, "general2_attention_fwd: null pointer");
    if (g2_fits_lds(S, D)) {
        const size_t lds = g2_lds_bytes(S, D);
        GF_TRY((lds_optin<general2_fwd_lds_kernel>(lds, "general2_fwd")));
        hipLaunchKernelGGL(general2_fwd_lds_kernel, dim3((S + G2_QB - 1) / G2_QB, B), dim3(G2_NT), lds, (hipStream_t)stream, x, mem,
                           mask, att, alpha, tanh_s, S, B, D);
        GF_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(general2_fwd_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, x, mem, mask, att, alpha, tanh_s,
                       S, B, D);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_general2_attention_bwd(const float* d_att, const float* x, const float* mem, const float* mask,
                                             const float* alpha, const float* tanh_s, float* du_ws, float* dx, float* dmem,
                                             int S, int B, int D, void* stream) {
    GF_TRY(check_g2(S, B, D));
    GF_CHECK_ARG(d_att && x && mem && mask && alpha && tanh_s && du_ws && dx && dmem, "general2_attention_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (g2_fits_lds(S, D)) {
        const size_t lds = g2_lds_bytes(S, D);
        const dim3 grid((S + G2_QB - 1) / G2_QB, B);
        GF_TRY((lds_optin<general2_bwd_q_lds_kernel>(lds, "general2_bwd_q")));
        hipLaunchKernelGGL(general2_bwd_q_lds_kernel, grid, dim3(G2_NT), lds, st, d_att, mem, mask, alpha, tanh_s, du_ws, dx, S, B, D);
        GF_LAUNCH_CHECK();
        hipLaunchKernelGGL(general2_bwd_m_lds_kernel, grid, dim3(G2_NT), (size_t)2 * S * G2_QB * sizeof(float), st, d_att, x, alpha,
                           du_ws, dmem, S, B, D);
        GF_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(general2_bwd_q_kernel, dim3(S, B), dim3(256), 0, st, d_att, mem, mask, alpha, tanh_s, du_ws, dx, S, B, D);
    GF_LAUNCH_CHECK();
    hipLaunchKernelGGL(general2_bwd_m_kernel, dim3(S, B), dim3(256), 0, st, d_att, x, alpha, du_ws, dmem, S, B, D);
    GF_LAUNCH_CHECK();
    return 0;
}
