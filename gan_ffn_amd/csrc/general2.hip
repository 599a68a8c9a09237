// general2.hip — masked "general2" matching attention, every time step of a dialogue as the query (gfx950).
//
// Replaces the loop of BiModel.forward over MatchingAttention(att_type="general2") calls
// (/root/reference/model.py:1043-1049 calling :169-182, :193): for dialogue b and query step t
//     u_j = m_j * < x_t , m_j * M_j >            x = transform(M) (done by the caller's GEMM), m = umask[b]
//     s_j = tanh(u_j)
//     a_j = softmax_j(s)_j * m_j / sum_k softmax_k(s)_k * m_k  =  e^{s_j} m_j / sum_k e^{s_k} m_k
//     att_t = sum_j a_j M_j
// s is bounded by tanh, so no max-subtraction is needed; a dialogue with no valid step yields 0/0 = NaN as the
// reference does.  One workgroup per (query step, dialogue); S <= 128, D <= 1024 (D = 2 D_e = 200 in config 5, 600 in the MELD classifier).
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
    hipLaunchKernelGGL(general2_bwd_q_kernel, dim3(S, B), dim3(256), 0, st, d_att, mem, mask, alpha, tanh_s, du_ws, dx, S, B, D);
    GF_LAUNCH_CHECK();
    hipLaunchKernelGGL(general2_bwd_m_kernel, dim3(S, B), dim3(256), 0, st, d_att, x, alpha, du_ws, dmem, S, B, D);
    GF_LAUNCH_CHECK();
    return 0;
}
