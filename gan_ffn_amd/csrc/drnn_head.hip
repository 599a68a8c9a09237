// drnn_head.hip — the data movement around the DialogueRNN recurrence of configuration 5 (GAN_FFN_DialogueRNN's BiModel,
// /root/reference/model.py:1008-1062) that the step runner (engine.DrnnEngine) needs as single launches:
//  * BiModel._reverse_seq (model.py:1008-1021): reverse each dialogue's valid prefix, zero the padding — forward and, being
//    its own transpose, the gradient (optionally accumulated into the destination);
//  * emotions = cat(dropout_rec(e_f), dropout_rec(reverse(e_b))) (model.py:1035-1041) and its backward;
//  * d masked by a saved activation (the backward through ReLU + dropout of BiModel.linear, model.py:1051-1053).
// Dropout follows the Philox contract of common.h (one call = 4 consecutive token rows of a column).
#include "common.h"

namespace ganffn {

namespace {

// out[s, b, :] (+)= s < len_b ? x[len_b - 1 - s, b, :] : 0
__global__ void seq_reverse_kernel(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ out, int S, int B,
                                   int D4, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)S * B * D4) return;
    const int d4 = (int)(i % D4), tb = (int)(i / D4), b = tb % B, s = tb / B;
    const int len = lens[b];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < len) v = reinterpret_cast<const float4*>(x)[((size_t)(len - 1 - s) * B + b) * D4 + d4];
    float4* o = reinterpret_cast<float4*>(out) + i;
    if (accumulate) { const float4 a = *o; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    *o = v;
}

// forward: em[s, b, 0:De] = drop_A(e_f[s, b]);  em[s, b, De:2De] = drop_B(rev(e_b)[s, b])
// backward (BWD): d_e_f[s, b] = drop_A(d_em[s, b, 0:De]);  d_e_b[len-1-s, b] = drop_B(d_em[s, b, De:2De]) (rows >= len: 0)
template <bool BWD>
__global__ void drnn_join_kernel(const float* __restrict__ a, const float* __restrict__ bsrc, const int* __restrict__ lens,
                                 float* __restrict__ o1, float* __restrict__ o2, int S, int B, int De, float p, uint32_t site_f,
                                 uint32_t site_b, const uint64_t* __restrict__ rng, uint64_t add, int train) {
    const int T = S * B, G = (T + 3) >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)G * 2 * De) return;
    const int rg = (int)(idx / (2 * De)), cc = (int)(idx - (long)rg * 2 * De);
    const bool second = cc >= De;
    const int c = second ? cc - De : cc;
    const DropCtx dc = make_drop(rng, add, second ? site_b : site_f, p, train);
    float mult[4];
    drop_mult4(dc, (uint32_t)rg, (uint32_t)De, (uint32_t)c, mult);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = rg * 4 + q;
        if (t >= T) continue;
        const int s = t / B, b = t - s * B, len = lens[b];
        if (!BWD) {
            // a = e_f, bsrc = e_b (in the reverse direction's own order), o1 = emotions [T x 2De]
            float v;
            if (!second) v = a[(size_t)t * De + c];
            else v = s < len ? bsrc[((size_t)(len - 1 - s) * B + b) * De + c] : 0.f;
            o1[(size_t)t * 2 * De + cc] = v * mult[q];
        } else {
            // a = d_emotions [T x 2De], o1 = d_e_f, o2 = d_e_b
            const float v = a[(size_t)t * 2 * De + cc] * mult[q];
            if (!second) o1[(size_t)t * De + c] = v;
            else if (s < len) o2[((size_t)(len - 1 - s) * B + b) * De + c] = v;
            else o2[(size_t)t * De + c] = 0.f;           // rows beyond the dialogue: written exactly once, by their own row
        }
    }
}

__global__ void mask_pos_kernel(float* __restrict__ d, const float* __restrict__ aux, float mscale, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = aux[i] > 0.f ? d[i] * mscale : 0.f;
}

}  // namespace
}  // namespace ganffn

using namespace ganffn;

extern "C" int ganffn_seq_reverse(const float* x, const int32_t* lens, float* out, int S, int B, int D, int accumulate, void* stream) {
    GF_CHECK_ARG(x && lens && out && S > 0 && B > 0 && D > 0 && (D & 3) == 0, "seq_reverse: bad arguments");
    GF_CHECK_ARG(aligned16(x) && aligned16(out), "seq_reverse: buffers must be 16-byte aligned");
    const long n = (long)S * B * (D / 4);
    hipLaunchKernelGGL(seq_reverse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, lens, out, S, B,
                       D / 4, accumulate);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_drnn_join_fwd(const float* e_f, const float* e_b, const int32_t* lens, float* emotions, int S, int B, int De,
                                    float p, uint32_t site_f, uint32_t site_b, const uint64_t* rng, uint64_t rng_offset_add,
                                    int train, void* stream) {
    GF_CHECK_ARG(e_f && e_b && lens && emotions && S > 0 && B > 0 && De > 0, "drnn_join_fwd: bad arguments");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "drnn_join_fwd: rng required in train mode");
    const long n = (long)((S * B + 3) / 4) * 2 * De;
    hipLaunchKernelGGL(drnn_join_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e_f, e_b, lens,
                       emotions, (float*)nullptr, S, B, De, p, site_f, site_b, rng, rng_offset_add, train);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_drnn_join_bwd(const float* d_emotions, const int32_t* lens, float* d_e_f, float* d_e_b, int S, int B, int De,
                                    float p, uint32_t site_f, uint32_t site_b, const uint64_t* rng, uint64_t rng_offset_add,
                                    int train, void* stream) {
    GF_CHECK_ARG(d_emotions && lens && d_e_f && d_e_b && S > 0 && B > 0 && De > 0, "drnn_join_bwd: bad arguments");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "drnn_join_bwd: rng required in train mode");
    const long n = (long)((S * B + 3) / 4) * 2 * De;
    hipLaunchKernelGGL(drnn_join_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_emotions,
                       (const float*)nullptr, lens, d_e_f, d_e_b, S, B, De, p, site_f, site_b, rng, rng_offset_add, train);
    GF_LAUNCH_CHECK();
    return 0;
}

// d[i] = aux[i] > 0 ? d[i] * mscale : 0  (in place): the gradient through dropout(relu(.)) given the saved output
extern "C" int ganffn_mask_pos_inplace(float* d, const float* aux, float mscale, int64_t n, void* stream) {
    GF_CHECK_ARG(d && aux && n > 0, "mask_pos_inplace: bad arguments");
    hipLaunchKernelGGL(mask_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, aux, mscale, (long)n);
    GF_LAUNCH_CHECK();
    return 0;
}
