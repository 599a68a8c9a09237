// disc_head.hip — the discriminator head (d_model 100 -> 64 -> 16 -> 1; /root/reference/model.py:1322-1326, 1360-1363,
// 1393-1396) as ONE kernel per direction:
//     forward   g0 = gelu(x); a1 = gelu(drop(g0 W1^T + b1)); a2 = gelu(drop(a1 W2^T + b2)); prob = sigmoid(drop(a2 w3 + b3))
//     backward  dprob -> d_pre3 -> d_pre2 -> d_pre1 -> dx   (+ the per-workgroup partial sums of the fc3 gradients)
// The head is 15 kFLOP per token — nothing next to the encoder stack — but it ran as 6 launches forward (GELU pass, two
// one-tile-wide GEMMs, the tail, a copy) and 8-10 backward, 18 + 12 times per iteration: ~1.5 ms of 5-14 us launch floors
// per iteration (profiles/r02_*).  Everything here is token-local, so a wave takes a group of 4 consecutive tokens (= one
// Philox call per output column, common.h drop_mult4) through the whole chain with plain fp32 FMAs; the weights (7.4 k
// floats) sit in LDS, transposed so that lane = output column reads consecutive addresses.  The weight gradients of fc1 /
// fc2 stay on the TN GEMMs (they reduce over the tokens); the backward kernel writes the d_pre1 / d_pre2 they read.
#include "common.h"

#pragma clang fp contract(off)

namespace ganffn {

namespace {

constexpr int HE = 100, H1 = 64, H2 = 16;

struct DiscHeadArgs {
    const float* x;                                   // [T x 100] encoder output
    const float* w1; const float* b1;                 // [64 x 100], [64]
    const float* w2; const float* b2;                 // [16 x 64], [16]
    const float* w3; const float* b3;                 // [16], [1]
    float* g0; float* u1; float* a1; float* u2; float* a2; float* prob;     // saved for backward (head_saved layout)
    float* out;                                       // [T] probabilities (may alias nothing; written besides prob)
    int T; float p; const uint64_t* rng; uint64_t add; int train;
};

__global__ __launch_bounds__(256) void disc_head_fwd_kernel(DiscHeadArgs a) {
    __shared__ float w1t[HE * H1];        // [k][c] = W1[c][k]
    __shared__ float w2t[H1 * H2];        // [k][c2] = W2[c2][k]
    __shared__ __attribute__((aligned(16))) float gs[4][4][HE + 4];      // per wave: gelu(x) of its 4 tokens
    __shared__ __attribute__((aligned(16))) float a1s[4][4][H1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (consecutive threads -> consecutive LDS words: conflict-free stores; the strided global reads hit L2, 30 KB in all)
    for (int i = tid; i < H1 * HE; i += 256) { const int k = i / H1, c = i - k * H1; w1t[i] = a.w1[c * HE + k]; }
    for (int i = tid; i < H2 * H1; i += 256) { const int k = i / H2, c = i - k * H2; w2t[i] = a.w2[c * H1 + k]; }
    const int rg = blockIdx.x * 4 + wv, t0 = rg * 4, T = a.T;
    const bool act = t0 < T;                               // (waves beyond the last row group still take part in the barriers)
    const DropCtx d1 = make_drop(a.rng, a.add, SITE_HEAD1, a.p, a.train);
    const DropCtx d2 = make_drop(a.rng, a.add, SITE_HEAD2, a.p, a.train);
    const DropCtx d3 = make_drop(a.rng, a.add, SITE_HEAD3, a.p, a.train);
    // ---- g0 = gelu(x): lanes cover the 100 columns in two chunks
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = min(t0 + q, T - 1);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int e = lane + 64 * ch;
            if (e < HE) {
                const float g = gelu_f(a.x[(size_t)t * HE + e]);
                gs[wv][q][e] = g;
                if (act && t0 + q < T) a.g0[(size_t)t * HE + e] = g;
            }
        }
    }
    __syncthreads();
    // ---- fc1: lane = output column c (64)
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int c = lane;
#pragma unroll 5
        for (int k = 0; k < HE; k += 4) {
            float4 g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = *reinterpret_cast<const float4*>(&gs[wv][q][k]);
            const float w0 = w1t[k * H1 + c], w1_ = w1t[(k + 1) * H1 + c], w2_ = w1t[(k + 2) * H1 + c], w3_ = w1t[(k + 3) * H1 + c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __fmaf_rn(g[q].x, w0, acc[q]); acc[q] = __fmaf_rn(g[q].y, w1_, acc[q]);
                acc[q] = __fmaf_rn(g[q].z, w2_, acc[q]); acc[q] = __fmaf_rn(g[q].w, w3_, acc[q]);
            }
        }
        float mult[4];
        drop_mult4(d1, (uint32_t)rg, (uint32_t)H1, (uint32_t)c, mult);
        const float bb = a.b1[c];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float u = (acc[q] + bb) * mult[q], v = gelu_f(u);
            a1s[wv][q][c] = v;
            if (act && t0 + q < T) { a.u1[(size_t)(t0 + q) * H1 + c] = u; a.a1[(size_t)(t0 + q) * H1 + c] = v; }
        }
    }
    __syncthreads();
    // ---- fc2 + fc3: lane = (c2 = lane & 15, token tq = lane >> 4)
    {
        const int c2 = lane & 15, tq = lane >> 4;
        float acc = 0.f;
#pragma unroll 4
        for (int k = 0; k < H1; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(&a1s[wv][tq][k]);
            acc = __fmaf_rn(v.x, w2t[k * H2 + c2], acc); acc = __fmaf_rn(v.y, w2t[(k + 1) * H2 + c2], acc);
            acc = __fmaf_rn(v.z, w2t[(k + 2) * H2 + c2], acc); acc = __fmaf_rn(v.w, w2t[(k + 3) * H2 + c2], acc);
        }
        float mult[4];
        drop_mult4(d2, (uint32_t)rg, (uint32_t)H2, (uint32_t)c2, mult);
        const float m2 = tq == 0 ? mult[0] : tq == 1 ? mult[1] : tq == 2 ? mult[2] : mult[3];
        const float u2v = (acc + a.b2[c2]) * m2, a2v = gelu_f(u2v);
        const int t = t0 + tq;
        if (act && t < T) { a.u2[(size_t)t * H2 + c2] = u2v; a.a2[(size_t)t * H2 + c2] = a2v; }
        float s = a2v * a.w3[c2];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
        float m3v[4];
        drop_mult4(d3, (uint32_t)rg, 1u, 0u, m3v);
        const float m3 = tq == 0 ? m3v[0] : tq == 1 ? m3v[1] : tq == 2 ? m3v[2] : m3v[3];
        const float u3 = (s + a.b3[0]) * m3;
        const float pr = 1.0f / (1.0f + expf(-u3));
        if (act && t < T && c2 == 0) { a.prob[t] = pr; a.out[t] = pr; }
    }
}

struct DiscHeadBwdArgs {
    const float* dprob;                               // [T]
    const float* x;                                   // [T x 100] encoder output (pre-GELU)
    const float* w1; const float* w2; const float* w3;
    const float* u1; const float* u2; const float* a2; const float* prob;   // saved by the forward
    float* dx;                                        // [T x 100]
    float* d_pre1; float* d_pre2;                     // [T x 64], [T x 16]: operands of the fc1 / fc2 weight-gradient GEMMs (or null)
    float* gpart;                                     // [gridDim.x][36] partial sums of the fc3 gradients (or null)
    int T; float p; const uint64_t* rng; uint64_t add; int train;
};

__global__ __launch_bounds__(256) void disc_head_bwd_kernel(DiscHeadBwdArgs a) {
    __shared__ float w1s[H1 * HE];        // [c][e] as stored
    __shared__ float w2s[H2 * H1];        // [j][c] as stored
    __shared__ __attribute__((aligned(16))) float dp2s[4][4][H2];
    __shared__ __attribute__((aligned(16))) float dp1s[4][4][H1];
    __shared__ float red[4][33];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < H1 * HE; i += 256) w1s[i] = a.w1[i];
    for (int i = tid; i < H2 * H1; i += 256) w2s[i] = a.w2[i];
    const int rg = blockIdx.x * 4 + wv, t0 = rg * 4, T = a.T;
    const DropCtx d1 = make_drop(a.rng, a.add, SITE_HEAD1, a.p, a.train);
    const DropCtx d2 = make_drop(a.rng, a.add, SITE_HEAD2, a.p, a.train);
    const DropCtx d3 = make_drop(a.rng, a.add, SITE_HEAD3, a.p, a.train);
    // ---- d_pre3 (per token) and d_pre2: lane = (c2, tq)
    {
        const int c2 = lane & 15, tq = lane >> 4, t = t0 + tq;
        const bool ok = t < T;
        const int tc = min(t, T - 1);
        float m3v[4], m2v[4];
        drop_mult4(d3, (uint32_t)rg, 1u, 0u, m3v);
        drop_mult4(d2, (uint32_t)rg, (uint32_t)H2, (uint32_t)c2, m2v);
        const float m3 = tq == 0 ? m3v[0] : tq == 1 ? m3v[1] : tq == 2 ? m3v[2] : m3v[3];
        const float m2 = tq == 0 ? m2v[0] : tq == 1 ? m2v[1] : tq == 2 ? m2v[2] : m2v[3];
        const float pr = a.prob[tc];
        const float dpre3 = ok ? a.dprob[tc] * pr * (1.0f - pr) * m3 : 0.f;
        const float dp2 = dpre3 * a.w3[c2] * gelu_grad_f(a.u2[(size_t)tc * H2 + c2]) * m2;
        dp2s[wv][tq][c2] = dp2;
        if (ok && a.d_pre2) a.d_pre2[(size_t)t * H2 + c2] = dp2;
        if (a.gpart) {
            // fc3 gradients: gw3[c2] += sum_t d_pre3 a2, gb3 += sum_t d_pre3 — this wave's 4 tokens, then the 4 waves
            float sw = dpre3 * (ok ? a.a2[(size_t)tc * H2 + c2] : 0.f);
            sw += __shfl_xor(sw, 16, 64); sw += __shfl_xor(sw, 32, 64);
            float sb = c2 == 0 ? dpre3 : 0.f;
            sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
            if (tq == 0) red[wv][c2] = sw;
            if (lane == 0) red[wv][32] = sb;
        }
    }
    __syncthreads();
    if (a.gpart) {
        if (tid < H2) a.gpart[blockIdx.x * 36 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        if (tid == 32) a.gpart[blockIdx.x * 36 + 32] = ((red[0][32] + red[1][32]) + red[2][32]) + red[3][32];
    }
    // ---- d_pre1: lane = column c of fc1's output
    {
        const int c = lane;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < H2; j += 4) {
            const float w0 = w2s[j * H1 + c], w1_ = w2s[(j + 1) * H1 + c], w2_ = w2s[(j + 2) * H1 + c], w3_ = w2s[(j + 3) * H1 + c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&dp2s[wv][q][j]);
                acc[q] = __fmaf_rn(v.x, w0, acc[q]); acc[q] = __fmaf_rn(v.y, w1_, acc[q]);
                acc[q] = __fmaf_rn(v.z, w2_, acc[q]); acc[q] = __fmaf_rn(v.w, w3_, acc[q]);
            }
        }
        float mult[4];
        drop_mult4(d1, (uint32_t)rg, (uint32_t)H1, (uint32_t)c, mult);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = t0 + q, tc = min(t, T - 1);
            const float v = t < T ? acc[q] * mult[q] * gelu_grad_f(a.u1[(size_t)tc * H1 + c]) : 0.f;
            dp1s[wv][q][c] = v;
            if (t < T && a.d_pre1) a.d_pre1[(size_t)t * H1 + c] = v;
        }
    }
    __syncthreads();
    // ---- dx = (d_pre1 W1) * gelu'(x): lanes cover the 100 columns in two chunks
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int e = lane + 64 * ch;
        if (e < HE) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int c = 0; c < H1; c += 4) {
                const float w0 = w1s[c * HE + e], w1_ = w1s[(c + 1) * HE + e], w2_ = w1s[(c + 2) * HE + e], w3_ = w1s[(c + 3) * HE + e];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(&dp1s[wv][q][c]);
                    acc[q] = __fmaf_rn(v.x, w0, acc[q]); acc[q] = __fmaf_rn(v.y, w1_, acc[q]);
                    acc[q] = __fmaf_rn(v.z, w2_, acc[q]); acc[q] = __fmaf_rn(v.w, w3_, acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = t0 + q;
                if (t < T) a.dx[(size_t)t * HE + e] = acc[q] * gelu_grad_f(a.x[(size_t)t * HE + e]);
            }
        }
    }
}

}  // namespace

bool disc_head_fused_supported(int E, int D1, int D2) { return E == HE && D1 == H1 && D2 == H2; }
int disc_head_blocks(int T) { return ((T + 3) / 4 + 3) / 4; }

int launch_disc_head_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                         const float* b3, float* g0, float* u1, float* a1, float* u2, float* a2, float* prob, float* out, int T,
                         float p, const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    DiscHeadArgs a{x, w1, b1, w2, b2, w3, b3, g0, u1, a1, u2, a2, prob, out, T, p, rng, add, train};
    hipLaunchKernelGGL(disc_head_fwd_kernel, dim3(disc_head_blocks(T)), dim3(256), 0, st, a);
    GF_LAUNCH_CHECK();
    return 0;
}

// gpart: disc_head_blocks(T) * 36 floats (fc3 gradient partial sums; the caller reduces them in block order) or null
int launch_disc_head_bwd(const float* dprob, const float* x, const float* w1, const float* w2, const float* w3, const float* u1,
                         const float* u2, const float* a2, const float* prob, float* dx, float* d_pre1, float* d_pre2, float* gpart,
                         int T, float p, const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    DiscHeadBwdArgs a{dprob, x, w1, w2, w3, u1, u2, a2, prob, dx, d_pre1, d_pre2, gpart, T, p, rng, add, train};
    hipLaunchKernelGGL(disc_head_bwd_kernel, dim3(disc_head_blocks(T)), dim3(256), 0, st, a);
    GF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganffn
