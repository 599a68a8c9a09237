// elementwise.hip — HBM-bound row/elementwise kernels of the GAN-FFN step: PositionalEncoding add,
// dropout, residual+dropout+LayerNorm (fwd/bwd), GELU chains, discriminator tail, BCE, Adam,
// log-softmax + masked NLL.  All fp32.  Dropout follows the Philox contract (4 consecutive rows of one
// column share one Philox call), so row-wise kernels process rows in groups of 4.
#include "common.h"

// No implicit floating-point contraction in this file: left on, hipcc fuses a*b+c into an fma in some of the four row
// slots a wave works on and not in others, which makes a token's bits depend on its position in the batch (measured:
// 1 ulp in LayerNorm).  Every fma below is written explicitly (__fmaf_rn); these kernels are HBM-bound, so the extra
// rounding steps cost nothing.
#pragma clang fp contract(off)

namespace ganffn {

// ------------------------------------------------------------------------------------------
// A1: out = dropout_p(x + pe[s])        /root/reference/model.py:1196-1197
// one thread per (row group of 4, column)
// ------------------------------------------------------------------------------------------
__global__ void pe_dropout_kernel(const float* __restrict__ x, const float* __restrict__ pe, float* __restrict__ out,
                                  int T, int B, int E, float p, const uint64_t* __restrict__ rng, uint64_t add,
                                  int train) {
    const int G = (T + 3) >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)G * E) return;
    const int rg = (int)(idx / E), c = (int)(idx - (long)rg * E);
    const DropCtx dc = make_drop(rng, add, SITE_PE, p, train);
    float mult[4];
    drop_mult4(dc, (uint32_t)rg, (uint32_t)E, (uint32_t)c, mult);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = rg * 4 + q;
        if (t < T) {
            const int s = t / B;
            out[(size_t)t * E + c] = (x[(size_t)t * E + c] + pe[(size_t)s * E + c]) * mult[q];
        }
    }
}

// generic: out = f(x) * mult, f = identity (MODE 0) or gelu (MODE 1);  MODE 2: out = d * gelu'(u) * mult
template <int MODE>
__global__ void drop_kernel(const float* __restrict__ x, const float* __restrict__ u, float* __restrict__ out, int R,
                            int C, float p, uint32_t site, const uint64_t* __restrict__ rng, uint64_t add, int train) {
    const int G = (R + 3) >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)G * C) return;
    const int rg = (int)(idx / C), c = (int)(idx - (long)rg * C);
    const DropCtx dc = make_drop(rng, add, site, p, train);
    float mult[4];
    drop_mult4(dc, (uint32_t)rg, (uint32_t)C, (uint32_t)c, mult);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = rg * 4 + q;
        if (t < R) {
            const size_t o = (size_t)t * C + c;
            float v = x[o];
            if (MODE == 1) v = gelu_f(v);
            if (MODE == 2) v = v * gelu_grad_f(u[o]);
            out[o] = v * mult[q];
        }
    }
}

__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 x = reinterpret_cast<float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    reinterpret_cast<float4*>(a)[i] = x;
}

__global__ void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                            float* __restrict__ o, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i] + c[i];
}

// ------------------------------------------------------------------------------------------
// z = x + dropout(y); xhat = (z - mean) * rstd; out = xhat * w + b        (norm1 / norm2, post-LN)
// one wave per group of 4 rows; lane covers columns lane, lane+64, ...  (E <= 64*MAXC)
// ------------------------------------------------------------------------------------------
constexpr int LN_MAXC = 10;  // E <= 640 (d_model 600 of the MELD-dimension text stack)
constexpr int LN_MAXSLAB = 16;  // split-K / split-F slabs summed on the fly (api.hip MAX_SPLITS)

// NC = column chunks of 64 per lane (E <= 64 NC); NSB = slabs loaded per batch.  Every global load is issued from a
// clamped (always valid) address and masked afterwards by a select: a load under a branch makes hipcc wait for it
// (s_waitcnt vmcnt(0)) before the next one is issued, which serialised ~100 loads per lane in the first version.
template <int NC, int NSB>
__global__ __launch_bounds__(256) void add_drop_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              float* __restrict__ out, float* __restrict__ xhat,
                                                              float* __restrict__ rstd, int T, int E, float eps, float p,
                                                              uint32_t site, const uint64_t* __restrict__ rng,
                                                              uint64_t add, int train, int nslab, long slab_stride) {
    const int lane = threadIdx.x & 63;
    const int rg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rg * 4 >= T) return;
    const DropCtx dc = make_drop(rng, add, site, p, train);
    size_t off[NC][4];      // element offsets (clamped)
    bool okc[NC], okr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) okr[q] = rg * 4 + q < T;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c = lane + 64 * k;
        okc[k] = c < E;
#pragma unroll
        for (int q = 0; q < 4; ++q) off[k][q] = (size_t)min(rg * 4 + q, T - 1) * E + min(c, E - 1);
    }
    float xv[NC][4], yy[NC][4];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xv[k][q] = x[off[k][q]];
            yy[k][q] = 0.f;
        }
    for (int s0 = 0; s0 < nslab; s0 += NSB) {
        float v[NSB][NC][4];
#pragma unroll
        for (int j = 0; j < NSB; ++j) {
            const float* ys = y + (size_t)min(s0 + j, nslab - 1) * slab_stride;
#pragma unroll
            for (int k = 0; k < NC; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[j][k][q] = ys[off[k][q]];
        }
#pragma unroll
        for (int j = 0; j < NSB; ++j) {
            const float m = (s0 + j < nslab) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) yy[k][q] = __fmaf_rn(m, v[j][k][q], yy[k][q]);
        }
    }
    float z[NC][4];
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        float mult[4];
        drop_mult4(dc, (uint32_t)rg, (uint32_t)E, (uint32_t)(lane + 64 * k), mult);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // explicit fma / separate roundings below: left to -ffp-contract the compiler fuses some of the four row
            // slots and not others (packed-math pairs), which makes a row's bits depend on its position in the batch
            z[k][q] = (okc[k] && okr[q]) ? __fmaf_rn(yy[k][q], mult[q], xv[k][q]) : 0.f;
            sum[q] += z[k][q];
        }
    }
    const float invE = 1.0f / (float)E;
    float mean[4], rs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mean[q] = wave_sum(sum[q]) * invE;
    float var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NC; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = okc[k] ? __fsub_rn(z[k][q], mean[q]) : 0.f;
            var[q] = __fmaf_rn(d, d, var[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) rs[q] = rsqrtf(wave_sum(var[q]) * invE + eps);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c = min(lane + 64 * k, E - 1);
        const float ww = w[c], bb = b[c];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (okc[k] && okr[q]) {
                const float xh = __fmul_rn(__fsub_rn(z[k][q], mean[q]), rs[q]);
                if (xhat) xhat[off[k][q]] = xh;
                out[off[k][q]] = __fmaf_rn(xh, ww, bb);
            }
        }
    }
    if (rstd && lane < 4 && rg * 4 + lane < T) rstd[rg * 4 + lane] = rs[lane];
}

// backward: g = d_out * w; dz = rstd * (g - mean(g) - xhat * mean(g * xhat)); dy = dz * dropmult
// gw += sum_t d_out * xhat, gb += sum_t d_out: per-block partial sums (registers -> LDS) are STORED to
// gpart[block][{w, b}][E] and summed in block order by ln_param_reduce_kernel — no atomics, so the parameter gradients do
// not depend on the order workgroups run in.  A one-block launch (gpart == nullptr) adds its sums directly.
template <int NC, int NSB>
__global__ __launch_bounds__(256) void add_drop_ln_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ xhat,
                                                              const float* __restrict__ rstd, const float* __restrict__ w,
                                                              float* __restrict__ dz, float* __restrict__ dy,
                                                              float* __restrict__ gw, float* __restrict__ gb, int T, int E,
                                                              float p, uint32_t site, const uint64_t* __restrict__ rng,
                                                              uint64_t add, int train, int nslab, long slab_stride,
                                                              const float* __restrict__ addend, float* __restrict__ gpart) {
    __shared__ float red[2][4][64 * NC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const DropCtx dc = make_drop(rng, add, site, p, train);
    const int G = (T + 3) >> 2;
    const float invE = 1.0f / (float)E;
    float aw[NC], ab[NC], wreg[NC];
    bool okc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        aw[k] = 0.f; ab[k] = 0.f;
        const int c = lane + 64 * k;
        okc[k] = c < E;
        wreg[k] = w[min(c, E - 1)];
    }
    for (int rg = blockIdx.x * 4 + wv; rg < G; rg += gridDim.x * 4) {
        size_t off[NC][4];
        bool okr[4];
        float rs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            okr[q] = rg * 4 + q < T;
            rs[q] = rstd[min(rg * 4 + q, T - 1)];
        }
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) off[k][q] = (size_t)min(rg * 4 + q, T - 1) * E + min(lane + 64 * k, E - 1);
        float d[NC][4], xh[NC][4];
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xh[k][q] = xhat[off[k][q]];
                d[k][q] = addend ? addend[off[k][q]] : 0.f;
            }
        for (int s0 = 0; s0 < nslab; s0 += NSB) {
            float v[NSB][NC][4];
#pragma unroll
            for (int j = 0; j < NSB; ++j) {
                const float* ds = d_out + (size_t)min(s0 + j, nslab - 1) * slab_stride;
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[j][k][q] = ds[off[k][q]];
            }
#pragma unroll
            for (int j = 0; j < NSB; ++j) {
                const float m = (s0 + j < nslab) ? 1.f : 0.f;
#pragma unroll
                for (int k = 0; k < NC; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[k][q] += m * v[j][k][q];
            }
        }
        float g[NC][4];
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NC; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = okc[k] && okr[q];
                const float dd = ok ? d[k][q] : 0.f, h = ok ? xh[k][q] : 0.f;
                xh[k][q] = h;
                aw[k] += dd * h;
                ab[k] += dd;
                g[k][q] = dd * wreg[k];
                s1[q] += g[k][q];
                s2[q] += g[k][q] * h;
            }
        }
        float c1[4], c2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c1[q] = wave_sum(s1[q]) * invE;
            c2[q] = wave_sum(s2[q]) * invE;
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            float mult[4] = {1.f, 1.f, 1.f, 1.f};
            if (dy) drop_mult4(dc, (uint32_t)rg, (uint32_t)E, (uint32_t)(lane + 64 * k), mult);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (okc[k] && okr[q]) {
                    const float v = rs[q] * (g[k][q] - c1[q] - xh[k][q] * c2[q]);
                    dz[off[k][q]] = v;
                    if (dy) dy[off[k][q]] = v * mult[q];
                }
            }
        }
    }
    if (gw == nullptr) return;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        red[0][wv][lane + 64 * k] = aw[k];
        red[1][wv][lane + 64 * k] = ab[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) {
        const float sw = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (gpart != nullptr) {
            gpart[((size_t)blockIdx.x * 2 + 0) * E + c] = sw;
            gpart[((size_t)blockIdx.x * 2 + 1) * E + c] = sb;
        } else {               // gridDim.x == 1 (launcher's rule): this block owns the sums
            gw[c] += sw;
            gb[c] += sb;
        }
    }
}

// gw[c] += sum_b part[b][0][c], gb[c] += sum_b part[b][1][c] for up to LN_RED_MAX LayerNorm instances per launch
// (blockIdx.y = instance), partials added in block order
constexpr int LN_RED_MAX = 32;
struct LnRedGroup {
    float* gw[LN_RED_MAX];
    float* gb[LN_RED_MAX];
    const float* part[LN_RED_MAX];
    int nblk[LN_RED_MAX];
};
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(LnRedGroup grp, int E, int overwrite) {
    // 64 columns x 16 partial groups per workgroup: thread (c, q) adds partial blocks q, q + 16, ... (independent loads, a
    // fixed order), then the 16 group sums are added in order — the association is fixed, so the result is reproducible
    __shared__ float red[2][16][64];
    const int j = blockIdx.y, cl = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const float* p = grp.part[j];
    const int nb = grp.nblk[j];
    float sw = 0.f, sb = 0.f;
    if (c < E) {
#pragma unroll 4
        for (int b = q; b < nb; b += 16) {
            sw += p[((size_t)b * 2 + 0) * E + c];
            sb += p[((size_t)b * 2 + 1) * E + c];
        }
    }
    red[0][q][cl] = sw;
    red[1][q][cl] = sb;
    __syncthreads();
    if (q == 0 && c < E) {
        float tw = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { tw += red[0][i][cl]; tb += red[1][i][cl]; }
        if (overwrite) { grp.gw[j][c] = tw; grp.gb[j][c] = tb; }
        else { grp.gw[j][c] += tw; grp.gb[j][c] += tb; }
    }
}

// ------------------------------------------------------------------------------------------
// discriminator tail: u3 = drop(a2 . w3 + b3); prob = sigmoid(u3)      model.py:1326
// and its backward fused with the fc2 activation backward:
//   d_u3 = dprob * prob*(1-prob); d_pre3 = d_u3 * m3; d_a2 = d_pre3 * w3; d_pre2 = d_a2 * gelu'(u2) * m2
// one thread per token; D2 <= 32
// ------------------------------------------------------------------------------------------
__global__ void disc_tail_fwd_kernel(const float* __restrict__ a2, const float* __restrict__ w3, const float* __restrict__ b3,
                                     float* __restrict__ prob, int T, int D2, float p, const uint64_t* __restrict__ rng,
                                     uint64_t add, int train) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const DropCtx dc = make_drop(rng, add, SITE_HEAD3, p, train);
    float acc = b3[0];
    for (int k = 0; k < D2; ++k) acc += a2[(size_t)t * D2 + k] * w3[k];
    acc *= drop_mult1(dc, (uint32_t)t, 1u, 0u);
    prob[t] = 1.0f / (1.0f + expf(-acc));
}

__global__ __launch_bounds__(256) void disc_tail_bwd_kernel(const float* __restrict__ dprob, const float* __restrict__ prob,
                                                            const float* __restrict__ a2, const float* __restrict__ u2,
                                                            const float* __restrict__ w3, float* __restrict__ d_pre2,
                                                            float* __restrict__ gw3, float* __restrict__ gb3, int T, int D2,
                                                            float p, const uint64_t* __restrict__ rng, uint64_t add,
                                                            int train, float* __restrict__ gpart) {
    __shared__ float red[4][33];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const DropCtx d3 = make_drop(rng, add, SITE_HEAD3, p, train);
    const DropCtx d2 = make_drop(rng, add, SITE_HEAD2, p, train);
    float dpre3 = 0.f;
    if (t < T) {
        const float pr = prob[t];
        dpre3 = dprob[t] * pr * (1.0f - pr) * drop_mult1(d3, (uint32_t)t, 1u, 0u);
    }
    for (int k = 0; k < D2; ++k) {
        float a = 0.f;
        if (t < T) {
            const size_t o = (size_t)t * D2 + k;
            a = a2[o];
            d_pre2[o] = dpre3 * w3[k] * gelu_grad_f(u2[o]) * drop_mult1(d2, (uint32_t)t, (uint32_t)D2, (uint32_t)k);
        }
        if (gw3) {
            const float s = wave_sum(dpre3 * a);
            if (lane == 0) red[wv][k] = s;
        }
    }
    if (gw3) {
        const float s = wave_sum(dpre3);
        if (lane == 0) red[wv][32] = s;
        __syncthreads();
        // per-block sums -> gpart[block][0..D2-1 | 32]; disc_tail_reduce_kernel adds them in block order (no atomics)
        if (threadIdx.x < D2) gpart[blockIdx.x * 36 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x == 32) gpart[blockIdx.x * 36 + 32] = red[0][32] + red[1][32] + red[2][32] + red[3][32];
    }
}
// one wave per output (gw3[0..D2-1], gb3 = slot 32): lanes stride over the blocks, fixed-order wave reduction
__global__ __launch_bounds__(1024) void disc_tail_reduce_kernel(const float* __restrict__ gpart, int nblk, int D2, float* __restrict__ gw3,
                                                                float* __restrict__ gb3) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = w; k <= D2; k += 16) {
        const int slot = k < D2 ? k : 32;
        float s = 0.f;
        for (int b = lane; b < nblk; b += 64) s += gpart[b * 36 + slot];
        s = wave_sum(s);
        if (lane == 0) {
            if (k < D2) gw3[k] += s; else gb3[0] += s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// A10: BCELoss(mean)     torch.nn.BCELoss semantics (log clamped at -100)
// ------------------------------------------------------------------------------------------
// target of element i: (i % period) < split ? target : target_b   (period 0: constant `target`).
// The batched discriminator pass holds [real | fake] dialogues side by side in the batch axis, so one
// call with period 2B, split B yields (BCE(real,1) + BCE(fake,0)) / 2 = the D loss (train_IEMOCAP.py:220-223).
// ONE workgroup of 1024 threads: the mean is a fixed-order tree (lane strides, wave shuffles, 16 wave sums in order), so the
// loss value is bit-reproducible; n = S*B (<= a few 10^4) makes a wider grid pointless.
__global__ __launch_bounds__(1024) void bce_fwd_kernel(const float* __restrict__ prob, float target_a, float target_b,
                                                       int period, int split, int n, float scale, float* __restrict__ loss,
                                                       int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float target = (period > 0 && (i % period) >= split) ? target_b : target_a;
        const float p = prob[i];
        const float lp = fmaxf(logf(p), -100.f);
        const float l1p = fmaxf(logf(1.0f - p), -100.f);
        s -= target * lp + (1.0f - target) * l1p;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        t = t * scale / (float)n;
        loss[0] = accumulate ? loss[0] + t : t;
    }
}

__global__ void bce_bwd_kernel(const float* __restrict__ prob, float target_a, float target_b, int period, int split, int n,
                               float scale, float* __restrict__ dprob) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float target = (period > 0 && (i % period) >= split) ? target_b : target_a;
    const float p = prob[i];
    dprob[i] = scale / (float)n * (p - target) / fmaxf(p * (1.0f - p), 1e-12f);
}

// ------------------------------------------------------------------------------------------
// A10: Adam, flat slab.  t = *step + 1 (the counter is bumped by adam_step_inc afterwards)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, float lr_bc1, float rs_bc2, float b1,
                                         float b2, float eps, float wd, float gscale) {
    gi *= gscale;
    if (wd != 0.f) gi += wd * pi;
    mi = b1 * mi + (1.0f - b1) * gi;
    vi = b2 * vi + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / rs_bc2 + eps;
    pi = pi - lr_bc1 * (mi / denom);
}

constexpr int ADAM_MAX_EXTRA = 7;      // token chunks beyond the first of an unreduced weight gradient (gemm_tn100.hip WMAXSPLIT - 1)
// VEC = 4: one float4 of p, g, m, v per thread (the slabs are 16-byte aligned); the n % 4 tail goes to a VEC = 1 launch.
// HBM-bound: 28 B per parameter (read p, g, m, v; write p, m, v).
template <int VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const int32_t* __restrict__ step, long n, float lr,
                                                   float b1, float b2, float eps, float wd, float gscale) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (i >= n) return;
    const float t = (float)(*step + 1);
    const float bc1 = 1.0f - powf(b1, t);
    const float bc2 = 1.0f - powf(b2, t);
    const float lr_bc1 = lr / bc1, rs_bc2 = sqrtf(bc2);
    if (VEC == 4) {
        float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
        const float4 gg = *reinterpret_cast<const float4*>(g + i);
        adam_one(pp.x, gg.x, mm.x, vv.x, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
        adam_one(pp.y, gg.y, mm.y, vv.y, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
        adam_one(pp.z, gg.z, mm.z, vv.z, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
        adam_one(pp.w, gg.w, mm.w, vv.w, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
        *reinterpret_cast<float4*>(m + i) = mm;
        *reinterpret_cast<float4*>(v + i) = vv;
        *reinterpret_cast<float4*>(p + i) = pp;
    } else {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, g[i], mi, vi, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}
// Adam over a slab whose encoder weight / bias gradients are UNREDUCED (ganffn_encoder_bwd_parts): for the covered elements —
// i < enc_floats and (i % layer_floats) < covered — the gradient is g[i] (token chunk 0) + part[0][i] + part[1][i] + ... in chunk
// order, the sum tn100_reduce_kernel would have formed (same association -> same bits); everything else (LayerNorm parameters,
// heads, `object`) is g[i] alone.  layer_floats and covered are multiples of 4: a float4 is covered or not as a whole.
__global__ __launch_bounds__(256) void adam_parts_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, const int32_t* __restrict__ step, long n, float lr,
                                                         float b1, float b2, float eps, float wd, float gscale,
                                                         const float* __restrict__ part, long part_stride, int n_extra,
                                                         long enc_floats, long layer_floats, long covered) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float t = (float)(*step + 1);
    const float bc1 = 1.0f - powf(b1, t);
    const float bc2 = 1.0f - powf(b2, t);
    const float lr_bc1 = lr / bc1, rs_bc2 = sqrtf(bc2);
    float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
    float4 gg = *reinterpret_cast<const float4*>(g + i);
    const bool cov = i < enc_floats && (i % layer_floats) < covered;           // (uniform over long runs of threads)
    if (cov) {
        float4 e[ADAM_MAX_EXTRA];
#pragma unroll
        for (int z = 0; z < ADAM_MAX_EXTRA; ++z) e[z] = *reinterpret_cast<const float4*>(part + (size_t)min(z, n_extra - 1) * part_stride + i);
#pragma unroll
        for (int z = 0; z < ADAM_MAX_EXTRA; ++z)
            if (z < n_extra) { gg.x += e[z].x; gg.y += e[z].y; gg.z += e[z].z; gg.w += e[z].w; }
    }
    adam_one(pp.x, gg.x, mm.x, vv.x, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
    adam_one(pp.y, gg.y, mm.y, vv.y, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
    adam_one(pp.z, gg.z, mm.z, vv.z, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
    adam_one(pp.w, gg.w, mm.w, vv.w, lr_bc1, rs_bc2, b1, b2, eps, wd, gscale);
    *reinterpret_cast<float4*>(m + i) = mm;
    *reinterpret_cast<float4*>(v + i) = vv;
    *reinterpret_cast<float4*>(p + i) = pp;
}
__global__ void adam_step_inc(int32_t* step) { *step += 1; }

__global__ void rng_advance_kernel(uint64_t* rng, uint64_t delta) { rng[1] += delta; }

// ------------------------------------------------------------------------------------------
// A1: PE table, fp32 exactly as the reference computes it (model.py:1182-1188):
//   div[i] = exp((2i) * (-ln(1e4)/E)) in fp32;  pe[p, 2i] = sin(p*div[i]); pe[p, 2i+1] = cos(p*div[i])
// ------------------------------------------------------------------------------------------
__global__ void pe_table_kernel(float* __restrict__ pe, int max_len, int E) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= max_len * (E / 2)) return;
    const int pos = idx / (E / 2), i = idx % (E / 2);
    const float c = (float)(-9.210340371976184 / (double)E);  // -ln(1e4)/E rounded to fp32 like python float -> tensor mul
    const float div = expf((float)(2 * i) * c);
    const float a = (float)pos * div;
    pe[(size_t)pos * E + 2 * i] = sinf(a);
    pe[(size_t)pos * E + 2 * i + 1] = cosf(a);
}

// ------------------------------------------------------------------------------------------
// A11: log_softmax over C classes + masked weighted NLL   (model.py:1448-1449, :74-81)
// acc2[0] += sum w[y] m lp[y] ; acc2[1] += sum w[y] m   then a finishing kernel divides
// ------------------------------------------------------------------------------------------
// ONE workgroup of 1024 threads striding over the tokens: the two loss sums are fixed-order trees (bit-reproducible); the
// work (S*B tokens x C <= 7 classes) is far too small for a wider grid to matter.
__global__ __launch_bounds__(1024) void logsoftmax_nll_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                              const float* __restrict__ umask, const float* __restrict__ cw,
                                                              float* __restrict__ logp, float* __restrict__ acc2, int S, int B,
                                                              int C) {
    __shared__ float red[2][16];
    float num = 0.f, den = 0.f;
    for (int t = threadIdx.x; t < S * B; t += 1024) {      // token t = s*B + b
        const int s = t / B, b = t - s * B;
        const float* x = logits + (size_t)t * C;
        float m = x[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(x[c] - m);
        const float lse = m + logf(sum);
        for (int c = 0; c < C; ++c) logp[(size_t)t * C + c] = x[c] - lse;
        if (labels) {
            const int y = (int)labels[(size_t)b * S + s];
            const float mk = umask[(size_t)b * S + s];
            const float wy = cw ? cw[y] : 1.f;
            num += wy * mk * (x[y] - lse);
            den += wy * mk;
        }
    }
    if (labels) {
        num = wave_sum(num);
        den = wave_sum(den);
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = num;
            red[1][threadIdx.x >> 6] = den;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, b = 0.f;
            for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
            acc2[0] = a;
            acc2[1] = b;
        }
    }
}
__global__ void nll_finish_kernel(const float* __restrict__ acc2, float* __restrict__ loss) { loss[0] = -acc2[0] / acc2[1]; }
// dlogits[t,c] = -(w[y] m / den) * (1[c==y] - softmax[t,c])
__global__ void nll_bwd_kernel(const float* __restrict__ logp, const int64_t* __restrict__ labels,
                               const float* __restrict__ umask, const float* __restrict__ cw, const float* __restrict__ acc2,
                               float* __restrict__ dlogits, int S, int B, int C) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * B) return;
    const int s = t / B, b = t - s * B;
    const int y = (int)labels[(size_t)b * S + s];
    const float coef = (cw ? cw[y] : 1.f) * umask[(size_t)b * S + s] / acc2[1];
    for (int c = 0; c < C; ++c) {
        const float sm = expf(logp[(size_t)t * C + c]);
        dlogits[(size_t)t * C + c] = -coef * ((c == y ? 1.f : 0.f) - sm);
    }
}

// ------------------------------------------------------------------------------------------
// naive linear for shapes the MFMA GEMM does not take (N or K not a multiple of 4: fc 100->6)
// ------------------------------------------------------------------------------------------
__global__ void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                        float* __restrict__ y, int T, int K, int N) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)T * N) return;
    const int t = (int)(idx / N), n = (int)(idx - (long)t * N);
    float acc = b ? b[n] : 0.f;
    for (int k = 0; k < K; ++k) acc += x[(size_t)t * K + k] * w[(size_t)n * K + k];
    y[idx] = acc;
}
__global__ void small_linear_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                       int T, int K, int N) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)T * K) return;
    const int t = (int)(idx / K), k = (int)(idx - (long)t * K);
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += dy[(size_t)t * N + n] * w[(size_t)n * K + k];
    dx[idx] = acc;
}
// gw[n,k] += sum_t dy[t,n] x[t,k]; gb[n] += sum_t dy[t,n].  One workgroup of 1024 threads per (n, 64 columns k): 16 token
// groups (tokens t = g, g + 16, ...) x 64 columns, independent loads 8 deep, then the 16 group sums are added in group
// order — a fixed association, so the result is reproducible.  (The first version looped over all T tokens in ONE thread
// per output: 2820 dependent load pairs = 1.37 ms for the 6 x 200 class-head gradient of configuration 5.)
__global__ __launch_bounds__(1024) void small_linear_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ gw, float* __restrict__ gb, int T, int K, int N) {
    __shared__ float red[16][65];
    const int n = blockIdx.x, kl = threadIdx.x & 63, g = threadIdx.x >> 6, k = blockIdx.y * 64 + kl;
    const int kc = min(k, K - 1);
    float acc = 0.f, accb = 0.f;
    for (int t0 = g; t0 < T; t0 += 16 * 8) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = min(t0 + 16 * u, T - 1);
            a[u] = dy[(size_t)t * N + n];
            b[u] = x[(size_t)t * K + kc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float m = (t0 + 16 * u < T) ? 1.f : 0.f;
            acc += m * a[u] * b[u];
            accb += m * a[u];
        }
    }
    red[g][kl] = acc;
    if (kl == 0) red[g][64] = accb;
    __syncthreads();
    if (g == 0) {
        float sw = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sw += red[i][kl];
        if (k < K) gw[(size_t)n * K + k] += sw;
        if (gb && kl == 0 && blockIdx.y == 0) {
            float sb = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sb += red[i][64];
            gb[n] += sb;
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline int nblk(long n, int bs) { return (int)((n + bs - 1) / bs); }

int launch_pe_dropout(const float* x, const float* pe, float* out, int S, int B, int E, float p,
                      const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    const int T = S * B;
    const long n = (long)((T + 3) / 4) * E;
    hipLaunchKernelGGL(pe_dropout_kernel, dim3(nblk(n, 256)), dim3(256), 0, st, x, pe, out, T, B, E, p, rng, add, train);
    GF_LAUNCH_CHECK();
    return 0;
}

int launch_dropout(const float* x, float* out, int R, int C, float p, uint32_t site, const uint64_t* rng,
                   uint64_t add, int train, hipStream_t st) {
    const long n = (long)((R + 3) / 4) * C;
    hipLaunchKernelGGL(drop_kernel<0>, dim3(nblk(n, 256)), dim3(256), 0, st, x, (const float*)nullptr, out, R, C, p, site, rng, add, train);
    GF_LAUNCH_CHECK();
    return 0;
}
int launch_dropout_bwd_inplace(float* dx, int R, int C, float p, uint32_t site, const uint64_t* rng, uint64_t add,
                               int train, hipStream_t st) {
    if (!(train && p > 0.f)) return 0;
    return launch_dropout(dx, dx, R, C, p, site, rng, add, train, st);
}
int launch_gelu_drop_fwd(const float* x, float* out, int R, int C, float p, uint32_t site, const uint64_t* rng,
                         uint64_t add, int train, hipStream_t st) {
    const long n = (long)((R + 3) / 4) * C;
    hipLaunchKernelGGL(drop_kernel<1>, dim3(nblk(n, 256)), dim3(256), 0, st, x, (const float*)nullptr, out, R, C, p, site, rng, add, train);
    GF_LAUNCH_CHECK();
    return 0;
}
int launch_gelu_bwd_drop(const float* d, const float* u, float* out, int R, int C, float p, uint32_t site,
                         const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    const long n = (long)((R + 3) / 4) * C;
    hipLaunchKernelGGL(drop_kernel<2>, dim3(nblk(n, 256)), dim3(256), 0, st, d, u, out, R, C, p, site, rng, add, train);
    GF_LAUNCH_CHECK();
    return 0;
}

int launch_add_inplace(float* a, const float* b, int64_t n, hipStream_t st) {
    GF_CHECK_ARG((n & 3) == 0 && aligned16(a) && aligned16(b), "add_inplace: n %% 4 and 16-byte alignment required");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(nblk(n / 4, 256)), dim3(256), 0, st, a, b, (long)(n / 4));
    GF_LAUNCH_CHECK();
    return 0;
}

int launch_add_drop_ln_fwd(const float* x, const float* y, const float* w, const float* b, float* out, float* xhat,
                           float* rstd, int T, int E, float eps, float p, uint32_t site, const uint64_t* rng,
                           uint64_t add, int train, hipStream_t st, int nslab, long slab_stride) {
    GF_CHECK_ARG(E <= 64 * LN_MAXC, "layernorm: E=%d > %d", E, 64 * LN_MAXC);
    GF_CHECK_ARG(nslab >= 1 && nslab <= LN_MAXSLAB, "layernorm: nslab=%d out of [1,%d]", nslab, LN_MAXSLAB);
    const int G = (T + 3) / 4;
    const dim3 grid((G + 3) / 4), blk(256);
#define GF_LN_FWD(NC, NSB)                                                                                              \
    hipLaunchKernelGGL((add_drop_ln_fwd_kernel<NC, NSB>), grid, blk, 0, st, x, y, w, b, out, xhat, rstd, T, E, eps, p, \
                       site, rng, add, train, nslab, slab_stride)
    if (E <= 64) GF_LN_FWD(1, 8);
    else if (E <= 128) GF_LN_FWD(2, 8);
    else if (E <= 256) GF_LN_FWD(4, 4);
    else if (E <= 320) GF_LN_FWD(5, 4);
    else if (E <= 512) GF_LN_FWD(8, 2);
    else GF_LN_FWD(10, 2);
#undef GF_LN_FWD
    GF_LAUNCH_CHECK();
    return 0;
}

int ln_bwd_blocks(int T) {
    int blocks = ((T + 3) / 4 + 3) / 4;
    return blocks > 256 ? 256 : blocks;
}

// gw / gb (when non-NULL): with a partial buffer gpart (>= ln_bwd_blocks(T) * 2 * E floats) the launch is full-width and
// the caller must run launch_ln_param_reduce afterwards; without one a single workgroup does the whole tensor and adds
// its sums directly (unit-test hook; still deterministic).
int launch_add_drop_ln_bwd(const float* d_out, const float* xhat, const float* rstd, const float* w, float* dz,
                           float* dy, float* gw, float* gb, int T, int E, float p, uint32_t site,
                           const uint64_t* rng, uint64_t add, int train, hipStream_t st, int nslab, long slab_stride,
                           const float* addend, float* gpart) {
    GF_CHECK_ARG(E <= 64 * LN_MAXC, "layernorm: E=%d > %d", E, 64 * LN_MAXC);
    GF_CHECK_ARG(nslab >= 1 && nslab <= LN_MAXSLAB, "layernorm: nslab=%d out of [1,%d]", nslab, LN_MAXSLAB);
    int blocks = ln_bwd_blocks(T);
    if (gw != nullptr && gpart == nullptr) blocks = 1;
    if (gw == nullptr) gpart = nullptr;
    const dim3 grid(blocks), blk(256);
#define GF_LN_BWD(NC, NSB)                                                                                            \
    hipLaunchKernelGGL((add_drop_ln_bwd_kernel<NC, NSB>), grid, blk, 0, st, d_out, xhat, rstd, w, dz, dy, gw, gb, T, \
                       E, p, site, rng, add, train, nslab, slab_stride, addend, gpart)
    if (E <= 64) GF_LN_BWD(1, 8);
    else if (E <= 128) GF_LN_BWD(2, 8);
    else if (E <= 256) GF_LN_BWD(4, 4);
    else if (E <= 320) GF_LN_BWD(5, 4);
    else if (E <= 512) GF_LN_BWD(8, 2);
    else GF_LN_BWD(10, 2);
#undef GF_LN_BWD
    GF_LAUNCH_CHECK();
    return 0;
}

// add the per-block partial sums of n LayerNorm backward launches (gw[i], gb[i] += sum over nblk[i] blocks of part[i])
int launch_ln_param_reduce(int n, float* const* gw, float* const* gb, const float* const* part, const int* nblk, int E,
                           hipStream_t st, bool overwrite) {
    for (int i0 = 0; i0 < n; i0 += LN_RED_MAX) {
        LnRedGroup grp;
        const int m = n - i0 < LN_RED_MAX ? n - i0 : LN_RED_MAX;
        for (int i = 0; i < m; ++i) {
            grp.gw[i] = gw[i0 + i]; grp.gb[i] = gb[i0 + i]; grp.part[i] = part[i0 + i]; grp.nblk[i] = nblk[i0 + i];
        }
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((E + 63) / 64, m), dim3(1024), 0, st, grp, E, overwrite ? 1 : 0);
        GF_LAUNCH_CHECK();
    }
    return 0;
}

int launch_disc_tail_fwd(const float* a2, const float* w3, const float* b3, float* prob, int T, int D2, float p,
                         const uint64_t* rng, uint64_t add, int train, hipStream_t st) {
    hipLaunchKernelGGL(disc_tail_fwd_kernel, dim3(nblk(T, 256)), dim3(256), 0, st, a2, w3, b3, prob, T, D2, p, rng, add, train);
    GF_LAUNCH_CHECK();
    return 0;
}
// gpart: nblk(T, 256) * 36 floats of per-block partial sums (required when gw3 != NULL)
int launch_disc_tail_bwd(const float* dprob, const float* prob, const float* a2, const float* u2, const float* w3,
                         float* d_pre2, float* gw3, float* gb3, int T, int D2, float p, const uint64_t* rng, uint64_t add,
                         int train, hipStream_t st, float* gpart) {
    GF_CHECK_ARG(D2 <= 32, "disc tail: D2=%d > 32", D2);
    GF_CHECK_ARG(gw3 == nullptr || (gpart != nullptr && gb3 != nullptr), "disc tail: fc3 gradients need gb3 and the partial-sum workspace");
    const int nb = nblk(T, 256);
    hipLaunchKernelGGL(disc_tail_bwd_kernel, dim3(nb), dim3(256), 0, st, dprob, prob, a2, u2, w3, d_pre2, gw3, gb3, T,
                       D2, p, rng, add, train, gpart);
    GF_LAUNCH_CHECK();
    if (gw3 != nullptr) {
        hipLaunchKernelGGL(disc_tail_reduce_kernel, dim3(1), dim3(1024), 0, st, (const float*)gpart, nb, D2, gw3, gb3);
        GF_LAUNCH_CHECK();
    }
    return 0;
}

// gw3[k] += sum_b gpart[b][k], gb3 += sum_b gpart[b][32] (partial sums in block order)
int launch_disc_tail_reduce(const float* gpart, int nblk_, int D2, float* gw3, float* gb3, hipStream_t st) {
    GF_CHECK_ARG(gpart && gw3 && gb3 && D2 <= 32 && nblk_ >= 1, "disc tail reduce: bad arguments");
    hipLaunchKernelGGL(disc_tail_reduce_kernel, dim3(1), dim3(1024), 0, st, gpart, nblk_, D2, gw3, gb3);
    GF_LAUNCH_CHECK();
    return 0;
}

int launch_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int T, int K, int N, hipStream_t st) {
    hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(nblk((long)T * N, 256)), dim3(256), 0, st, x, w, b, y, T, K, N);
    GF_LAUNCH_CHECK();
    return 0;
}
int launch_small_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* gw, float* gb, int T, int K,
                            int N, hipStream_t st) {
    if (dx) {
        hipLaunchKernelGGL(small_linear_dx_kernel, dim3(nblk((long)T * K, 256)), dim3(256), 0, st, dy, w, dx, T, K, N);
        GF_LAUNCH_CHECK();
    }
    if (gw) {
        hipLaunchKernelGGL(small_linear_dw_kernel, dim3(N, (K + 63) / 64), dim3(1024), 0, st, dy, x, gw, gb, T, K, N);
        GF_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace ganffn

// ==========================================================================================
// C ABI entry points that are pure elementwise work
// ==========================================================================================
using namespace ganffn;

extern "C" int ganffn_rng_advance(uint64_t* rng, uint64_t delta, void* stream) {
    GF_CHECK_ARG(rng, "rng_advance: null rng");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng, delta);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_pe_table(float* pe, int max_len, int E, void* stream) {
    GF_CHECK_ARG(pe && max_len > 0 && E > 0 && (E & 1) == 0, "pe_table: E must be even (model.py:1187-1188)");
    const int n = max_len * (E / 2);
    hipLaunchKernelGGL(pe_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pe, max_len, E);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_bce_fwd(const float* prob, float target, int n, float scale, float* loss_out, int accumulate,
                              void* stream) {
    return ganffn_bce2_fwd(prob, target, target, 0, 0, n, scale, loss_out, accumulate, stream);
}
extern "C" int ganffn_bce_bwd(const float* prob, float target, int n, float scale, float* dprob, void* stream) {
    return ganffn_bce2_bwd(prob, target, target, 0, 0, n, scale, dprob, stream);
}

extern "C" int ganffn_bce2_fwd(const float* prob, float target_a, float target_b, int period, int split, int n, float scale,
                               float* loss_out, int accumulate, void* stream) {
    GF_CHECK_ARG(prob && loss_out && n > 0 && period >= 0 && split >= 0, "bce_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(1024), 0, st, prob, target_a, target_b, period, split, n, scale, loss_out,
                       accumulate);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_bce2_bwd(const float* prob, float target_a, float target_b, int period, int split, int n, float scale,
                               float* dprob, void* stream) {
    GF_CHECK_ARG(prob && dprob && n > 0 && period >= 0 && split >= 0, "bce_bwd: bad arguments");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, prob, target_a, target_b, period, split, n, scale, dprob);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_adam_bump(int32_t* step, void* stream) {
    GF_CHECK_ARG(step, "adam_bump: null step");
    hipLaunchKernelGGL(adam_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t* step,
                                int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                float grad_scale, void* stream) {
    GF_TRY(ganffn_adam_update(params, grads, exp_avg, exp_avg_sq, step, n, lr, beta1, beta2, eps, weight_decay, grad_scale, stream));
    return ganffn_adam_bump(step, stream);
}

extern "C" int ganffn_adam_update(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int32_t* step,
                                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  float grad_scale, void* stream) {
    GF_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && step && n > 0, "adam_update: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = aligned16(params) && aligned16(grads) && aligned16(exp_avg) && aligned16(exp_avg_sq);
    const long n4 = vec ? ((long)n & ~3L) : 0;
    if (n4 > 0) {
        hipLaunchKernelGGL(adam_kernel<4>, dim3((unsigned)((n4 / 4 + 255) / 256)), dim3(256), 0, st, params, grads, exp_avg,
                           exp_avg_sq, (const int32_t*)step, n4, lr, beta1, beta2, eps, weight_decay, grad_scale);
        GF_LAUNCH_CHECK();
    }
    if (n4 < n) {
        hipLaunchKernelGGL(adam_kernel<1>, dim3((unsigned)((n - n4 + 255) / 256)), dim3(256), 0, st, params + n4, grads + n4,
                           exp_avg + n4, exp_avg_sq + n4, (const int32_t*)step, (long)n - n4, lr, beta1, beta2, eps,
                           weight_decay, grad_scale);
        GF_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int ganffn_adam_step_parts(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int32_t* step, int64_t n,
                                      float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                      const float* parts, int64_t part_stride, int n_parts, int64_t enc_floats, int64_t layer_floats,
                                      int64_t covered_per_layer, void* stream) {
    GF_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && step && n > 0, "adam_step_parts: bad arguments");
    GF_CHECK_ARG(n_parts >= 1 && n_parts <= ADAM_MAX_EXTRA + 1, "adam_step_parts: n_parts=%d out of [1,%d]", n_parts, ADAM_MAX_EXTRA + 1);
    if (n_parts == 1) return ganffn_adam_step(params, grads, exp_avg, exp_avg_sq, step, n, lr, beta1, beta2, eps, weight_decay, grad_scale, stream);
    GF_CHECK_ARG(parts && aligned16(parts) && (part_stride & 3) == 0 && part_stride >= enc_floats, "adam_step_parts: bad partial slabs");
    GF_CHECK_ARG((n & 3) == 0 && aligned16(params) && aligned16(grads) && aligned16(exp_avg) && aligned16(exp_avg_sq),
                 "adam_step_parts: slabs must be 16-byte aligned and a multiple of 4 floats");
    GF_CHECK_ARG(layer_floats > 0 && (layer_floats & 3) == 0 && (covered_per_layer & 3) == 0 && covered_per_layer <= layer_floats &&
                     enc_floats % layer_floats == 0 && enc_floats <= n, "adam_step_parts: bad layer geometry");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_parts_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq,
                       (const int32_t*)step, (long)n, lr, beta1, beta2, eps, weight_decay, grad_scale, parts, (long)part_stride,
                       n_parts - 1, (long)enc_floats, (long)layer_floats, (long)covered_per_layer);
    GF_LAUNCH_CHECK();
    return ganffn_adam_bump(step, stream);
}

extern "C" int ganffn_add3(const float* a, const float* b, const float* c, float* out, int64_t n, void* stream) {
    GF_CHECK_ARG(a && b && c && out && n > 0, "add3: bad arguments");
    hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, (long)n);
    GF_LAUNCH_CHECK();
    return 0;
}

extern "C" int ganffn_logsoftmax_nll(const float* logits, const int64_t* labels, const float* umask, const float* class_w,
                                     float* log_prob, float* loss_out, float* dlogits, float* workspace2, int S, int B, int C,
                                     void* stream) {
    GF_CHECK_ARG(logits && log_prob && S > 0 && B > 0 && C > 0, "logsoftmax_nll: bad arguments");
    GF_CHECK_ARG(!labels || (umask && loss_out && workspace2), "logsoftmax_nll: labels need umask, loss_out, workspace2");
    hipStream_t st = (hipStream_t)stream;
    const int T = S * B;
    hipLaunchKernelGGL(logsoftmax_nll_kernel, dim3(1), dim3(1024), 0, st, logits, labels, umask, class_w, log_prob,
                       workspace2, S, B, C);
    GF_LAUNCH_CHECK();
    if (labels) {
        hipLaunchKernelGGL(nll_finish_kernel, dim3(1), dim3(1), 0, st, (const float*)workspace2, loss_out);
        GF_LAUNCH_CHECK();
        if (dlogits) {
            hipLaunchKernelGGL(nll_bwd_kernel, dim3((T + 255) / 256), dim3(256), 0, st, (const float*)log_prob, labels, umask,
                               class_w, (const float*)workspace2, dlogits, S, B, C);
            GF_LAUNCH_CHECK();
        }
    }
    return 0;
}

extern "C" int ganffn_dropout(const float* x, float* out, int R, int C, float p, uint32_t site, const uint64_t* rng,
                              uint64_t rng_offset_add, void* stream) {
    GF_CHECK_ARG(x && out && R > 0 && C > 0, "dropout: bad arguments");
    GF_CHECK_ARG(p <= 0.f || rng, "dropout: rng required");
    return launch_dropout(x, out, R, C, p, site, rng, rng_offset_add, 1, (hipStream_t)stream);
}

extern "C" int ganffn_add_dropout_layernorm_fwd(const float* x, const float* y, const float* w, const float* b, float* out,
                                                float* xhat, float* rstd, int T, int E, float eps, float p, uint32_t site,
                                                const uint64_t* rng, uint64_t rng_offset_add, void* stream) {
    GF_CHECK_ARG(x && y && w && b && out && T > 0 && E > 0, "add_dropout_layernorm_fwd: bad arguments");
    GF_CHECK_ARG(p <= 0.f || rng, "add_dropout_layernorm_fwd: rng required");
    return launch_add_drop_ln_fwd(x, y, w, b, out, xhat, rstd, T, E, eps, p, site, rng, rng_offset_add, 1, (hipStream_t)stream);
}

extern "C" int ganffn_add_dropout_layernorm_bwd(const float* d_out, const float* xhat, const float* rstd, const float* w,
                                                float* dz, float* dy, float* gw, float* gb, int T, int E, float p,
                                                uint32_t site, const uint64_t* rng, uint64_t rng_offset_add, void* stream) {
    GF_CHECK_ARG(d_out && xhat && rstd && w && dz && T > 0 && E > 0, "add_dropout_layernorm_bwd: bad arguments");
    GF_CHECK_ARG(p <= 0.f || rng, "add_dropout_layernorm_bwd: rng required");
    return launch_add_drop_ln_bwd(d_out, xhat, rstd, w, dz, dy, gw, gb, T, E, p, site, rng, rng_offset_add, 1,
                                  (hipStream_t)stream, 1, 0, nullptr, nullptr);
}
