// ffn.hip — fused feed-forward block of the encoder layer for d_model = 100 (acoustic / text generators and all
// three discriminators: 26 of the 30 stack-forwards and 22 of the 24 stack-backwards of one GAN iteration).
//
// Reference op (torch nn.TransformerEncoderLayer._ff_block, call site /root/reference/model.py:1210):
//     y = linear2(dropout(relu(linear1(x))))          x [T x 100], hidden [T x 2048], y [T x 100]
// and its input-gradient chain  dh = (dy W2) * relu'/dropout mask,  dx = dh W1.
//
// Why fused: unfused, the [T x 2048] hidden makes an HBM round trip between two GEMMs and the second GEMM
// (N = 100, K = 2048) has only T/64 x 2 output tiles for 256 CUs.  Here both products run TRANSPOSED with the
// token on the MFMA lane axis, so the first product's accumulator tile (32 hidden units x 32 tokens) IS the B
// operand of the second product — the hidden activations never leave the registers between the two GEMMs
// (they are also streamed out once, because the weight-gradient GEMMs need h and dh).
//     GEMM1   hT[f][t]  = sum_e  W1[f][e] x[t][e]       A = W1 tile (LDS),  B = x fragments (registers, loaded once)
//     GEMM2   yT[e][t] += sum_f  W2[e][f] hT[f][t]      A = W2 tile (LDS),  B = GEMM1's accumulator registers
// The backward dgrad pass is the same skeleton with the two weight tiles read in the other orientation
// (dhT = W2^T dy^T, masked by the saved h; dxT += W1^T dhT), so no transposed weight copies exist.
// Work split: workgroup = 4 waves = 128 tokens x (F / FSPLIT) hidden units; weight tiles are staged once per
// workgroup in LDS (double-buffered, register-staged) and shared by the 4 waves; the FSPLIT partial outputs are
// written as slabs and summed by the LayerNorm kernel that consumes them (no atomics, deterministic).
// All MFMAs are v_mfma_f32_32x32x2_f32 (exact fp32): 52 + 64 per (32 tokens x 32 hidden units).
#include "common.h"

namespace ganffn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int FE = 100;        // d_model handled by this kernel
constexpr int KG = 13;         // groups of 8 along e (104 >= 100)
constexpr int ET = 4;          // 32-wide tiles covering e (128 >= 100)
constexpr int S1 = 108;        // LDS row stride of the W1 tile [32 f][100 e]  (108 = 4 * 27: b128 row reads conflict-free)
constexpr int S2 = 36;         // LDS row stride of the W2 tile [128 e][32 f]  (36 = 4 * 9)
constexpr int W1_FLOATS = 32 * S1;
constexpr int W2_FLOATS = 128 * S2;
constexpr int STAGE = W1_FLOATS + W2_FLOATS + 32;   // + b1 tile

__device__ __forceinline__ int krow(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }
}  // namespace

struct FfnArgs {
    const float* x;      // fwd: x [T x E];  bwd: dy [T x E]
    const float* w1;     // [F x E]
    const float* b1;     // [F]       (fwd)
    const float* w2;     // [E x F]
    const float* b2;     // [E]       (fwd, added by split 0)
    float* h;            // fwd: out (may be null: nothing kept);  bwd: in (saved post-ReLU/dropout hidden)
    float* dh;           // bwd: out [T x F]
    float* slabs;        // [FSPLIT][T x E] partial outputs
    long slab_stride;
    int T, F;
    float mscale;        // bwd: 1/(1-p) if dropout was active else 1
    float p; uint32_t site; const uint64_t* rng; uint64_t rng_add; int train;   // fwd dropout
};

template <int BWD>
__global__ __launch_bounds__(256) void ffn_fused_kernel(FfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int t = blockIdx.x * 128 + w * 32 + r;
    const bool tok = t < a.T;
    const int nft = a.F / 32 / gridDim.y;       // f-tiles of this workgroup
    const int ft0 = blockIdx.y * nft;

    // zero the LDS pad that GEMM1 reads as k = 100..103 (W1 tile columns 100..107; W2 tile rows 100..127)
    for (int i = tid; i < 2 * 32 * 8; i += 256) {
        const int buf = i / 256, rem = i % 256;
        smem[buf * STAGE + (rem >> 3) * S1 + 100 + (rem & 7)] = 0.f;
    }
    for (int i = tid; i < 2 * 28 * S2; i += 256) {
        const int buf = i / (28 * S2), rem = i % (28 * S2);
        smem[buf * STAGE + W1_FLOATS + 100 * S2 + rem] = 0.f;
    }

    // B operand of GEMM1: this lane's token row, k = 8g + 4h + j
    float xf[KG][4];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        const int k = 8 * g + 4 * h;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tok && k < FE) v = *reinterpret_cast<const float4*>(a.x + (size_t)t * FE + k);
        xf[g][0] = v.x; xf[g][1] = v.y; xf[g][2] = v.z; xf[g][3] = v.w;
    }

    floatx16 acc2[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[e][i] = 0.f;

    // register staging of one weight tile pair: 800 + 800 float4 over 256 threads
    float4 s1[4], s2[4];
    float sb = 0.f;
    auto gload = [&](int ft) {
        const int f0 = ft * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            s1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            s2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < 800) {
                s1[j] = *reinterpret_cast<const float4*>(a.w1 + (size_t)(f0 + i / 25) * FE + (i % 25) * 4);
                s2[j] = *reinterpret_cast<const float4*>(a.w2 + (size_t)(i >> 3) * a.F + f0 + (i & 7) * 4);
            }
        }
        if (!BWD && tid < 32) sb = a.b1[f0 + tid];
    };
    auto sstore = [&](int buf) {
        float* W1s = smem + buf * STAGE;
        float* W2s = W1s + W1_FLOATS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            if (i < 800) {
                *reinterpret_cast<float4*>(W1s + (i / 25) * S1 + (i % 25) * 4) = s1[j];
                *reinterpret_cast<float4*>(W2s + (i >> 3) * S2 + (i & 7) * 4) = s2[j];
            }
        }
        if (!BWD && tid < 32) W2s[W2_FLOATS + tid] = sb;
    };

    DropCtx dc;
    if (!BWD) dc = make_drop(a.rng, a.rng_add, a.site, a.p, a.train);

    gload(ft0);
    sstore(0);
    __syncthreads();

    for (int j = 0; j < nft; ++j) {
        const int f0 = (ft0 + j) * 32;
        if (j + 1 < nft) gload(ft0 + j + 1);
        const float* W1s = smem + (j & 1) * STAGE;
        const float* W2s = W1s + W1_FLOATS;
        const float* b1s = W2s + W2_FLOATS;

        // ---- GEMM1: hidden tile (32 hidden units x 32 tokens), K = 100 (13 groups of 8, last half zero)
        floatx16 acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            float af[4];
            if (!BWD) {
                const float4 q = *reinterpret_cast<const float4*>(W1s + r * S1 + 8 * g + 4 * h);
                af[0] = q.x; af[1] = q.y; af[2] = q.z; af[3] = q.w;
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) af[jj] = W2s[(8 * g + 4 * h + jj) * S2 + r];
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj], xf[g][jj], acc1, 0, 0, 0);
        }

        // ---- epilogue 1 in registers: register i <-> hidden unit f0 + krow(i, h), this lane's token
        if (!BWD) {
            uint32_t mq[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
            if (dc.on) {
                // one Philox call = 4 consecutive tokens (the 4 lanes of a quad) at one hidden unit
                uint32_t mine = 0;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int f = f0 + krow(4 * gq + (lane & 3), h);
                    uint32_t wd[4];
                    philox4((uint32_t)(t >> 2) * (uint32_t)a.F + (uint32_t)f, dc.site, dc.o0, dc.o1, dc.k0, dc.k1, wd);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (wd[q] >= dc.thr) mine |= 1u << (gq * 4 + q);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) mq[q] = __shfl(mine, (lane & ~3) | q, 64);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = fmaxf(acc1[i] + b1s[krow(i, h)], 0.f);
                const bool keep = (mq[i & 3] >> ((i >> 2) * 4 + (lane & 3))) & 1u;
                acc1[i] = keep ? v * dc.scale : 0.f;
            }
            if (a.h != nullptr && tok) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(a.h + (size_t)t * a.F + f0 + 8 * gq + 4 * h) =
                        make_float4(acc1[4 * gq], acc1[4 * gq + 1], acc1[4 * gq + 2], acc1[4 * gq + 3]);
            }
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tok) hv = *reinterpret_cast<const float4*>(a.h + (size_t)t * a.F + f0 + 8 * gq + 4 * h);
                acc1[4 * gq + 0] = hv.x > 0.f ? acc1[4 * gq + 0] * a.mscale : 0.f;
                acc1[4 * gq + 1] = hv.y > 0.f ? acc1[4 * gq + 1] * a.mscale : 0.f;
                acc1[4 * gq + 2] = hv.z > 0.f ? acc1[4 * gq + 2] * a.mscale : 0.f;
                acc1[4 * gq + 3] = hv.w > 0.f ? acc1[4 * gq + 3] * a.mscale : 0.f;
                if (tok)
                    *reinterpret_cast<float4*>(a.dh + (size_t)t * a.F + f0 + 8 * gq + 4 * h) =
                        make_float4(acc1[4 * gq], acc1[4 * gq + 1], acc1[4 * gq + 2], acc1[4 * gq + 3]);
            }
        }

        // ---- GEMM2: yT[e][t] += sum over this tile's 32 hidden units; B operand = acc1 registers
#pragma unroll
        for (int e = 0; e < ET; ++e) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float af[4];
                if (!BWD) {
                    const float4 q = *reinterpret_cast<const float4*>(W2s + (32 * e + r) * S2 + 8 * gq + 4 * h);
                    af[0] = q.x; af[1] = q.y; af[2] = q.z; af[3] = q.w;
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) af[jj] = W1s[krow(4 * gq + jj, h) * S1 + 32 * e + r];
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc2[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj], acc1[4 * gq + jj], acc2[e], 0, 0, 0);
            }
        }

        if (j + 1 < nft) sstore((j + 1) & 1);
        __syncthreads();
    }

    // ---- partial output slab of this F split: register i of tile e <-> column 32e + krow(i, h), this lane's token
    if (tok) {
        float* out = a.slabs + (size_t)blockIdx.y * a.slab_stride + (size_t)t * FE;
#pragma unroll
        for (int e = 0; e < ET; ++e) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int e0 = 32 * e + 8 * gq + 4 * h;
                if (e0 < FE) {
                    float4 v = make_float4(acc2[e][4 * gq], acc2[e][4 * gq + 1], acc2[e][4 * gq + 2], acc2[e][4 * gq + 3]);
                    if (!BWD && blockIdx.y == 0) {
                        const float4 bb = *reinterpret_cast<const float4*>(a.b2 + e0);
                        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                    }
                    *reinterpret_cast<float4*>(out + e0) = v;
                }
            }
        }
    }
}

bool ffn_fused_supported(int E, int F) { return E == FE && F >= 512 && (F % 512) == 0; }

// number of F splits (= slabs written): 16 keeps >= ~1.4 workgroups per CU at T = 3008
int ffn_fused_splits(int T, int F) {
    int s = 16;
    while (s > 1 && (F / 32) % s != 0) s >>= 1;
    return s;
}

static int launch_ffn(const FfnArgs& a, int splits, int bwd, hipStream_t st) {
    const size_t lds = 2 * STAGE * sizeof(float);
    dim3 grid((a.T + 127) / 128, splits);
    if (bwd) {
        GF_TRY((lds_optin<ffn_fused_kernel<1>>(lds, "ffn")));
        hipLaunchKernelGGL(ffn_fused_kernel<1>, grid, dim3(256), lds, st, a);
    } else {
        GF_TRY((lds_optin<ffn_fused_kernel<0>>(lds, "ffn")));
        hipLaunchKernelGGL(ffn_fused_kernel<0>, grid, dim3(256), lds, st, a);
    }
    GF_LAUNCH_CHECK();
    return 0;
}

int launch_ffn_fused_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* h,
                         float* slabs, long slab_stride, int T, int E, int F, float p, uint32_t site, const uint64_t* rng,
                         uint64_t add, int train, int* splits_out, hipStream_t st) {
    GF_CHECK_ARG(ffn_fused_supported(E, F), "ffn_fused: unsupported E=%d F=%d", E, F);
    GF_CHECK_ARG(x && w1 && b1 && w2 && b2 && slabs && T > 0, "ffn_fused_fwd: null pointer");
    GF_CHECK_ARG(aligned16(x) && aligned16(w1) && aligned16(w2) && aligned16(b2) && aligned16(slabs) && (!h || aligned16(h)),
                 "ffn_fused_fwd: 16-byte alignment required");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "ffn_fused_fwd: rng required when dropout is active");
    const int splits = ffn_fused_splits(T, F);
    FfnArgs a{x, w1, b1, w2, b2, h, nullptr, slabs, slab_stride, T, F, 1.f, p, site, rng, add, train};
    *splits_out = splits;
    return launch_ffn(a, splits, 0, st);
}

int launch_ffn_fused_bwd(const float* dy, const float* w1, const float* w2, const float* h, float* dh, float* slabs,
                         long slab_stride, int T, int E, int F, float mscale, int* splits_out, hipStream_t st) {
    GF_CHECK_ARG(ffn_fused_supported(E, F), "ffn_fused: unsupported E=%d F=%d", E, F);
    GF_CHECK_ARG(dy && w1 && w2 && h && dh && slabs && T > 0, "ffn_fused_bwd: null pointer");
    GF_CHECK_ARG(aligned16(dy) && aligned16(w1) && aligned16(w2) && aligned16(h) && aligned16(dh) && aligned16(slabs),
                 "ffn_fused_bwd: 16-byte alignment required");
    const int splits = ffn_fused_splits(T, F);
    FfnArgs a{dy, w1, nullptr, w2, nullptr, const_cast<float*>(h), dh, slabs, slab_stride, T, F, mscale, 0.f, 0, nullptr, 0, 0};
    *splits_out = splits;
    return launch_ffn(a, splits, 1, st);
}

}  // namespace ganffn
