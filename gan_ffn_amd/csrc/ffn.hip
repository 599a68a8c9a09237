// ffn.hip — fused feed-forward block of the encoder layer for d_model = 100 (acoustic / text generators and all
// three discriminators: 26 of the 30 stack-forwards and 22 of the 24 stack-backwards of one GAN iteration).
//
// Reference op (torch nn.TransformerEncoderLayer._ff_block, call site /root/reference/model.py:1210):
//     y = linear2(dropout(relu(linear1(x))))          x [T x 100], hidden [T x 2048], y [T x 100]
// and its input-gradient chain  dh = (dy W2) * relu'/dropout mask,  dx = dh W1.
//
// Why fused: unfused, the [T x 2048] hidden makes an HBM round trip between two GEMMs, the first GEMM (K = 100) is
// seven short K tiles per workgroup with every latency exposed, and the second (N = 100, K = 2048) has T/64 x 2 output
// tiles for 256 CUs.  Here both products run TRANSPOSED with the token on the MFMA lane axis, so the first product's
// accumulator tile (32 hidden units x 32 tokens) IS the B operand of the second product — the hidden activations never
// leave the registers between the two GEMMs (they are streamed out once, because the weight-gradient GEMMs need h / dh).
//     GEMM1   hT[f][t]  = sum_e  W1[f][e] x[t][e]       A = W1 fragments,  B = x fragments (LDS, staged once per workgroup)
//     GEMM2   yT[e][t] += sum_f  W2[e][f] hT[f][t]      A = W2 fragments,  B = GEMM1's accumulator registers
// The backward dgrad pass is the same skeleton with the two weights in the other orientation
// (dhT = W2^T dy^T, masked by the saved h; dxT += W1^T dhT).
//
// Round-2 structure (the round-1 kernel staged weight tiles through LDS with a barrier per tile and measured no faster
// than the two GEMMs):
//   * WEIGHTS ARE PRE-PACKED in MFMA fragment order (ffn_pack_kernel, once per encoder pass, ~15 MB): every A operand
//     is one fully coalesced 16-byte-per-lane global load straight into registers — no LDS staging, no layout
//     transformation and NO BARRIER in the main loop, so the two waves per SIMD drift apart and one wave's epilogue
//     (bias / ReLU / Philox / stores: VALU) runs under the other's MFMAs;
//   * a workgroup is 2 waves that share one 32-token tile (x staged once in LDS) and split its hidden range; their
//     partial outputs are summed through LDS at the end, so a launch writes 16 (T <= 4096) or 8 partial slabs, which the
//     LayerNorm kernel that consumes them sums on the fly (no atomics, deterministic);
//   * work split for balance: one unit = 32 tokens x 32 hidden units = 116 MFMAs; T = 3008 gives 6016 units = 3008
//     waves of 2 units (2.94 per SIMD), T = 6016 gives 3008 waves of 4 units — dispatched workgroup by workgroup, so the
//     SIMDs stay within a few % of evenly loaded.
// All MFMAs are v_mfma_f32_32x32x2_f32 (exact fp32): 52 + 64 per unit.
#include "common.h"

namespace ganffn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int FE = 100;        // d_model handled by this kernel
constexpr int KG = 13;         // groups of 8 along e (104 >= 100)
constexpr int ET = 4;          // 32-wide tiles covering e (128 >= 100)
constexpr int XS = 108;        // LDS row stride of the x tile [32 tokens][100 e] (108 = 4 * 27: b128 row reads conflict-free)
constexpr int G1_VEC = KG * 64;            // float4 per f-tile of the GEMM1 A fragments
constexpr int G2_VEC = ET * 4 * 64;        // float4 per f-tile of the GEMM2 A fragments
constexpr int UNIT_VEC = G1_VEC + G2_VEC;  // 1856 float4 = 29,696 bytes per f-tile

__device__ __forceinline__ int krow(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }
}  // namespace

// ------------------------------------------------------------------------------------------
// weight packing: [layer][f-tile]{ G1 fragments [g][lane] | G2 fragments [et][gq][lane] } as float4
//   forward :  G1 = W1[32ft + r][8g + 4h + j]              G2 = W2[32et + r][32ft + 8gq + 4h + j]
//   backward:  G1 = W2[8g + 4h + j][32ft + r]              G2 = W1[32ft + 8gq + 4h + j][32et + r]
// (lane = 32h + r; j = component; entries with an e index >= 100 are zero)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ w1_base, const float* __restrict__ w2_base,
                                                       long layer_stride, float4* __restrict__ packed, int F, int bwd,
                                                       int nvec_layer) {
    const int layer = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec_layer) return;
    const float* w1 = w1_base + (size_t)layer * layer_stride;   // [F x 100]
    const float* w2 = w2_base + (size_t)layer * layer_stride;   // [100 x F]
    const int ft = v / UNIT_VEC, u = v - ft * UNIT_VEC;
    const int lane = u & 63, r = lane & 31, h = lane >> 5;
    float o[4];
    if (u < G1_VEC) {
        const int g = u >> 6;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 8 * g + 4 * h + j, f = 32 * ft + r;
            o[j] = e < FE ? (bwd ? w2[(size_t)e * F + f] : w1[(size_t)f * FE + e]) : 0.f;
        }
    } else {
        const int q = (u - G1_VEC) >> 6, et = q >> 2, gq = q & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 32 * et + r, f = 32 * ft + 8 * gq + 4 * h + j;
            o[j] = e < FE ? (bwd ? w1[(size_t)f * FE + e] : w2[(size_t)e * F + f]) : 0.f;
        }
    }
    packed[(size_t)layer * nvec_layer + v] = make_float4(o[0], o[1], o[2], o[3]);
}

struct FfnArgs {
    const float* x;        // fwd: x [T x E];  bwd: dy [T x E]
    const float4* packed;  // this layer's packed weights (ffn_pack_kernel)
    const float* b1;       // [F]       (fwd)
    const float* b2;       // [E]       (fwd, added by slab 0)
    float* h;              // fwd: out (may be null: nothing kept);  bwd: in (saved post-ReLU/dropout hidden)
    float* dh;             // bwd: out [T x F]
    float* slabs;          // [gridDim.y][T x E] partial outputs
    long slab_stride;
    int T, F, units;       // units = f-tiles per wave
    float mscale;          // bwd: 1/(1-p) if dropout was active else 1
    float p; uint32_t site; const uint64_t* rng; uint64_t rng_add; int train;   // fwd dropout
};

template <int BWD>
__global__ __launch_bounds__(128, 2) void ffn_fused_kernel(FfnArgs a) {
    // x tile [32][XS] (13.8 KB), later the 16 KB reduce buffer | linear1 bias of this workgroup's hidden range (<= 256)
    __shared__ __attribute__((aligned(16))) float smem[ET * 16 * 64 + 256];
    float* const bias_s = smem + ET * 16 * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int t = blockIdx.x * 32 + r;
    const bool tok = t < a.T;
    const int ft0 = (blockIdx.y * 2 + w) * a.units;

    // stage the x tile (32 tokens x 100, zero-padded to 108 columns; rows beyond T zero): all loads are issued from
    // clamped addresses before the first LDS write (a load under a branch is waited for on the spot)
    {
        constexpr int NV = 32 * (XS / 4), NJ = (NV + 127) / 128;      // 864 float4, 7 per thread
        float4 v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = min(tid + 128 * j, NV - 1);
            const int row = i / (XS / 4), c4 = (i - row * (XS / 4)) * 4;
            const int tt = blockIdx.x * 32 + row;
            const float4 q = *reinterpret_cast<const float4*>(a.x + (size_t)min(tt, a.T - 1) * FE + min(c4, FE - 4));
            const float m = (tt < a.T && c4 < FE) ? 1.f : 0.f;
            v[j] = make_float4(q.x * m, q.y * m, q.z * m, q.w * m);
        }
        float bv0 = 0.f, bv1 = 0.f;
        if (!BWD) {      // this workgroup's 2 * units * 32 bias values (units <= 4)
            const int nb = 2 * a.units * 32, fb = blockIdx.y * nb;
            bv0 = a.b1[fb + min(tid, nb - 1)];
            bv1 = a.b1[fb + min(tid + 128, nb - 1)];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = tid + 128 * j;
            if (NV % 128 == 0 || i < NV) {
                const int row = i / (XS / 4), c4 = (i - row * (XS / 4)) * 4;
                *reinterpret_cast<float4*>(smem + row * XS + c4) = v[j];
            }
        }
        bias_s[tid] = bv0;
        bias_s[tid + 128] = bv1;
    }

    floatx16 acc2[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[e][i] = 0.f;

    DropCtx dc;
    if (!BWD) dc = make_drop(a.rng, a.rng_add, a.site, a.p, a.train);
    const bool store_h = !BWD && a.h != nullptr && tok;

    const float4* pk = a.packed + (size_t)ft0 * UNIT_VEC + lane;
    float4 w1f[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) w1f[g] = pk[g * 64];
    __syncthreads();                                      // x tile staged

    for (int j = 0; j < a.units; ++j) {
        const int f0 = (ft0 + j) * 32;
        // Everything this unit's epilogue and the first half of GEMM2 will need is requested BEFORE GEMM1 and pinned there
        // (left to itself hipcc sinks the loads to their first use, exposing a full memory round trip each):
        //   GEMM2's A fragments of hidden-unit groups gq = 0, 1 (gq = 2, 3 follow under GEMM2's first half — all 16
        //   vectors at once would spill), and the backward epilogue's operand, the saved h tile (linear1's bias sits in LDS)
        float4 wA[ET], wB[ET], wC[ET], aux[4];
#pragma unroll
        for (int e = 0; e < ET; ++e) {
            wA[e] = pk[G1_VEC + (e * 4 + 0) * 64];
            wB[e] = pk[G1_VEC + (e * 4 + 1) * 64];
        }
        if (BWD) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                aux[gq] = *reinterpret_cast<const float4*>(a.h + (size_t)min(t, a.T - 1) * a.F + f0 + 8 * gq + 4 * h);
        }
        const float4* pk2 = pk + G1_VEC;
        __builtin_amdgcn_sched_barrier(0);

        // ---- GEMM1: hidden tile (32 hidden units x 32 tokens), K = 100 (13 groups of 8, last half zero)
        floatx16 acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
        {
            // x fragments in chunks of 4 groups, one chunk ahead of the MFMAs (all 13 at once would cost 52 registers)
            const float* xrow = smem + r * XS + 4 * h;
            float4 xa[4], xn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const float4*>(xrow + 8 * i);
#pragma unroll
            for (int g0 = 0; g0 < KG; g0 += 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (g0 + 4 + i < KG) xn[i] = *reinterpret_cast<const float4*>(xrow + 8 * (g0 + 4 + i));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (g0 + i < KG) {
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[g0 + i].x, xa[i].x, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[g0 + i].y, xa[i].y, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[g0 + i].z, xa[i].z, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1f[g0 + i].w, xa[i].w, acc1, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) xa[i] = xn[i];
            }
        }
        // next unit's GEMM1 fragments: in flight under the epilogue and GEMM2 (the last unit re-reads its own: harmless)
        pk += (j + 1 < a.units) ? UNIT_VEC : 0;
#pragma unroll
        for (int g = 0; g < KG; ++g) w1f[g] = pk[g * 64];
        __builtin_amdgcn_sched_barrier(0);

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
                        mine |= (wd[q] >= dc.thr ? 1u : 0u) << (gq * 4 + q);
                }
                mq[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0x00, 0xF, 0xF, true);
                mq[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0x55, 0xF, 0xF, true);
                mq[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xAA, 0xF, 0xF, true);
                mq[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xFF, 0xF, 0xF, true);
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bq = *reinterpret_cast<const float4*>(bias_s + (w * a.units + j) * 32 + 8 * gq + 4 * h);
                const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * gq + q;
                    const float v = fmaxf(acc1[i] + bv[q], 0.f);
                    const bool keep = (mq[q] >> (gq * 4 + (lane & 3))) & 1u;
                    acc1[i] = keep ? v * dc.scale : 0.f;
                }
            }
            if (store_h) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(a.h + (size_t)t * a.F + f0 + 8 * gq + 4 * h) =
                        make_float4(acc1[4 * gq], acc1[4 * gq + 1], acc1[4 * gq + 2], acc1[4 * gq + 3]);
            }
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                acc1[4 * gq + 0] = aux[gq].x > 0.f ? acc1[4 * gq + 0] * a.mscale : 0.f;
                acc1[4 * gq + 1] = aux[gq].y > 0.f ? acc1[4 * gq + 1] * a.mscale : 0.f;
                acc1[4 * gq + 2] = aux[gq].z > 0.f ? acc1[4 * gq + 2] * a.mscale : 0.f;
                acc1[4 * gq + 3] = aux[gq].w > 0.f ? acc1[4 * gq + 3] * a.mscale : 0.f;
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                if (tok)
                    *reinterpret_cast<float4*>(a.dh + (size_t)t * a.F + f0 + 8 * gq + 4 * h) =
                        make_float4(acc1[4 * gq], acc1[4 * gq + 1], acc1[4 * gq + 2], acc1[4 * gq + 3]);
            }
        }

        // ---- GEMM2: yT[e][t] += sum over this tile's 32 hidden units; B operand = acc1 registers.  Consecutive MFMAs
        // share the B register (only A changes): the issue pattern the fp32 MFMA sustains at full rate.
#define GF_FFN_G2(W, GQ)                                                                                                   \
        _Pragma("unroll") for (int e = 0; e < ET; ++e) acc2[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[e].x, acc1[4 * (GQ) + 0], acc2[e], 0, 0, 0); \
        _Pragma("unroll") for (int e = 0; e < ET; ++e) acc2[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[e].y, acc1[4 * (GQ) + 1], acc2[e], 0, 0, 0); \
        _Pragma("unroll") for (int e = 0; e < ET; ++e) acc2[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[e].z, acc1[4 * (GQ) + 2], acc2[e], 0, 0, 0); \
        _Pragma("unroll") for (int e = 0; e < ET; ++e) acc2[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[e].w, acc1[4 * (GQ) + 3], acc2[e], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < ET; ++e) wC[e] = pk2[(e * 4 + 2) * 64];
        __builtin_amdgcn_sched_barrier(0);
        GF_FFN_G2(wA, 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < ET; ++e) wA[e] = pk2[(e * 4 + 3) * 64];
        __builtin_amdgcn_sched_barrier(0);
        GF_FFN_G2(wB, 1)
        GF_FFN_G2(wC, 2)
        GF_FFN_G2(wA, 3)
#undef GF_FFN_G2
    }

    // ---- sum the two waves' partial outputs through LDS (fixed order: wave 0 + wave 1), then the slab store
    __syncthreads();                                      // both waves are done reading the x tile
    if (w == 1) {
#pragma unroll
        for (int e = 0; e < ET; ++e)
#pragma unroll
            for (int i = 0; i < 16; ++i) smem[(e * 16 + i) * 64 + lane] = acc2[e][i];
    }
    __syncthreads();
    if (w == 0) {
        // linear2's bias rides on slab 0: loaded for every tile up front (a load inside the store loop would be waited for
        // together with the stores before it, once per tile)
        const float bmul = (!BWD && blockIdx.y == 0) ? 1.f : 0.f;
        float4 bb[ET][4];
#pragma unroll
        for (int e = 0; e < ET; ++e)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int e0 = min(32 * e + 8 * gq + 4 * h, FE - 4);
                bb[e][gq] = BWD ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(a.b2 + e0);
            }
        float* out = a.slabs + (size_t)blockIdx.y * a.slab_stride + (size_t)min(t, a.T - 1) * FE;
#pragma unroll
        for (int e = 0; e < ET; ++e) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int e0 = 32 * e + 8 * gq + 4 * h;
                float4 v = make_float4(acc2[e][4 * gq] + smem[(e * 16 + 4 * gq) * 64 + lane],
                                       acc2[e][4 * gq + 1] + smem[(e * 16 + 4 * gq + 1) * 64 + lane],
                                       acc2[e][4 * gq + 2] + smem[(e * 16 + 4 * gq + 2) * 64 + lane],
                                       acc2[e][4 * gq + 3] + smem[(e * 16 + 4 * gq + 3) * 64 + lane]);
                v.x += bmul * bb[e][gq].x; v.y += bmul * bb[e][gq].y; v.z += bmul * bb[e][gq].z; v.w += bmul * bb[e][gq].w;
                if (tok && e0 < FE) *reinterpret_cast<float4*>(out + e0) = v;
            }
        }
    }
}

bool ffn_fused_supported(int E, int F) { return E == FE && F >= 256 && (F % 256) == 0; }

// f-tiles per wave: 2 below ~4096 tokens (T = 3008: 3008 waves, 2.94 per SIMD), 4 above (T = 6016: again 3008 waves)
static int ffn_units(int T) { return T <= 4096 ? 2 : 4; }
int ffn_fused_splits(int T, int F) { return F / 32 / (2 * ffn_units(T)); }

long ffn_pack_floats(int F) { return (long)(F / 32) * UNIT_VEC * 4; }

// pack linear1 / linear2 of L consecutive layers (fwd or bwd orientation) into packed[L][ffn_pack_floats(F)]
int launch_ffn_pack(const float* params, long layer_stride, long off_w1, long off_w2, float* packed, int L, int F, int bwd,
                    hipStream_t st) {
    GF_CHECK_ARG(params, "ffn_pack: bad arguments");
    return launch_ffn_pack_ptrs(params + off_w1, params + off_w2, layer_stride, packed, L, F, bwd, st);
}

// the same with the two weight matrices given by their own base pointers (separate allocations: one layer, stride 0)
int launch_ffn_pack_ptrs(const float* w1, const float* w2, long layer_stride, float* packed, int L, int F, int bwd,
                         hipStream_t st) {
    GF_CHECK_ARG(w1 && w2 && packed && aligned16(packed) && L >= 1, "ffn_pack: bad arguments");
    const int nvec = (F / 32) * UNIT_VEC;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((nvec + 255) / 256, L), dim3(256), 0, st, w1, w2, layer_stride,
                       reinterpret_cast<float4*>(packed), F, bwd, nvec);
    GF_LAUNCH_CHECK();
    return 0;
}

static int launch_ffn(const FfnArgs& a, int splits, int bwd, hipStream_t st) {
    dim3 grid((a.T + 31) / 32, splits);
    if (bwd) hipLaunchKernelGGL(ffn_fused_kernel<1>, grid, dim3(128), 0, st, a);
    else hipLaunchKernelGGL(ffn_fused_kernel<0>, grid, dim3(128), 0, st, a);
    GF_LAUNCH_CHECK();
    return 0;
}

int launch_ffn_fused_fwd(const float* x, const float* packed, const float* b1, const float* b2, float* h, float* slabs,
                         long slab_stride, int T, int E, int F, float p, uint32_t site, const uint64_t* rng, uint64_t add,
                         int train, int* splits_out, hipStream_t st) {
    GF_CHECK_ARG(ffn_fused_supported(E, F), "ffn_fused: unsupported E=%d F=%d", E, F);
    GF_CHECK_ARG(x && packed && b1 && b2 && slabs && T > 0, "ffn_fused_fwd: null pointer");
    GF_CHECK_ARG(aligned16(x) && aligned16(packed) && aligned16(b1) && aligned16(b2) && aligned16(slabs) && (!h || aligned16(h)),
                 "ffn_fused_fwd: 16-byte alignment required");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "ffn_fused_fwd: rng required when dropout is active");
    const int splits = ffn_fused_splits(T, F);
    FfnArgs a{x, reinterpret_cast<const float4*>(packed), b1, b2, h, nullptr, slabs, slab_stride, T, F, ffn_units(T), 1.f, p, site,
              rng, add, train};
    *splits_out = splits;
    return launch_ffn(a, splits, 0, st);
}

int launch_ffn_fused_bwd(const float* dy, const float* packed, const float* h, float* dh, float* slabs, long slab_stride, int T,
                         int E, int F, float mscale, int* splits_out, hipStream_t st) {
    GF_CHECK_ARG(ffn_fused_supported(E, F), "ffn_fused: unsupported E=%d F=%d", E, F);
    GF_CHECK_ARG(dy && packed && h && dh && slabs && T > 0, "ffn_fused_bwd: null pointer");
    GF_CHECK_ARG(aligned16(dy) && aligned16(packed) && aligned16(h) && aligned16(dh) && aligned16(slabs),
                 "ffn_fused_bwd: 16-byte alignment required");
    const int splits = ffn_fused_splits(T, F);
    FfnArgs a{dy, reinterpret_cast<const float4*>(packed), nullptr, nullptr, const_cast<float*>(h), dh, slabs, slab_stride, T, F,
              ffn_units(T), mscale, 0.f, 0, nullptr, 0, 0};
    *splits_out = splits;
    return launch_ffn(a, splits, 1, st);
}

}  // namespace ganffn
