// ffn3.hip — the d_model-100 feed-forward block as ONE forward kernel (round 4; opt-in: bit 7 of ganffn_debug_set_ffn_mode):
//     h = dropout(relu(x W1^T + b1))  [T x 2048],   y = h W2^T + b2  [T x 100]      (/root/reference/model.py:1210 -> torch
//     TransformerEncoderLayer._ff_block: linear1, ReLU, dropout, linear2; the residual + dropout + LayerNorm2 around it stay
//     in the consumer, rowchain.hip)
// in place of gemm_wres_kernel (K = 100 -> 2048) + gemm_n100_kernel (2048 -> 100) with the 24.6 MB hidden tensor written by
// the first and read back by the second.
//
// What differs from ffn.hip's two fused attempts (DESIGN.md section 3 "Fused FFN"): those gave every 32-token workgroup (v1)
// / 32-token x 32-hidden work unit (v2) its own copy of the weights — 357 MB of L2 -> CU traffic per launch.  Here a
// workgroup is 64 tokens x ONE FIFTH of the hidden units (235 workgroups at T = 3008): its waves share each 16-unit weight
// tile through LDS (12.8 KB per tile, pairs of tiles double-buffered, 78 MB per launch), the tokens' x fragments stay in
// registers for the whole chunk, and the hidden tile never leaves the registers between the two products:
//   * product 1, per 16 hidden units: acc1[hid][tok] = W1_tile x^T on v_mfma_f32_16x16x4_f32 (exact fp32), hidden unit on
//     the m axis, token on the n axis, K = 100 = 24 MFMAs on float4 operand reads + 1 tail MFMA, two accumulation chains;
//   * epilogue in registers: + b1, ReLU, Philox dropout (the quad of 4 consecutive tokens shares a rowgroup: lane ql of a
//     quad evaluates the call of hidden unit f0 + ql and the keep bits are exchanged by DPP, as in rowchain.hip), one 16-byte
//     store of the lane's 4 hidden units (skipped when no backward follows);
//   * product 2: lane (token c, group g) holds h[f0 + 4g + r][c] in accumulator register r — exactly the B operand of an
//     MFMA whose k slot g stands for hidden unit f0 + 4g + r (the attention kernels' accumulator-as-operand trick: the
//     contraction index is permuted identically on the W2 side), so y^T[feat][tok] += W2_tile h needs no LDS round trip:
//     7 feature tiles x 4 MFMAs per hidden tile;
//   * 8 waves: two waves per SIMD (T <= 4096: both halves of the workgroup hold the same 64 tokens and split each staged
//     pair of tiles, joined through LDS at the end; above: 128 tokens, every wave takes both tiles);
//   * hidden chunk z writes its partial y to slab z (+ b2 in slab 0); the LayerNorm-side consumer sums the slabs in order,
//     exactly as it does for gemm_n100's K chunks: no atomics, bit-reproducible.
//
// Measured (MI355X, tools/lab/ffn3_time.py, back-to-back launches, train mode): 34.6 us against 43.8 us for the two kernels
// at T = 3008, 60.7 against 64.2 us at T = 6016; eval 30.9 / 53.0 us.  In the step it does NOT pay (tools/lab/ffn3_ab.sh):
// single stream 44.30 -> 43.81 ms — less than the 0.72 ms of launch floors it removes, which the three-stream schedule hides
// anyway — and the default three-stream step 33.42 -> 33.85 ms (T <= 4096 only) / 34.19 ms (every T).  The reason is in the instruction mix: per
// hidden tile a wave issues 53 MFMAs (1,696 cycles) and ~100 VALU instructions (Philox 50, epilogue, addresses), and on
// this chip fp32 MFMA and VALU time ADD (issuing the Philox rounds between product 1's MFMAs changed nothing: 36.3 us) —
// 2 waves x 13 pairs x ~2,300 cycles = 25 us + fill/drain = the 28 us it runs at, and the two-kernel path pays the same
// MFMAs and the same Philox calls.  Fusing saves the hidden tensor's read-back, which was never the bound.  Kept opt-in.
#include "common.h"

namespace ganffn {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FE = 100, FF = 2048, FTOK = 64, FHT = 16;
constexpr int LD1 = 116;                 // W1 tile [16 hid][100 k (+ zero pad to 112)]: rows sit 52 banks apart -> conflict-free b128
constexpr int LD2 = 20;                  // W2 tile [112 feat][16 hid]: rows 20 apart -> conflict-free b128
constexpr int W1T = FHT * LD1, W2T = 112 * LD2, FSTAGE = W1T + W2T;      // 1856 + 2240 = 4096 floats per stage
constexpr int FPAIR = 2 * FSTAGE;             // a staged pair of hidden tiles
constexpr int FMAXCH = 8;
constexpr int FMAXB1 = FF;               // b1 of the chunk's hidden units (at most all of them: one chunk)

struct Ffn3Args {
    const float* x; const float* w1; const float* b1; const float* w2; const float* b2;
    float* h;                  // [T x 2048] or null (no backward follows)
    float* slabs; long slab_stride;
    int T, nch;                // hidden chunks = output slabs
    float p; uint32_t site; const uint64_t* rng; uint64_t add; int train;
};

template <int R>
__device__ __forceinline__ uint32_t quad_bcast3(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, R * 0x55, 0xF, 0xF, true);
}

// 8 waves: waves 0..3 (group 0) and 4..7 (group 1) hold the SAME 64 tokens and take the even / odd hidden tile of each
// staged PAIR of tiles — two waves per SIMD, so that one wave's LDS waits, epilogue and barrier arrival sit under the
// other's MFMAs — and group 1's partial y is added to group 0's through LDS at the end (fixed order).
//
// SPLIT = false (T > 4096): the 8 waves hold 128 DIFFERENT tokens instead and every wave takes both tiles of a pair — the
// same work per SIMD, half the workgroups (one resident round at T = 6016: 47 x 5), no join at the end.
template <bool SPLIT, bool DROP>
__global__ __launch_bounds__(512) void ffn3_fwd_kernel(Ffn3Args a) {
    __shared__ __attribute__((aligned(16))) float smem[2 * FPAIR + 4 + FMAXB1];
    constexpr int DUMP = 2 * FPAIR, B1S = 2 * FPAIR + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = SPLIT ? wave >> 2 : 0, w4 = SPLIT ? wave & 3 : wave;
    const int c = lane & 15, g = lane >> 4, ql = c & 3;
    const int m0 = blockIdx.x * (SPLIT ? FTOK : 2 * FTOK), z = blockIdx.y;
    // pairs of hidden tiles of this chunk: 64 pairs cut into nch nearly equal runs
    constexpr int NPAIR = FF / FHT / 2;
    const int p_beg = (int)(((long)z * NPAIR) / a.nch), p_end = (int)(((long)(z + 1) * NPAIR) / a.nch);
    const int np = p_end - p_beg;

    // ---- loaders: per hidden tile 400 float4 of W1 (16 rows x 25) and 400 of W2 (100 rows x 4); threads 0..399 take one of
    // each for both tiles of the pair, the others load clamped duplicates into the dump slot (nothing under a condition)
    const bool ldok = tid < 400;
    const int i0 = ldok ? tid : 0;
    const int r1 = i0 / 25, c1 = (i0 - r1 * 25) * 4;                 // W1: row (hid), k
    const int r2 = i0 >> 2, c2 = (i0 & 3) * 4;                        // W2: row (feat), hid
    const int s1 = ldok ? r1 * LD1 + c1 : DUMP, s2o = ldok ? W1T + r2 * LD2 + c2 : DUMP;
    float4 q1a, q1b, q2a, q2b;
#define GF_F3_GLOAD(PP)                                                                                                \
    {                                                                                                                  \
        const int f0_ = (p_beg + min((PP), np - 1)) * 2 * FHT;   /* (a prefetch beyond the last pair re-reads it) */     \
        q1a = *reinterpret_cast<const float4*>(a.w1 + (size_t)(f0_ + r1) * FE + c1);                                   \
        q1b = *reinterpret_cast<const float4*>(a.w1 + (size_t)(f0_ + FHT + r1) * FE + c1);                             \
        q2a = *reinterpret_cast<const float4*>(a.w2 + (size_t)r2 * FF + f0_ + c2);                                     \
        q2b = *reinterpret_cast<const float4*>(a.w2 + (size_t)r2 * FF + f0_ + FHT + c2);                               \
    }
#define GF_F3_SSTORE(BUF)                                                                                              \
    {                                                                                                                  \
        float* const sd_ = ldok ? smem + (BUF) * FPAIR : smem;                                                         \
        float* const se_ = ldok ? smem + (BUF) * FPAIR + FSTAGE : smem;                                                \
        *reinterpret_cast<float4*>(sd_ + s1) = q1a;                                                                    \
        *reinterpret_cast<float4*>(se_ + s1) = q1b;                                                                    \
        *reinterpret_cast<float4*>(sd_ + s2o) = q2a;                                                                   \
        *reinterpret_cast<float4*>(se_ + s2o) = q2b;                                                                   \
    }

    // ---- this lane's token: x fragments for all of K stay in registers (product 1's B operand): k = 16 q + 4 g + j
    const int tok = m0 + w4 * 16 + c;
    const float* xrow = a.x + (size_t)min(tok, a.T - 1) * FE;
    float4 xf[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) xf[q] = *reinterpret_cast<const float4*>(xrow + 16 * q + 4 * g);
    const float xtail = xrow[96 + g];                                       // k = 96 + g: the 25th k-step
    GF_F3_GLOAD(0)

    // zero the parts of the four tile images the loaders never write: W1 tile columns 100..115, W2 tile rows 100..111
    constexpr int ZT = FHT * 16 + 12 * LD2;
    for (int i = tid; i < 4 * ZT; i += 512) {
        const int st = i / ZT, j = i - st * ZT;
        if (j < FHT * 16) smem[st * FSTAGE + (j >> 4) * LD1 + FE + (j & 15)] = 0.f;
        else smem[st * FSTAGE + W1T + 100 * LD2 + (j - FHT * 16)] = 0.f;
    }
    for (int i = tid; i < np * 2 * FHT; i += 512) smem[B1S + i] = a.b1[p_beg * 2 * FHT + i];
    const DropCtx dc = make_drop(a.rng, a.add, a.site, a.p, a.train);
    const uint32_t rowgroup = (uint32_t)(m0 + w4 * 16 + c) >> 2;            // (m0 % 64 == 0: a quad of lanes = one rowgroup)

    floatx4 acc2[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) acc2[m] = floatx4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                      // zero fill before the first pair lands beside it
    GF_F3_SSTORE(0)
    __syncthreads();

    for (int t = 0; t < np; ++t) {
        GF_F3_GLOAD(t + 1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int half = SPLIT ? grp : 0; half < (SPLIT ? grp + 1 : 2); ++half) {
        const float* s = smem + (t & 1) * FPAIR + half * FSTAGE;
        const int f0 = (p_beg + t) * 2 * FHT + half * FHT;
        float wa[6][4];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(s + c * LD1 + 16 * q + 4 * g);
            wa[q][0] = v.x; wa[q][1] = v.y; wa[q][2] = v.z; wa[q][3] = v.w;
        }
        const float wt = s[c * LD1 + 96 + g];
        const float4 bb = *reinterpret_cast<const float4*>(smem + B1S + (2 * t + half) * FHT + 4 * g);
        // ---- product 1: acc1[hid 4g' + r][tok c'] over k, two accumulation chains
        floatx4 e0 = floatx4{0.f, 0.f, 0.f, 0.f}, e1 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q][0], xf[q].x, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q][1], xf[q].y, e1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q][2], xf[q].z, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q][3], xf[q].w, e1, 0, 0, 0);
        }
        e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt, xtail, e0, 0, 0, 0);
        // product 2's operand reads land under the epilogue
        float4 wb[7];
        {
            const float* s2 = s + W1T + c * LD2 + 4 * g;
#pragma unroll
            for (int m = 0; m < 7; ++m) wb[m] = *reinterpret_cast<const float4*>(s2 + 16 * m * LD2);
        }
        // ---- epilogue: this lane holds hidden units f0 + 4g .. + 3 of token `tok`; its quad shares a rowgroup: lane ql of
        // the quad evaluated hidden unit f0 + 4g + ql for the quad's 4 tokens
        float mult[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (DROP) {
            // (issuing the Philox rounds between product 1's MFMAs was measured: no change — fp32 MFMA and VALU time add up)
            uint32_t wd[4];
            philox4(rowgroup * (uint32_t)FF + (uint32_t)(f0 + 4 * g + ql), dc.site, dc.o0, dc.o1, dc.k0, dc.k1, wd);
            uint32_t mine = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) mine |= (wd[j] >= dc.thr ? 1u : 0u) << j;
            const uint32_t b4[4] = {quad_bcast3<0>(mine), quad_bcast3<1>(mine), quad_bcast3<2>(mine), quad_bcast3<3>(mine)};
#pragma unroll
            for (int r = 0; r < 4; ++r) mult[r] = ((b4[r] >> ql) & 1u) ? dc.scale : 0.f;
        }
        float hv[4];
        hv[0] = fmaxf((e0[0] + e1[0]) + bb.x, 0.f) * mult[0];
        hv[1] = fmaxf((e0[1] + e1[1]) + bb.y, 0.f) * mult[1];
        hv[2] = fmaxf((e0[2] + e1[2]) + bb.z, 0.f) * mult[2];
        hv[3] = fmaxf((e0[3] + e1[3]) + bb.w, 0.f) * mult[3];
        if (a.h != nullptr && tok < a.T)
            *reinterpret_cast<float4*>(a.h + (size_t)tok * FF + f0 + 4 * g) = make_float4(hv[0], hv[1], hv[2], hv[3]);
        // ---- product 2: y^T[feat][tok] += W2[feat][f0 + 4g + r] h[f0 + 4g + r][tok]: MFMA r, k slot g; consecutive MFMAs
        // go to different accumulators
#pragma unroll
        for (int m = 0; m < 7; ++m) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[m].x, hv[0], acc2[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 7; ++m) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[m].y, hv[1], acc2[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 7; ++m) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[m].z, hv[2], acc2[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 7; ++m) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[m].w, hv[3], acc2[m], 0, 0, 0);
      }
        __builtin_amdgcn_sched_barrier(0);
        GF_F3_SSTORE((t + 1) & 1)       // (after the last pair: its own data again, into the stage nobody reads — keeps the
                                        //  stores unconditional, so that no wait for the h store is placed at the loop head)
        __syncthreads();
    }
#undef GF_F3_GLOAD
#undef GF_F3_SSTORE

    // ---- group 1's partial y joins group 0's through LDS (the loop's last barrier is behind every read of the weight
    // images: the stage the last pair did not use carries it)
    float* img = smem + ((np & 1) ? FPAIR : 0);
    if (SPLIT && grp == 1) {
#pragma unroll
        for (int m = 0; m < 7; ++m)
            *reinterpret_cast<float4*>(img + ((w4 * 7 + m) * 64 + lane) * 4) = make_float4(acc2[m][0], acc2[m][1], acc2[m][2], acc2[m][3]);
    }
    if (SPLIT) __syncthreads();
    // ---- partial y of this hidden chunk: lane holds features 16 m + 4 g .. + 3 of its token
    if (grp == 0 && tok < a.T) {
        float* yrow = a.slabs + (size_t)z * a.slab_stride + (size_t)tok * FE;
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            const int f = 16 * m + 4 * g;
            if (f < FE) {
                float4 o = make_float4(acc2[m][0], acc2[m][1], acc2[m][2], acc2[m][3]);
                if (SPLIT) {
                    const float4 o1 = *reinterpret_cast<const float4*>(img + ((w4 * 7 + m) * 64 + lane) * 4);
                    o.x += o1.x; o.y += o1.y; o.z += o1.z; o.w += o1.w;
                }
                if (z == 0) {
                    const float4 b = *reinterpret_cast<const float4*>(a.b2 + f);
                    o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
                }
                *reinterpret_cast<float4*>(yrow + f) = o;
            }
        }
    }
}

}  // namespace

bool ffn3_supported(int E, int F) { return E == FE && F == FF; }

// hidden chunks (= output slabs): about one workgroup per CU at the training sizes (47 / 94 token tiles x 5), more chunks
// for small batches
static bool ffn3_split(int T) { return T <= 4096; }
int ffn3_chunks(int T, int max_slabs) {
    const int tok = ffn3_split(T) ? FTOK : 2 * FTOK;
    const int tiles = (T + tok - 1) / tok;
    int n = tiles >= 40 ? 5 : (256 + tiles - 1) / tiles;
    if (n > FMAXCH) n = FMAXCH;
    if (n > max_slabs) n = max_slabs;
    return n < 1 ? 1 : n;
}

int launch_ffn3_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* h, float* slabs,
                    long slab_stride, int T, float p, uint32_t site, const uint64_t* rng, uint64_t add, int train, int max_slabs,
                    int* splits_out, hipStream_t st) {
    GF_CHECK_ARG(x && w1 && b1 && w2 && b2 && slabs && T > 0 && max_slabs >= 1, "ffn3_fwd: bad arguments");
    GF_CHECK_ARG(aligned16(x) && aligned16(w1) && aligned16(b1) && aligned16(w2) && aligned16(b2) && aligned16(slabs) &&
                     (!h || aligned16(h)) && (slab_stride & 3) == 0, "ffn3_fwd: operands must be 16-byte aligned");
    GF_CHECK_ARG(!(train && p > 0.f) || rng, "ffn3_fwd: rng required when dropout is active");
    Ffn3Args a{x, w1, b1, w2, b2, h, slabs, slab_stride, T, ffn3_chunks(T, max_slabs), p, site, rng, add, train};
    const bool drop = train && p > 0.f;
    const dim3 grid_s((T + FTOK - 1) / FTOK, a.nch), grid_w((T + 2 * FTOK - 1) / (2 * FTOK), a.nch);
    if (ffn3_split(T)) {
        if (drop) hipLaunchKernelGGL((ffn3_fwd_kernel<true, true>), grid_s, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((ffn3_fwd_kernel<true, false>), grid_s, dim3(512), 0, st, a);
    } else {
        if (drop) hipLaunchKernelGGL((ffn3_fwd_kernel<false, true>), grid_w, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((ffn3_fwd_kernel<false, false>), grid_w, dim3(512), 0, st, a);
    }
    GF_LAUNCH_CHECK();
    if (splits_out) *splits_out = a.nch;
    return 0;
}

}  // namespace ganffn
