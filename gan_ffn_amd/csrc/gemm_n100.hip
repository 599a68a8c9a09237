// gemm_n100.hip — [T x K] x [K x 100] products with a long K (linear2 forward and the linear1 dgrad of the d_model-100
// feed-forward block: K = 2048; /root/reference/model.py:1210 -> torch TransformerEncoderLayer._ff_block and its backward).
//
// Why its own kernel: on the generic 64 x 64 tiles (gemm.hip) a 100-wide output is two column tiles = 128 columns, 22 %
// of every MFMA is padding, and the shape ran at 36-45 % of the fp32 MFMA peak (profiles/r02_*: 22 us at T = 3008, 35 us at
// T = 6016 for 7.9 / 15.7 us of arithmetic).  Here:
//  * v_mfma_f32_16x16x4_f32 (exact fp32) with the output FEATURE on the m axis: 100 features = 6 tiles of 16 + 4 features on
//    v_mfma_f32_4x4x1_16B_f32 (round 4, TAIL4 below: no padding; until then a seventh tile, 112 = 10.7 % padding);
//  * a wave owns ALL 7 feature tiles of 16 tokens: the 7 MFMAs of a k-step share the token operand, 8 operand registers
//    feed 7 MFMAs (the 32 x 32 single-accumulator wave tile of the generic kernel needs 2 per MFMA);
//  * a workgroup = 4 waves = 64 tokens; K is cut into 4..16 chunks (one output slab each, summed in slab order by the
//    LayerNorm-side consumer, rowchain.hip — no atomics); the chunk count is chosen so that the launch's waves fill the
//    1024 SIMDs in whole rounds (n100_splits);
//  * operands go through LDS in 32-wide K tiles, XOR-swizzled 16-byte slots (conflict-free ds_read_b128 fragment reads),
//    double-buffered with one barrier per tile; a tile is 56 MFMAs per wave (~1800 cycles), which covers the global-load
//    latency of the next tile with a 1-deep register prefetch;
//  * the weight comes either as rows of K (NT: linear2's W2 [100 x K]) or K-major (NN: linear1's W1 [K x 100], read with
//    ds_read_b32) — no transposed copy is made.
// k order inside a 16-wide group: lane group g takes k = 16 q + 4 g + j at the j-th MFMA, identically for both operands and
// for every token (a dialogue's bits do not depend on its batch position).
// Round 4: KW = 2 — EIGHT waves per workgroup: two waves per 16-token group, each taking one 16-wide half of every 32-wide K
// tile (q = its half), so a SIMD holds two waves of ONE workgroup and one's LDS reads / waits run under the other's MFMAs
// without a second output slab (more K chunks buy the same co-residency with more slab traffic and more prologues).  The
// two halves are added through LDS in fixed order (half 0 + half 1) by the half-0 waves.
#include "common.h"

namespace ganffn {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NE = 100, NT7 = 7, NBK = 32, NBM = 64;
constexpr int LDWK = 116;          // K-major weight tile [32][116]: rows 4 apart sit 16 banks apart (116 = 4 mod 8)

// K-contiguous LDS tile [rows][32]: 16-byte slot s of row r at slot s ^ ((r / 2) % 8) (gemm.hip kc_off<32>)
__device__ __forceinline__ int sw32(int row, int slot) { return row * NBK + 4 * (slot ^ ((row >> 1) & 7)); }

struct N100Args {
    const float* A; int lda;       // activations [T x K]
    const float* W; int ldw;       // NT: [100 x K] rows of K;  NN: [K x 100] rows of 100
    const float* bias;             // [100] or null; added by chunk 0
    float* C; long slab_stride;    // C + z * slab_stride: [T x 100] partial product of K chunk z
    int T, K, kchunk;
    GF_LAB_ONLY(unsigned long long* stamps;)   // lab builds only (make LAB=1): per workgroup {realtime at entry, cycles at entry,
                                               // after the prologue, after the K loop, at exit}
};

// TAIL4: features 96..99 on v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4 x 4, k = 1, 8 cycles) instead of a seventh 16-wide tile
// that is three quarters padding (32 cycles): block b = 4 g + c / 4 holds (tokens 4 (c / 4) .. + 3) x (features 96 .. 99) for
// the k values of lane group g — exactly what lane (c, g) already supplies as the token operand — so a 16-wide k group costs
// 24 x 32 + 4 x 8 = 800 MFMA cycles instead of 896; the four lane groups' partial sums meet in the epilogue (two shuffles).
template <bool WKMAJOR, int KW, bool TAIL4>
__global__ __launch_bounds__(256 * KW) void gemm_n100_kernel(N100Args a) {
    static_assert(KW == 1 || KW == 2, "one or two waves per token group along K");
    constexpr int NTHR = 256 * KW;
    constexpr int WT = WKMAJOR ? NBK * LDWK : 112 * NBK;          // weight tile floats
    constexpr int STAGE = WT + NBM * NBK;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 4];          // + a dump slot for the loaders' surplus vectors
    constexpr int DUMP = 2 * STAGE;
    static_assert(KW == 1 || 2 * STAGE >= NBM * 116, "the K-half exchange image [64 tokens][116] re-uses the stage buffers");
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = (tid >> 6) & 3, khalf = tid >> 8;             // token group of 16; which 16-wide half of a K tile (KW == 2)
    const int m0 = blockIdx.x * NBM, z = blockIdx.y;
    const int kbeg = z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    const int nt = (kend - kbeg) / NBK;                           // K, kchunk multiples of 32
    GF_LAB_ONLY(unsigned long long* const stamp = (a.stamps && tid == 0) ? a.stamps + 5 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
                if (stamp) { stamp[0] = __builtin_amdgcn_s_memrealtime(); stamp[1] = __builtin_amdgcn_s_memtime(); })

    // loader geometry (loop-invariant element offsets, clamped rows).  Activations: 64 rows x 8 slots = 512 float4, 2 per
    // thread.  Weights: NT 112 rows x 8 slots = 896 float4 (rows >= 100 clamped: they feed discarded features), NN 32 k rows
    // x 25 float4 = 800: 4 per thread, the last one partly surplus — a surplus vector is loaded from a clamped address and
    // stored to the dump slot, so no load or LDS store sits under a condition.  Everything is a NAMED scalar / vector: hipcc
    // keeps small arrays that live across the K loop in scratch memory.
#define GF_N100_AIDX(J)                                                                                             \
    const int ia##J = min(tid + NTHR * J, NBM * 8 - 1), rowa##J = ia##J >> 3, sla##J = ia##J & 7;                  \
    const uint32_t offa##J = (uint32_t)min(m0 + rowa##J, a.T - 1) * (uint32_t)a.lda + (uint32_t)(sla##J << 2);      \
    const int ldsa##J = WT + sw32(rowa##J, sla##J);
    GF_N100_AIDX(0) GF_N100_AIDX(1)
#define GF_N100_WIDX(J)                                                                                             \
    const int iw##J = tid + NTHR * J;                                                                               \
    const bool okw##J = iw##J < (WKMAJOR ? NBK * 25 : 112 * 8);                                                     \
    const int icw##J = min(iw##J, (WKMAJOR ? NBK * 25 : 112 * 8) - 1);                                              \
    const int rw##J = WKMAJOR ? icw##J / 25 : icw##J >> 3;                                                          \
    const int cw##J = WKMAJOR ? (icw##J - rw##J * 25) << 2 : (icw##J & 7);                                          \
    const uint32_t offw##J = WKMAJOR ? (uint32_t)rw##J * (uint32_t)a.ldw + (uint32_t)cw##J                          \
                                     : (uint32_t)min(rw##J, NE - 1) * (uint32_t)a.ldw + (uint32_t)(cw##J << 2);     \
    const int ldw_in##J = WKMAJOR ? rw##J * LDWK + cw##J : sw32(rw##J, cw##J);                                      \
    const int ldsw0_##J = okw##J ? ldw_in##J : DUMP, ldsw1_##J = okw##J ? STAGE + ldw_in##J : DUMP;
    GF_N100_WIDX(0) GF_N100_WIDX(1) GF_N100_WIDX(2) GF_N100_WIDX(3)
    // two register sets (A: even tiles, B: odd tiles): the loads of tile t + 2 are issued at step t, so a tile has two
    // steps (~3600 MFMA cycles) to arrive — with one step of cover a lone workgroup on a CU was load-latency-bound
    float4 qaA0, qaA1, qwA0, qwA1, qwA2, qwA3, qaB0, qaB1, qwB0, qwB1, qwB2, qwB3;
    // (every offset below is clamped into its operand: unbounded descriptors; the launcher refuses operands of 4 GiB or more)
    const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(a.A, 0xFFFFFFFFu);
    const __amdgpu_buffer_rsrc_t rsW = buf_rsrc(a.W, 0xFFFFFFFFu);
#define GF_N100_GLOAD(R, TT)                                                                                        \
    {                                                                                                               \
        const int k0 = kbeg + min((TT), nt - 1) * NBK; /* (a prefetch beyond the last tile re-reads it: never consumed) */ \
        /* buffer loads (round 5): descriptor in scalar registers + loop-invariant 32-bit byte offset per lane + the tile's    */ \
        /* scalar byte offset — no vector ALU address arithmetic per tile.  (With 64-bit per-lane addresses hipcc computed     */ \
        /* each address into the destination registers of the load it feeds and waited for the PREVIOUS register set's loads   */ \
        /* at the top of every step: s_waitcnt vmcnt(0) — the two-step prefetch never had two steps.)                          */ \
        const uint32_t sofa = (uint32_t)k0 * 4u, sofw = WKMAJOR ? (uint32_t)k0 * (uint32_t)a.ldw * 4u : (uint32_t)k0 * 4u;  \
        qa##R##0 = buf_load_f4(rsA, 4u * offa0, sofa);                                                                 \
        qw##R##0 = buf_load_f4(rsW, 4u * offw0, sofw); qw##R##1 = buf_load_f4(rsW, 4u * offw1, sofw);                     \
        if constexpr (KW == 1) {        /* 256 threads: 2 activation and 4 weight vectors each; 512 threads: 1 and 2 */ \
            qa##R##1 = buf_load_f4(rsA, 4u * offa1, sofa);                                                             \
            qw##R##2 = buf_load_f4(rsW, 4u * offw2, sofw); qw##R##3 = buf_load_f4(rsW, 4u * offw3, sofw);                 \
        }                                                                                                           \
    }
#define GF_N100_SSTORE(R, BUF)                                                                                      \
    {                                                                                                               \
        float* const sdst = smem + (BUF) * STAGE;                                                                   \
        *reinterpret_cast<float4*>(sdst + ldsa0) = qa##R##0;                                                        \
        *reinterpret_cast<float4*>(smem + ((BUF) ? ldsw1_0 : ldsw0_0)) = qw##R##0;                                  \
        *reinterpret_cast<float4*>(smem + ((BUF) ? ldsw1_1 : ldsw0_1)) = qw##R##1;                                  \
        if constexpr (KW == 1) {                                                                                    \
            *reinterpret_cast<float4*>(sdst + ldsa1) = qa##R##1;                                                    \
            *reinterpret_cast<float4*>(smem + ((BUF) ? ldsw1_2 : ldsw0_2)) = qw##R##2;                              \
            *reinterpret_cast<float4*>(smem + ((BUF) ? ldsw1_3 : ldsw0_3)) = qw##R##3;                              \
        }                                                                                                           \
    }

    floatx4 acc[NT7];
#pragma unroll
    for (int m = 0; m < NT7; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};

    if (WKMAJOR) {
        // columns 100 .. 111 of the K-major weight tile are never written by the loader: zero them once (both stages)
        for (int i = tid; i < 2 * NBK * 12; i += NTHR) {
            const int st = i / (NBK * 12), r = (i / 12) % NBK, cc = i % 12;
            smem[st * STAGE + r * LDWK + NE + cc] = 0.f;
        }
    }
    // one K tile from LDS stage BUF: all fragment reads, then its 56 MFMAs (KW == 2: this wave's 16-wide half, 28 MFMAs)
    constexpr int NQ = 2 / KW;
    auto compute = [&](const int buf) __attribute__((always_inline)) {
        const float* s = smem + buf * STAGE;
        const float* sa = s + WT;
        float4 bx[NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            const int q = KW == 2 ? khalf : qi;
            bx[qi] = *reinterpret_cast<const float4*>(sa + sw32(wave * 16 + c, 4 * q + g));
        }
        float wv[NQ][NT7][4];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            const int q = KW == 2 ? khalf : qi;
#pragma unroll
            for (int m = 0; m < NT7; ++m) {
                const int wr = (TAIL4 && m == NT7 - 1) ? 96 + (c & 3) : 16 * m + c;       // weight row (feature) this lane supplies
                if (WKMAJOR) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wv[qi][m][j] = s[(16 * q + 4 * g + j) * LDWK + wr];
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(s + sw32(wr, 4 * q + g));
                    wv[qi][m][0] = v.x; wv[qi][m][1] = v.y; wv[qi][m][2] = v.z; wv[qi][m][3] = v.w;
                }
            }
        }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            const float bq[4] = {bx[qi].x, bx[qi].y, bx[qi].z, bx[qi].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < NT7; ++m) {
                    if (TAIL4 && m == NT7 - 1) acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[qi][m][j], bq[j], acc[m], 0, 0, 0);
                    else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[qi][m][j], bq[j], acc[m], 0, 0, 0);
                }
        }
    };
    GF_N100_GLOAD(A, 0)
    GF_N100_GLOAD(B, 1)
    GF_N100_SSTORE(A, 0)
    __syncthreads();
    GF_LAB_ONLY(if (stamp) stamp[2] = __builtin_amdgcn_s_memtime();)

    // steps in pairs: even tiles live in register set A / LDS stage 0, odd tiles in set B / stage 1.  The sched_barriers
    // keep the global loads ahead of the MFMAs of the step they are issued in (hipcc sinks them to their use otherwise).
    for (int t = 0; t < nt; t += 2) {
        GF_N100_GLOAD(A, t + 2)
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) GF_N100_SSTORE(B, 1)
        __syncthreads();
        if (t + 1 < nt) {                                  // wave-uniform
            GF_N100_GLOAD(B, t + 3)
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nt) GF_N100_SSTORE(A, 0)
            __syncthreads();
        }
    }
#undef GF_N100_GLOAD
#undef GF_N100_SSTORE
#undef GF_N100_AIDX
#undef GF_N100_WIDX

    GF_LAB_ONLY(if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();)
    // epilogue: lane (c, g) holds token m0 + 16 wave + c, features 16 m + 4 g .. + 3
    if constexpr (KW == 2) {
        // the K halves meet in LDS (the stage buffers are dead: the loop ended on a barrier): half 1 writes its tile image
        // [64 tokens][116], half 0 adds it to its own accumulators — always (half 0) + (half 1)
        float* xrow = smem + (wave * 16 + c) * 116 + 4 * g;
        if (khalf == 1) {
#pragma unroll
            for (int m = 0; m < NT7; ++m) *reinterpret_cast<float4*>(xrow + 16 * m) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        }
        __syncthreads();
        if (khalf == 1) return;
#pragma unroll
        for (int m = 0; m < NT7; ++m) {
            const float4 o = *reinterpret_cast<const float4*>(xrow + 16 * m);
            acc[m][0] += o.x; acc[m][1] += o.y; acc[m][2] += o.z; acc[m][3] += o.w;
        }
    }
    if constexpr (TAIL4) {
        // features 96..99: register r of every lane group holds the partial sum over ITS k values of (token c, feature 96 + r)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[NT7 - 1][r];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[NT7 - 1][r] = v;
        }
    }
    const int tok = m0 + wave * 16 + c;
    if (tok < a.T) {
        float* crow = a.C + (size_t)z * a.slab_stride + (size_t)tok * NE;
#pragma unroll
        for (int m = 0; m < NT7; ++m) {
            const int f = 16 * m + 4 * g;
            if (f < NE) {
                float4 o = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
                if (a.bias != nullptr && z == 0) {
                    const float4 b = *reinterpret_cast<const float4*>(a.bias + f);
                    o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
                }
                *reinterpret_cast<float4*>(crow + f) = o;
            }
        }
    }
    GF_LAB_ONLY(if (stamp) stamp[4] = __builtin_amdgcn_s_memtime();)
}

}  // namespace

// lab knobs of the mode word (common.h `Mode`): bits 8..15 force the K-chunk count, bit 23 = features 96..99 on a padded seventh
// 16-wide tile (round 3's form), bits 20..21 force four (1) or eight (2) waves per workgroup
GF_LAB_ONLY(unsigned long long* g_n100_stamps = nullptr;)   // lab builds only: device buffer for in-kernel time stamps

bool n100_supported(int N, int K) { return N == NE && K >= 256 && (K % NBK) == 0; }

// K chunks (= output slabs, <= max_splits) and waves per workgroup.  Measured on MI355X (tools/lab/n100_lab.py, K = 2048,
// gpurun_out/r4_n100_lab.txt; round 3's 4-wave numbers in brackets):
//  * 8 waves (two per 16-token group, one 16-wide half of every K tile each): rows-of-K weights T = 3008: 4 / 5 / 8 / 10 / 16
//    chunks 20.8 / 18.1 / 20.1 / 19.2 / 22.9 us [22.0 / 19.1 / 20.4 / 19.0 / 19.0]; T = 6016: 4 / 5 / 6 / 8 chunks
//    34.0 / 29.2 / 41.2 / 36.1 us [35.4 / 30.5 / 35.1 / 29.9];
//  * K-major weights (58 LDS reads per tile instead of 16): T = 3008: 5 / 8 / 16 chunks 19.3 / 20.3 / 18.1 us
//    [22.9 / 21.6 / 18.5]; T = 6016: 5 / 6 / 8 chunks 29.9 / 34.0 / 29.1 us [32.4 / 35.0 / 29.9] — with the second wave per
//    SIMD inside the workgroup the K-major layout no longer needs the extra chunks (and their slabs) for co-residency.
// Rule (8 waves): a CU runs its workgroups' MFMA work back to back, so the launch costs ceil(workgroups / 256 CUs) x the
// chunk's K tiles, plus the slab traffic: every slab is written here and read again by the LayerNorm-side consumer
// (~0.8 tile-times per slab at T = 3008; the consumer at T = 6016 with 8 slabs is bandwidth-bound on them: 19 MB).
int n100_splits(int T, int K, int max_splits, int w_kmajor) {
    if (const int fs = mode().n100_force_splits(); fs > 0) return fs < max_splits ? fs : max_splits;
    const int tiles_m = (T + NBM - 1) / NBM, ksteps = K / NBK;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= max_splits && s <= ksteps; ++s) {
        const int per = (ksteps + s - 1) / s;
        if ((s - 1) * per >= ksteps) continue;                       // an empty last chunk
        const long wgs = (long)tiles_m * s;
        const long rounds = (wgs + 255) / 256;
        const double cost = (double)rounds * per + 0.8 * ((double)T / 3008.0) * s;
        if (cost < best_cost) { best_cost = cost; best = s; }
    }
    (void)w_kmajor;
    return best;
}

// waves per 16-token group along K: two (8-wave workgroups) — faster at every chunk count the rule above picks; the 4-wave
// form stays behind the lab knob for A/B
static int n100_kw(int T, int splits) {
    (void)T; (void)splits;
    return 2;
}

// C slabs = A[T x K] . W^T (w_kmajor == 0: W [100 x K]) or A . W (w_kmajor == 1: W [K x 100]); *splits_io: in = cap, out = slabs written
int launch_gemm_n100(const float* A, int lda, const float* W, int ldw, int w_kmajor, const float* bias, float* C, long slab_stride,
                     int T, int K, int* splits_io, hipStream_t st) {
    GF_CHECK_ARG(A && W && C && splits_io && T > 0 && n100_supported(NE, K), "gemm_n100: bad arguments (K=%d)", K);
    GF_CHECK_ARG(aligned16(A) && aligned16(W) && aligned16(C) && (lda & 3) == 0 && (ldw & 3) == 0 && (slab_stride & 3) == 0 &&
                     (!bias || aligned16(bias)), "gemm_n100: operands must be 16-byte aligned");
    GF_CHECK_ARG((unsigned long long)T * (unsigned long long)lda * 4ull < (1ull << 32) &&
                     (unsigned long long)(w_kmajor ? K : NE) * (unsigned long long)ldw * 4ull < (1ull << 32),
                 "gemm_n100: an operand of 4 GiB or more is not supported");
    const int ksteps = K / NBK;
    int s = n100_splits(T, K, *splits_io < 1 ? 1 : *splits_io, w_kmajor);
    const int per = (ksteps + s - 1) / s;
    s = (ksteps + per - 1) / per;
    *splits_io = s;
    N100Args a{A, lda, W, ldw, bias, C, slab_stride, T, K, per * NBK GF_LAB_ONLY(, g_n100_stamps)};
    const dim3 grid((T + NBM - 1) / NBM, s);
    const Mode md = mode();
    const int kw = md.n100_force_kw() ? md.n100_force_kw() : n100_kw(T, s);
    if (kw == 2 && !md.n100_pad7()) {
        if (w_kmajor) hipLaunchKernelGGL((gemm_n100_kernel<true, 2, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((gemm_n100_kernel<false, 2, true>), grid, dim3(512), 0, st, a);
    } else if (kw == 2) {
        if (w_kmajor) hipLaunchKernelGGL((gemm_n100_kernel<true, 2, false>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((gemm_n100_kernel<false, 2, false>), grid, dim3(512), 0, st, a);
    } else {
        if (w_kmajor) hipLaunchKernelGGL((gemm_n100_kernel<true, 1, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_n100_kernel<false, 1, false>), grid, dim3(256), 0, st, a);
    }
    GF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganffn
