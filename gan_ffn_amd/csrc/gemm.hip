// gemm.hip — fp32 MFMA GEMMs for gfx950 (v_mfma_f32_32x32x2_f32), LDS-tiled, with fused epilogues.
//
// Three operand forms cover forward, dgrad and wgrad of every nn.Linear on the hot path
// (torch nn.Linear / MultiheadAttention in/out-proj / TransformerEncoderLayer linear1/2, call sites
// /root/reference/model.py:1210-1216,1244-1249,1307-1313):
//   NT  C[MxN]  = A[MxK] * W[NxK]^T          forward  (A, W both K-contiguous)
//   NN  C[MxN]  = A[MxK] * B[KxN]            dgrad    (B N-contiguous)
//   TN  C[MxN] += At[KxM]^T * B[KxN]         wgrad    (both K-major); deterministic: one owner workgroup per output
//                                                      tile adds in place, or (split-K) partial slabs + an ordered reduce
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain in k order (no reduced precision).
//
// Tiling: 256 threads = 4 waves (2 x 2); block tile 64 x 64, BK = 16; each wave owns one 32x32 tile (16 accumulator
// VGPRs).  LDS images:
//   K-contiguous operand: [rows][BK] floats, unpadded, 16-byte slots XOR-swizzled by the row (kc_off below): the
//     loaders' ds_write_b128 and the ds_read_b128 fragment reads (lane (r,h) -> row r, k = kk + 4h .. 4h+3) are both
//     bank-conflict-free;
//   K-major operand:      [16][cols + 4] — lanes read consecutive columns with ds_read_b32.
// The k index inside an 8-wide group is permuted identically for A and B (MFMA j of the group takes
// k = kk + 4h + j on lane half h), which is all an MFMA needs.
// Global->LDS staging goes through a register prefetch queue (2-4 tiles deep) into two LDS stages: the loads of tile
// t+PD are issued before the MFMAs of tile t, tile t+1 is written to the other stage after them; one barrier per K tile.
#include "common.h"

#include <type_traits>

namespace ganffn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// BK in {16, 32, 64}; K-contiguous LDS row stride BK + 4 = 4 * odd  (20, 36, 68)

enum { MODE_NT = 0, MODE_NN = 1, MODE_TN = 2 };

struct GemmArgs {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    float* colsum;
    int M, N, K;
    int kchunk;  // split-K chunk (multiple of 64); K when not split
    long slab_stride;  // NT/NN split-K: split z writes its partial product to C + z*slab_stride (plain stores)
    EpiArgs ea;
    // TN split-K (deterministic): split z stores its partial dW to part + z*part_stride ([M x N] dense, then [M] column
    // sums); a reduce kernel adds the slabs to C / colsum in split order.  part == nullptr: the launch is not split and
    // the single owner workgroup of a tile adds its result to C in place (no atomics either way).
    float* part = nullptr;
    long part_stride = 0;
};

// Operand tile loaders.  Every global load is UNCONDITIONAL and comes from a clamped (always valid) address:
//  * rows / columns beyond the matrix edge are clamped onto the last valid one — they only ever feed output rows /
//    columns that are never stored, so their values do not matter;
//  * k beyond kend must contribute zero: the loaded vector is multiplied by a 0/1 factor.  (A select would do, but
//    hipcc turns "select(ok, load, 0)" back into a load under an exec-mask branch, and a load under a branch splits
//    the K loop into small basic blocks with a wait at each one.)  The clamped address reads finite data of the same
//    operand row, so 0 * x = 0 unless the operand itself holds Inf/NaN, in which case the product does anyway.
// LDS image of a K-contiguous tile: [rows][BK] with NO padding; the 16-byte slot s of row r sits at slot
// s ^ ((r / rows-per-256-B-bank-row) % slots-per-row).  With it both the ds_write_b128 of the loaders (8 consecutive
// lanes = 2 rows x 4 slots at BK = 16, banks mod 32) and the ds_read_b128 fragment reads (16-lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, same slot, banks mod 64) are conflict-free; the padded [rows][BK + 4] image
// it replaces had 2-way conflicts on every store (SQ_LDS_BANK_CONFLICT = 34 % of the LDS cycles) and is 25 % larger.
template <int BK>
__device__ __forceinline__ int kc_off(int row, int slot) {
    constexpr int NS = BK / 4, RPB = 64 / BK;
    return row * BK + 4 * (slot ^ ((row / RPB) % NS));
}

template <int ROWS, int BK, int NTH>  // K-contiguous operand tile: ROWS x BK floats -> regs (ROWS*BK/4/NTH float4 per thread)
struct KcTile {
    static constexpr int KV = BK / 4;
    static constexpr int LDK = BK;
    static constexpr int TOTALV = ROWS * KV;
    static constexpr int NV = (TOTALV + NTH - 1) / NTH;
    float4 v[NV];
    float f[NV];   // 0/1 factor, applied when the tile is written to LDS (the load must not be waited for here)
    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = min(tid + j * NTH, TOTALV - 1);
            const int row = i / KV, kc = (i % KV) << 2;
            const int gk = k0 + kc;                       // K % 4 == 0: a float4 never straddles kend
            f[j] = gk < kend ? 1.f : 0.f;
            v[j] = *reinterpret_cast<const float4*>(P + (size_t)min(row0 + row, nrows - 1) * ld + max(min(gk, kend - 4), 0));
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * NTH;
            const int row = i / KV, kc = (i % KV) << 2;
            if (TOTALV % NTH == 0 || i < TOTALV)
                *reinterpret_cast<float4*>(S + kc_off<BK>(row, kc >> 2)) = make_float4(v[j].x * f[j], v[j].y * f[j], v[j].z * f[j], v[j].w * f[j]);
        }
    }
};

template <int COLS, int BK, int NTH>  // K-major operand tile: BK x COLS floats
struct KmTile {
    static constexpr int TOTALV = COLS * BK / 4;
    static constexpr int NV = (TOTALV + NTH - 1) / NTH;
    static constexpr int LD = COLS + 4;
    float4 v[NV];
    float f[NV];
    __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int col0, int ncols, int k0, int kend, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = min(tid + j * NTH, TOTALV - 1);
            const int kr = i / (COLS / 4), c4 = (i % (COLS / 4)) << 2;
            const int gk = k0 + kr;
            f[j] = gk < kend ? 1.f : 0.f;
            v[j] = *reinterpret_cast<const float4*>(P + (size_t)max(min(gk, kend - 1), 0) * ld + max(min(col0 + c4, ncols - 4), 0));
        }
    }
    // column sums of the tile, per thread: thread tid's vectors all cover columns 4 * (tid % (COLS/4)) .. +3
    // when NTH % (COLS/4) == 0 (true for every instantiation: 256 % 16)
    __device__ __forceinline__ void add_to(float4& s4, int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (TOTALV % NTH == 0 || tid + j * NTH < TOTALV) {
                s4.x += v[j].x * f[j]; s4.y += v[j].y * f[j]; s4.z += v[j].z * f[j]; s4.w += v[j].w * f[j];
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * NTH;
            const int kr = i / (COLS / 4), c4 = (i % (COLS / 4)) << 2;
            if (TOTALV % NTH == 0 || i < TOTALV)
                *reinterpret_cast<float4*>(S + kr * LD + c4) = make_float4(v[j].x * f[j], v[j].y * f[j], v[j].z * f[j], v[j].w * f[j]);
        }
    }
};

template <int MODE, int BM, int BN, int BK>
struct Smem {
    static constexpr int LDK = BK;
    static constexpr int A_FLOATS = (MODE == MODE_TN) ? BK * (BM + 4) : BM * LDK;
    static constexpr int B_FLOATS = (MODE == MODE_NT) ? BN * LDK : BK * (BN + 4);
    static constexpr int STAGE = A_FLOATS + B_FLOATS;
    static constexpr int TOTAL = 2 * STAGE;
};

// Output stage shared by the LDS-tiled and the register-direct kernels: accumulator register i of tile (a, b) holds
// row mbase + 32a + (i&3) + 8(i>>2) + 4h, column nbase + 32b + r.
// dropout keep bits of a wave tile: bit 4 gq + q of keep[a][b] <-> accumulator register 4 gq + q (row rb + q of column col).
// Data-independent, so gemm_body evaluates it while the first operand tiles are still in flight from memory.
template <int EPI>
constexpr bool epi_has_dropout() { return EPI == EPI_RELU_DROP || EPI == EPI_DROP_GELU || EPI == EPI_GELU_BWD_DROP; }

template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_keep_bits(const GemmArgs& g, const DropCtx& dc, uint32_t (&keep)[TM][TN], const int mbase,
                                               const int nbase, const int r, const int h) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            uint32_t bits = 0xFFFFu;
            if (epi_has_dropout<EPI>() && dc.on) {
                bits = 0;
                const int col = nbase + b * 32 + r;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int rb = mbase + a * 32 + 8 * gq + 4 * h;
                    uint32_t w[4];
                    philox4((uint32_t)(rb >> 2) * (uint32_t)g.N + (uint32_t)col, dc.site, dc.o0, dc.o1, dc.k0, dc.k1, w);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bits |= (w[q] >= dc.thr ? 1u : 0u) << (4 * gq + q);
                }
            }
            keep[a][b] = bits;
        }
}

template <int EPI>
constexpr bool epi_has_aux() { return EPI == EPI_NONE || EPI == EPI_MASK_POS || EPI == EPI_GELU_BWD_DROP || EPI == EPI_GELU_BWD; }

// Epilogue operand (residual / saved activation): ALL loads of the wave tile are issued together, from clamped
// (always valid) offsets — a load under the per-element bounds test makes hipcc wait for it before the next
// one is issued (16 serialised round trips per tile in the first version of this epilogue).
template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_load_aux(const GemmArgs& g, float (&aux)[TM][TN][16], const int mbase, const int nbase,
                                              const int bz, const int r, const int h) {
    if (epi_has_aux<EPI>()) {
        const bool want = (EPI == EPI_NONE) ? (g.ea.aux_in != nullptr && bz == 0) : true;   // wave-uniform
        if (want) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int col = min(nbase + b * 32 + r, g.N - 1);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = min(mbase + a * 32 + 8 * (i >> 2) + 4 * h + (i & 3), g.M - 1);
                        aux[a][b][i] = g.ea.aux_in[(size_t)row * g.ldc + col];
                    }
                }
        } else {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) aux[a][b][i] = 0.f;
        }
    }
}

template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_apply_store(const GemmArgs& g, floatx16 (&acc)[TM][TN], const float (&aux)[TM][TN][16],
                                                 const int mbase, const int nbase, const int bz, const int r, const int h,
                                                 const DropCtx& dc, const uint32_t (&keep)[TM][TN]) {
    constexpr bool HAS_AUX = epi_has_aux<EPI>();
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = nbase + b * 32 + r;
            const bool colok = col < g.N;
            float bias = 0.f;
            if (EPI == EPI_NONE || EPI == EPI_RELU_DROP || EPI == EPI_DROP_GELU)
                if (g.ea.bias != nullptr && bz == 0) bias = g.ea.bias[min(col, g.N - 1)];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int rb = mbase + a * 32 + 8 * gq + 4 * h;  // 4 consecutive rows rb..rb+3
                float mult[4] = {1.f, 1.f, 1.f, 1.f};
                if (epi_has_dropout<EPI>()) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) mult[q] = ((keep[a][b] >> (4 * gq + q)) & 1u) ? dc.scale : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = rb + q;
                    const size_t off = (size_t)row * g.ldc + col;
                    float v = acc[a][b][gq * 4 + q];
                    float u = 0.f;
                    const float ax = HAS_AUX ? aux[a][b][gq * 4 + q] : 0.f;
                    if (EPI == EPI_NONE) {
                        v += bias;
                        v += ax;                                   // fused residual / branch add (0 when absent)
                    } else if (EPI == EPI_RELU_DROP) {
                        v = fmaxf(v + bias, 0.f) * mult[q];
                    } else if (EPI == EPI_DROP_GELU) {
                        u = (v + bias) * mult[q];
                        v = gelu_f(u);
                    } else if (EPI == EPI_MASK_POS) {
                        v = (ax > 0.f) ? v * g.ea.mscale : 0.f;
                    } else if (EPI == EPI_GELU_BWD_DROP) {
                        v = v * mult[q] * gelu_grad_f(ax);
                    } else if (EPI == EPI_GELU_BWD) {
                        v = v * gelu_grad_f(ax);
                    }
                    if (row < g.M && colok) {
                        if (EPI == EPI_DROP_GELU) g.ea.aux_out[off] = u;
                        g.C[(size_t)bz * g.slab_stride + off] = v;
                    }
                }
            }
        }
}

template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_store_epilogue(const GemmArgs& g, floatx16 (&acc)[TM][TN], const int mbase, const int nbase,
                                                    const int bz, const int r, const int h, const DropCtx& dc,
                                                    const uint32_t (&keep)[TM][TN]) {
    float aux[TM][TN][16];
    gemm_load_aux<EPI, TM, TN>(g, aux, mbase, nbase, bz, r, h);
    gemm_apply_store<EPI, TM, TN>(g, acc, aux, mbase, nbase, bz, r, h, dc, keep);
}

template <int MODE, int BM, int BN, int BK, int EPI, int WGM, int WGN>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz) {
    constexpr int NTH = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;  // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;
    using SM = Smem<MODE, BM, BN, BK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nt = (kend - kbeg + BK - 1) / BK;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // Register prefetch queue: PD tiles in flight.  The loads of tile t+PD are issued at iteration t, the tile
    // written to LDS at iteration t is the one loaded PD-1 iterations ago, so a K loop pays the global-load
    // latency once instead of once per tile (measured: the 1-deep version was load-latency-bound at 1-2 waves/SIMD).
    constexpr int PD = (BM * BN / (WGM * WGN) >= 64 * 64) ? 2 : 4;   // bigger per-wave tiles: fewer tiles in flight (VGPRs)
    KcTile<BM, BK, NTH> ta_kc[PD];
    KmTile<BM, BK, NTH> ta_km[PD];
    KcTile<BN, BK, NTH> tb_kc[PD];
    KmTile<BN, BK, NTH> tb_km[PD];

    // TN bias gradient: column sums of At over k, taken from the registers on their way to LDS (workgroups bx == 0)
    const bool want_colsum = MODE == MODE_TN && g.colsum != nullptr && bx == 0;
    float4 colsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
    static_assert(MODE != MODE_TN || (NTH % (BM / 4) == 0), "colsum: a thread's vectors must share their columns");

    auto gload = [&](auto slot, int t) {
        constexpr int u = decltype(slot)::value;
        const int k0 = kbeg + t * BK;     // beyond kend -> the tile loads zeros (no memory access)
        if (MODE == MODE_TN) ta_km[u].load(g.A, g.lda, m0, g.M, k0, kend, tid);
        else ta_kc[u].load(g.A, g.lda, m0, g.M, k0, kend, tid);
        if (MODE == MODE_NT) tb_kc[u].load(g.B, g.ldb, n0, g.N, k0, kend, tid);
        else tb_km[u].load(g.B, g.ldb, n0, g.N, k0, kend, tid);
    };
    auto sstore = [&](auto slot, int buf) {
        constexpr int u = decltype(slot)::value;
        float* sa = smem + buf * SM::STAGE;
        float* sb = sa + SM::A_FLOATS;
        if (MODE == MODE_TN) {
            ta_km[u].store(sa, tid);
            if (want_colsum) ta_km[u].add_to(colsum4, tid);   // every tile passes here exactly once
        } else {
            ta_kc[u].store(sa, tid);
        }
        if (MODE == MODE_NT) tb_kc[u].store(sb, tid); else tb_km[u].store(sb, tid);
    };


    // One K tile: ALL operand fragments of the tile are read from LDS first (BK/8 groups x (TM + TN) reads), then the
    // BK/2 x TM x TN MFMAs run back to back.  No run-time condition in here: a K tail is zero-filled by the loaders.
    auto compute = [&](int buf) {
        const float* sa = smem + buf * SM::STAGE;
        const float* sb = sa + SM::A_FLOATS;
        constexpr int NG = BK / 8;
        float af[NG][TM][4], bf[NG][TN][4];
#pragma unroll
        for (int gk = 0; gk < NG; ++gk) {
            const int kk = 8 * gk;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                if (MODE == MODE_TN) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[gk][a][j] = sa[(kk + 4 * h + j) * (BM + 4) + wm * WM + a * 32 + r];
                } else {
                    const float4 q = *reinterpret_cast<const float4*>(sa + kc_off<BK>(wm * WM + a * 32 + r, 2 * gk + h));
                    af[gk][a][0] = q.x; af[gk][a][1] = q.y; af[gk][a][2] = q.z; af[gk][a][3] = q.w;
                }
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if (MODE == MODE_NT) {
                    const float4 q = *reinterpret_cast<const float4*>(sb + kc_off<BK>(wn * WN + b * 32 + r, 2 * gk + h));
                    bf[gk][b][0] = q.x; bf[gk][b][1] = q.y; bf[gk][b][2] = q.z; bf[gk][b][3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf[gk][b][j] = sb[(kk + 4 * h + j) * (BN + 4) + wn * WN + b * 32 + r];
                }
            }
        }
        // MFMA order: consecutive instructions share one operand register wherever the wave tile allows it (a
        // boustrophedon walk over the TM x TN accumulators) — measured on MFMA-only loops (tools/lab/mfma_peak.hip):
        // an fp32 32x32x2 MFMA whose A AND B both differ from its predecessor's issues ~20 % slower.
#pragma unroll
        for (int gk = 0; gk < NG; ++gk)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int bb = 0; bb < TN; ++bb) {
                        const int b = (a & 1) ? TN - 1 - bb : bb;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[gk][a][j], bf[gk][b][j], acc[a][b], 0, 0, 0);
                    }
    };

    // prologue: PD tiles in flight, tile 0 to LDS
    gload(std::integral_constant<int, 0>{}, 0);
    if constexpr (PD > 1) gload(std::integral_constant<int, 1>{}, 1);
    if constexpr (PD > 2) gload(std::integral_constant<int, 2>{}, 2);
    if constexpr (PD > 3) gload(std::integral_constant<int, 3>{}, 3);
    // the epilogue's dropout keep bits (Philox: ~70 VALU instructions per call, 4 calls per tile) depend on indices only:
    // evaluated here, under the latency of the first global loads, instead of on the critical path after the last MFMA
    DropCtx dc;
    uint32_t keep[TM][TN];
    if constexpr (MODE != MODE_TN) {
        if (epi_has_dropout<EPI>()) dc = make_drop(g.ea.rng, g.ea.rng_add, g.ea.site, g.ea.p, g.ea.train);
        else dc.on = 0;
        gemm_keep_bits<EPI, TM, TN>(g, dc, keep, m0 + wm * WM, n0 + wn * WN, r, h);
    }
    sstore(std::integral_constant<int, 0>{}, 0);
    __syncthreads();

    // Steady state, iteration t: issue the global loads of tile t+PD, run tile t from LDS stage t&1, write tile t+1
    // (loaded PD-1 iterations ago) to the other stage, barrier.  The step body is straight-line code; the only branch
    // is the wave-uniform "is there a tile t" around a whole step.  Tiles at or beyond nt load as zeros.
#define GF_STEP(U)                                                                      \
    if (t0 + U < nt) {                                                                  \
        gload(std::integral_constant<int, U>{}, t0 + U + PD);   /* slot U is free */    \
        compute(U & 1);                                                                 \
        sstore(std::integral_constant<int, (U + 1) % PD>{}, (U + 1) & 1);               \
        __syncthreads();                                                                \
    }
    for (int t0 = 0; t0 < nt; t0 += PD) {
        GF_STEP(0)
        if constexpr (PD > 1) { GF_STEP(1) }
        if constexpr (PD > 2) { GF_STEP(2) }
        if constexpr (PD > 3) { GF_STEP(3) }
    }
#undef GF_STEP

    // ---------------- epilogue ----------------
    if (MODE == MODE_TN) {
        // No atomics: an output tile has ONE owner per K split.  Unsplit (part == nullptr) the owner adds in place
        // (the gradient slab is accumulated into, +=); split, it stores a partial slab that tn_reduce_kernel sums
        // in split order.  Either way the result does not depend on the order workgroups run in.
        float* const slab = g.part ? g.part + (size_t)bz * g.part_stride : nullptr;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + wn * WN + b * 32 + r;
                float old[16];
                if (!slab) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = min(m0 + wm * WM + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, g.M - 1);
                        old[i] = g.C[(size_t)row * g.ldc + min(col, g.N - 1)];
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = m0 + wm * WM + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (row < g.M && col < g.N) {
                        if (slab) slab[(size_t)row * g.N + col] = acc[a][b][i];
                        else g.C[(size_t)row * g.ldc + col] = old[i] + acc[a][b][i];
                    }
                }
            }
        if (want_colsum) {   // wave-uniform; smem is free: the K loop ended with a barrier
            constexpr int NR = NTH / (BM / 4);              // threads per column group (fixed summation order)
            static_assert(NR * BM <= SM::TOTAL, "colsum scratch must fit the tile buffers");
            *reinterpret_cast<float4*>(smem + (tid / (BM / 4)) * BM + ((tid % (BM / 4)) << 2)) = colsum4;
            __syncthreads();
            if (tid < BM) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < NR; ++i) sum += smem[i * BM + tid];
                if (m0 + tid < g.M) {
                    if (slab) slab[(size_t)g.M * g.N + m0 + tid] = sum;
                    else g.colsum[m0 + tid] += sum;       // bx == 0 is the only workgroup touching these rows' sums
                }
            }
        }
        return;
    }

    if constexpr (MODE != MODE_TN) gemm_store_epilogue<EPI, TM, TN>(g, acc, m0 + wm * WM, n0 + wn * WN, bz, r, h, dc, keep);
}

// SHORTK only names the launch class in profiles (1: the K <= 128 GEMMs of the d_model-100 networks, whose launches would
// otherwise hide behind the same symbol and grid as the d_model-512 ones); the code is the same.
template <int MODE, int BM, int BN, int BK, int EPI, int WGM = 2, int WGN = 2, int SHORTK = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_kernel(GemmArgs g) {
    gemm_body<MODE, BM, BN, BK, EPI, WGM, WGN>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// C[i] += sum_z part[z][i] (i < nC), colsum[i] += sum_z part[z][nC + i] (i < nS), slabs added in split order z = 0, 1, ...
__global__ __launch_bounds__(256) void tn_reduce_kernel(float* __restrict__ C, float* __restrict__ colsum,
                                                        const float* __restrict__ part, long part_stride, int splits, long nC,
                                                        int nS, int ldc, int N) {
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 < nC) {                                     // nC = M * N, N % 4 == 0: a float4 stays inside one row
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // slabs in split order; 8 loads in flight at a time (a plain loop waits for every load before the next)
        for (int z0 = 0; z0 < splits; z0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (size_t)min(z0 + u, splits - 1) * part_stride + i4);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float m = (z0 + u < splits) ? 1.f : 0.f;
                s.x += m * v[u].x; s.y += m * v[u].y; s.z += m * v[u].z; s.w += m * v[u].w;
            }
        }
        const long row = i4 / N, col = i4 - row * N;
        float* dst = C + row * ldc + col;
        dst[0] += s.x; dst[1] += s.y; dst[2] += s.z; dst[3] += s.w;
    } else if (colsum != nullptr && i4 < nC + 4 * (long)((nS + 3) / 4)) {
        for (int e = 0; e < 4; ++e) {
            const long i = i4 - nC + e;
            if (i < nS) {
                float s = part[nC + i];
                for (int z = 1; z < splits; ++z) s += part[(size_t)z * part_stride + nC + i];
                colsum[i] += s;
            }
        }
    }
}

// Grouped weight-gradient GEMM: up to MAXP independent TN problems (the 4 weight gradients of every encoder layer of
// a backward pass) in ONE launch.  A 100x100 or 300x100 gradient alone is a 100-block, latency-bound launch; grouped,
// its workgroups fill the gaps between the 768-block ones and ~30 launch floors per pass disappear.
constexpr int MAXP = 40;
struct TnProblem {
    const float* A; const float* B; float* C; float* colsum;
    int lda, ldb, ldc, M, N, K, kchunk;
    int tiles_n, tiles_mn;   // tiles along N, tiles_m * tiles_n
    int block0;              // first workgroup of this problem
};
struct TnGroup {
    TnProblem p[MAXP];
    int n;
};

__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(TnGroup grp) {
    // XCD-aware workgroup order: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (private
    // 4 MiB L2 each), so tiles that share an operand panel would all miss in different L2s (measured, rocprofv3
    // FETCH_SIZE: 3x the compulsory bytes on the d=512 launches).  Bijective remap (also when gridDim % 8 != 0): the
    // workgroups that share an XCD get one contiguous range of the logical (problem, k-split, tile) list.
    int pi = 0;
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xl = blockIdx.x & 7;
    const int b = (xl < r8 ? xl * (q8 + 1) : r8 * (q8 + 1) + (xl - r8) * q8) + (blockIdx.x >> 3);
#pragma unroll 1
    for (int i = 1; i < grp.n; ++i)
        if (b >= grp.p[i].block0) pi = i;
    const TnProblem& q = grp.p[pi];
    const int local = b - q.block0;
    const int bz = local / q.tiles_mn, t2 = local - bz * q.tiles_mn;
    GemmArgs g;
    g.A = q.A; g.lda = q.lda; g.B = q.B; g.ldb = q.ldb; g.C = q.C; g.ldc = q.ldc; g.colsum = q.colsum;
    g.M = q.M; g.N = q.N; g.K = q.K; g.kchunk = q.kchunk; g.slab_stride = 0;
    // tiles in panels of 8 along N (n fastest inside a panel, then m, then the next panel): an XCD's contiguous
    // range of ~32 tiles is then a 4 x 8 patch — 12 operand panels instead of the 33 of a 1 x 32 strip
    constexpr int PW = 8;
    const int tiles_m = q.tiles_mn / q.tiles_n;
    const int panel = t2 / (PW * tiles_m), rem = t2 - panel * PW * tiles_m;
    const int pw = min(PW, q.tiles_n - panel * PW);
    const int mt = rem / pw, nt = panel * PW + rem - mt * pw;
    gemm_body<MODE_TN, 64, 64, 16, EPI_NONE, 2, 2>(g, nt, mt, bz);
}

template <int MODE, int BM, int BN, int BK, int EPI, int WGM = 2, int WGN = 2, int SHORTK = 0>
static int launch_cfg(const GemmArgs& g, int splits, hipStream_t st) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits);
    constexpr size_t lds = Smem<MODE, BM, BN, BK>::TOTAL * sizeof(float);
    GF_TRY((lds_optin<gemm_kernel<MODE, BM, BN, BK, EPI, WGM, WGN, SHORTK>>(lds, "gemm")));
    hipLaunchKernelGGL((gemm_kernel<MODE, BM, BN, BK, EPI, WGM, WGN, SHORTK>), grid, dim3(64 * WGM * WGN), lds, st, g);
    GF_LAUNCH_CHECK();
    return 0;
}

template <int MODE, int EPI>
static int launch_pick(const GemmArgs& g, int splits, hipStream_t st) {
    // measured on MI355X (tools/gemm_bench.py, M = 3008 / 6016, N = 100 .. 2048, K = 100 .. 2048): with the
    // straight-line K loop the 4-wave 64x64x16 block is the fastest or within 2 % of the fastest of every block /
    // wave-tile shape tried (64x64x32, 128x64, 64x128, 128x128 with 2, 4, 8 or 16 waves), so it is the only one used.
    if (g.K <= 128) return launch_cfg<MODE, 64, 64, 16, EPI, 2, 2, 1>(g, splits, st);
    return launch_cfg<MODE, 64, 64, 16, EPI>(g, splits, st);
}

static int check_common(const float* A, int lda, const float* B, int ldb, const float* C, int M, int N, int K) {
    GF_CHECK_ARG(A && B && C, "gemm: null pointer");
    GF_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    GF_CHECK_ARG((lda & 3) == 0 && (ldb & 3) == 0, "gemm: leading dims must be multiples of 4 (lda=%d ldb=%d)", lda, ldb);
    GF_CHECK_ARG(aligned16(A) && aligned16(B), "gemm: operands must be 16-byte aligned");
    return 0;
}

#define EPI_SWITCH(MODE, g, st)                                                            \
    switch (epi) {                                                                         \
        case EPI_NONE: return launch_pick<MODE, EPI_NONE>(g, splits, st);                  \
        case EPI_RELU_DROP: return launch_pick<MODE, EPI_RELU_DROP>(g, 1, st);             \
        case EPI_DROP_GELU: return launch_pick<MODE, EPI_DROP_GELU>(g, 1, st);             \
        case EPI_MASK_POS: return launch_pick<MODE, EPI_MASK_POS>(g, 1, st);               \
        case EPI_GELU_BWD_DROP:                                                            \
        case EPI_GELU_BWD_DROP0: return launch_pick<MODE, EPI_GELU_BWD_DROP>(g, 1, st);    \
        case EPI_GELU_BWD: return launch_pick<MODE, EPI_GELU_BWD>(g, 1, st);               \
        default: return fail(-1, "gemm: unknown epilogue %d", epi);                        \
    }

// number of K splits for a plain (EPI_NONE) GEMM whose output has too few tiles to fill 256 CUs
int gemm_splitk_factor(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (tiles >= 384 || K < 512) return 1;
    long s = (768 + tiles - 1) / tiles;
    const long maxs = K / 256;
    if (s > maxs) s = maxs;
    if (s > 8) s = 8;
    return s < 1 ? 1 : (int)s;
}

static void set_split(GemmArgs& g, int& splits, long slab_stride) {
    if (splits <= 1) { splits = 1; g.kchunk = g.K; g.slab_stride = 0; return; }
    int kchunk = ((g.K + splits - 1) / splits + 63) / 64 * 64;
    splits = (g.K + kchunk - 1) / kchunk;
    g.kchunk = kchunk;
    g.slab_stride = slab_stride;
}

// splits > 1 (EPI_NONE only): split z writes its partial to C + z*slab_stride; the consumer sums the slabs.
// *splits_io returns the number of slabs actually written.
int launch_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                   int epi, const EpiArgs& ea, hipStream_t st, int* splits_io, long slab_stride) {
    GF_TRY(check_common(A, lda, W, ldw, C, M, N, K));
    GF_CHECK_ARG((K & 3) == 0, "gemm_nt: K=%d must be a multiple of 4", K);
    GemmArgs g{A, lda, W, ldw, C, ldc, nullptr, M, N, K, K, 0, ea};
    int splits = (splits_io && epi == EPI_NONE) ? *splits_io : 1;
    set_split(g, splits, slab_stride);
    if (splits_io) *splits_io = splits;
    EPI_SWITCH(MODE_NT, g, st)
}

int launch_gemm_nn(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K,
                   int epi, const EpiArgs& ea, hipStream_t st, int* splits_io, long slab_stride) {
    GF_TRY(check_common(A, lda, Bm, ldb, C, M, N, K));
    GF_CHECK_ARG((K & 3) == 0 && (N & 3) == 0, "gemm_nn: K=%d and N=%d must be multiples of 4", K, N);
    GemmArgs g{A, lda, Bm, ldb, C, ldc, nullptr, M, N, K, K, 0, ea};
    int splits = (splits_io && epi == EPI_NONE) ? *splits_io : 1;
    set_split(g, splits, slab_stride);
    if (splits_io) *splits_io = splits;
    EPI_SWITCH(MODE_NN, g, st)
}

// floats of partial-slab workspace a split TN launch of this shape can use (0: it will not split)
long gemm_tn_part_floats(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    long splits = (768 + tiles - 1) / tiles;             // ~3 blocks per CU measured best (tools/gemm_bench.py)
    const long maxsplits = (K + 63) / 64;                // at least 64 k per block
    if (splits > maxsplits) splits = maxsplits;
    if (splits <= 1) return 0;
    return splits * ((long)M * N + M);
}

static int launch_tn_reduce(float* C, int ldc, float* colsum, const float* part, long part_stride, int splits, int M, int N,
                            hipStream_t st) {
    const long nC = (long)M * N;
    const long n4 = nC / 4 + (colsum ? (M + 3) / 4 : 0);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, C, colsum, part, part_stride,
                       splits, nC, M, ldc, N);
    GF_LAUNCH_CHECK();
    return 0;
}

// C[M x N] += At^T B (+ column sums of At).  With a partial-slab workspace (part_ws, part_floats) the K range is split
// over workgroups (slabs + ordered reduce: deterministic); without one a single workgroup per tile runs the whole K.
int launch_gemm_tn_acc(const float* At, int lda, const float* Bm, int ldb, float* C, int ldc, float* colsum,
                       int M, int N, int K, hipStream_t st, float* part_ws, long part_floats) {
    GF_TRY(check_common(At, lda, Bm, ldb, C, M, N, K));
    GF_CHECK_ARG((M & 3) == 0 && (N & 3) == 0, "gemm_tn: M=%d and N=%d must be multiples of 4", M, N);
    EpiArgs ea;
    GemmArgs g{At, lda, Bm, ldb, C, ldc, colsum, M, N, K, K, 0, ea};
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    const long per = (long)M * N + M;
    long splits = (768 + tiles - 1) / tiles;
    const long maxsplits = (K + 63) / 64;
    if (splits > maxsplits) splits = maxsplits;
    if (part_ws == nullptr) splits = 1;
    else if (splits * per > part_floats) splits = part_floats / per;
    if (splits < 1) splits = 1;
    int kchunk = (int)(((K + splits - 1) / splits + 63) / 64 * 64);   // multiple of every BK
    splits = (K + kchunk - 1) / kchunk;
    g.kchunk = kchunk;
    if (splits > 1) {
        GF_CHECK_ARG(aligned16(part_ws), "gemm_tn: partial-slab workspace must be 16-byte aligned");
        g.part = part_ws;
        g.part_stride = (per + 3) & ~3L;
        if (splits * g.part_stride > part_floats) { splits = 1; g.part = nullptr; g.kchunk = K; }
    }
    GF_TRY((launch_cfg<MODE_TN, 64, 64, 16, EPI_NONE>(g, (int)splits, st)));
    if (splits > 1) GF_TRY(launch_tn_reduce(C, ldc, colsum, g.part, g.part_stride, (int)splits, M, N, st));
    return 0;
}

// dW_i[M_i x N_i] += At_i^T B_i for n problems in one launch (see gemm_tn_grouped_kernel).  No split-K: every output tile
// has one owner workgroup that runs the whole token range and adds its result in place — deterministic, and the group
// is wide enough without it (one encoder backward pass: 1136 tiles at d_model 100, 6144 at 512).
int launch_gemm_tn_grouped(const TnDesc* d, int n, hipStream_t st) {
    GF_CHECK_ARG(d && n >= 1 && n <= MAXP, "gemm_tn_grouped: n=%d out of [1,%d]", n, MAXP);
    TnGroup grp;
    grp.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        GF_TRY(check_common(d[i].At, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].M, d[i].N, d[i].K));
        GF_CHECK_ARG((d[i].M & 3) == 0 && (d[i].N & 3) == 0, "gemm_tn_grouped: M, N must be multiples of 4");
        TnProblem& q = grp.p[i];
        q.A = d[i].At; q.B = d[i].B; q.C = d[i].C; q.colsum = d[i].colsum;
        q.lda = d[i].lda; q.ldb = d[i].ldb; q.ldc = d[i].ldc; q.M = d[i].M; q.N = d[i].N; q.K = d[i].K;
        q.kchunk = d[i].K;
        const int tm = (d[i].M + 63) / 64;
        q.tiles_n = (d[i].N + 63) / 64;
        q.tiles_mn = tm * q.tiles_n;
        q.block0 = total;
        total += q.tiles_mn;
    }
    constexpr size_t lds = Smem<MODE_TN, 64, 64, 16>::TOTAL * sizeof(float);
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(total), dim3(256), lds, st, grp);
    GF_LAUNCH_CHECK();
    return 0;
}

}  // namespace ganffn
