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
// Full tiles are loaded without any per-tile masking or address arithmetic (loop-invariant per-thread offsets from a
// uniform base pointer); the K tail is one zero-filled tile loaded up front — fp32 MFMAs share the SIMD's vector ALU, so
// every VALU instruction in the K loop costs MFMA time (tools/lab/mfma_valu.hip).
// Besides the generic kernel: gemm_wres_kernel (K = 100 -> N >= 1024: persistent workgroups, weight fragments resident in
// registers, same k order and bits as the generic kernel) and gemm_tn_grouped_kernel (all weight-gradient problems of a
// backward pass in one XCD-aware launch).
#include "common.h"

#include <type_traits>

namespace ganffn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Four single instructions hipcc will not leave alone: written in C++, `x & sext(bit)` and `min(x, 1) << i | y` come back
// as v_and + v_cmp_ne + v_cndmask chains (instcombine's canonical select form) — 3-4 instructions where one or two do.
__device__ __forceinline__ uint32_t v_bit_to_mask(uint32_t word, int bit) {            // 0 or 0xFFFFFFFF
    uint32_t r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(word), "n"(bit));
    return r;
}
__device__ __forceinline__ uint32_t v_and_keep(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_and_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t v_nonzero_bit_or(uint32_t x, int bit, uint32_t acc) {   // acc | ((x != 0) << bit)
    uint32_t t, r;
    asm("v_min_u32 %0, 1, %1" : "=v"(t) : "v"(x));
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(t), "n"(bit), "v"(acc));
    return r;
}

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

// Full-tile loaders for the steady state of the K loop: no bounds arithmetic and no 0/1 factor — fp32 MFMAs execute on
// the SIMD's vector ALU (tools/lab/mfma_valu.hip: VALU instructions issued between a wave's MFMAs ADD to its time, and
// other waves' VALU work overlaps them only ~25 %), so every VALU instruction in the K loop is paid for in MFMA time.
// The per-thread element offsets (clamped rows / columns) are computed once; per tile only the uniform base pointer moves
// (scalar ALU), and the K tail — the one tile that needs zero-filling — is loaded once, up front, by the masked loaders.
template <int ROWS, int BK, int NTH>
struct KcFull {
    static constexpr int KV = BK / 4, TOTALV = ROWS * KV, NV = (TOTALV + NTH - 1) / NTH;
    float4 v[NV];
    static __device__ __forceinline__ void offsets(uint32_t (&off)[NV], int ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = min(tid + j * NTH, TOTALV - 1);
            off[j] = (uint32_t)min(row0 + i / KV, nrows - 1) * (uint32_t)ld + (uint32_t)((i % KV) << 2);
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ Pk, const uint32_t (&off)[NV]) {   // Pk = P + k0 (uniform)
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(Pk + off[j]);
    }
    __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * NTH;
            if (TOTALV % NTH == 0 || i < TOTALV) *reinterpret_cast<float4*>(S + kc_off<BK>(i / KV, i % KV)) = v[j];
        }
    }
};

template <int COLS, int BK, int NTH>
struct KmFull {
    static constexpr int TOTALV = COLS * BK / 4, NV = (TOTALV + NTH - 1) / NTH, LD = COLS + 4;
    float4 v[NV];
    static __device__ __forceinline__ void offsets(uint32_t (&off)[NV], int ld, int col0, int ncols, int tid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = min(tid + j * NTH, TOTALV - 1);
            off[j] = (uint32_t)(i / (COLS / 4)) * (uint32_t)ld + (uint32_t)max(min(col0 + ((i % (COLS / 4)) << 2), ncols - 4), 0);
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ Pk, const uint32_t (&off)[NV]) {   // Pk = P + k0 * ld (uniform)
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(Pk + off[j]);
    }
    __device__ __forceinline__ void add_to(float4& s4, int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (TOTALV % NTH == 0 || tid + j * NTH < TOTALV) { s4.x += v[j].x; s4.y += v[j].y; s4.z += v[j].z; s4.w += v[j].w; }
    }
    __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * NTH;
            if (TOTALV % NTH == 0 || i < TOTALV) *reinterpret_cast<float4*>(S + (i / (COLS / 4)) * LD + ((i % (COLS / 4)) << 2)) = v[j];
        }
    }
};

// prefetch-queue slots as NAMED members (an array of slot structs indexed by the unrolled step ended up in scratch memory)
template <class T>
struct Slots4 {
    T s0, s1, s2, s3;
    template <int U>
    __device__ __forceinline__ T& get() {
        if constexpr (U == 0) return s0;
        else if constexpr (U == 1) return s1;
        else if constexpr (U == 2) return s2;
        else return s3;
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
        if constexpr (EPI == EPI_MASK_POS) {
            if (g.ea.mask_in != nullptr) {      // the pattern as one 16-bit word per lane and tile instead of 16 activations
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const int col = min(nbase + b * 32 + r, g.N - 1);
                        // row tile clamped like the aux_in rows below: a wave whose 32 rows lie wholly past M (M % 64 in 1..32)
                        // would read one tile past the ceil(M / 32)-tile pattern (its outputs are never stored)
                        const int rt = min((mbase + a * 32) >> 5, (g.M - 1) >> 5);
                        const uint32_t w = g.ea.mask_in[((size_t)rt * g.N + col) * 2 + h];
#pragma unroll
                        for (int i = 0; i < 16; ++i) aux[a][b][i] = 0.f;
                        aux[a][b][0] = __uint_as_float(w);
                    }
                return;
            }
        }
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
            const bool bitmask = EPI == EPI_MASK_POS && g.ea.mask_in != nullptr;          // uniform
            const uint32_t mword = EPI == EPI_MASK_POS ? __float_as_uint(aux[a][b][0]) : 0u;
            uint32_t pos = 0;                                                              // EPI_RELU_DROP: bit i <-> register i > 0
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
                        pos |= (v > 0.f ? 1u : 0u) << (gq * 4 + q);
                    } else if (EPI == EPI_DROP_GELU) {
                        u = (v + bias) * mult[q];
                        v = gelu_f(u);
                    } else if (EPI == EPI_MASK_POS) {
                        const bool on = bitmask ? ((mword >> (gq * 4 + q)) & 1u) != 0u : ax > 0.f;
                        v = on ? v * g.ea.mscale : 0.f;
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
            if (EPI == EPI_RELU_DROP && g.ea.mask_out != nullptr && colok && mbase + a * 32 < g.M)
                g.ea.mask_out[((size_t)((mbase + a * 32) >> 5) * g.N + col) * 2 + h] = (uint16_t)pos;
        }
}

// gemm_apply_store for ONE wave tile that lies fully inside the matrix (wave-uniform precondition: mbase + 32 <= M,
// nbase + 32 <= N), bias preloaded by the caller: straight-line code, no per-element bounds test.
template <int EPI>
__device__ __forceinline__ void gemm_apply_store_full(const GemmArgs& g, const floatx16& acc, const float (&aux)[16], const float bias,
                                                      const int mbase, const int nbase, const int r, const int h,
                                                      const DropCtx& dc, const uint32_t keep, const uint32_t slab_bytes = 0u) {
    constexpr bool HAS_AUX = epi_has_aux<EPI>();
    if constexpr (EPI == EPI_RELU_DROP || EPI == EPI_MASK_POS || EPI == EPI_NONE) {
        // Round 5, the two epilogues of the K = 100 kernel written for their instruction count (the tile loop of linear1 carried
        // 168 vector instructions per 50 MFMAs before Philox, the dgrad 157: profiles/r05_ffn_k100_pmc.json, 8.4 / 5.9 per MFMA):
        //  * stores through a buffer descriptor: lane offset loop-invariant, the row (mbase + dr) * ldc in the SCALAR offset — no
        //    64-bit address per store;
        //  * a keep / pattern bit becomes an all-ones / all-zeros word (one bit-field extract) ANDed onto the product: the dropped
        //    value is +0 exactly as `x * 0.f` of a non-negative x (linear1) resp. the `? :` (dgrad) gave — same bits;
        //  * the ReLU pattern bit of register i is min(bits(v), 1) << i: v >= +0 always, so v > 0 <=> its bits are not 0;
        //  * eval mode (dc.on == 0, wave-uniform) skips the dropout arithmetic instead of multiplying by 1.
        if (EPI == EPI_MASK_POS && g.ea.mask_in == nullptr) {
            // (saved-activation form of the mask, lab bit 25 only: the general code below)
        } else {
            const __amdgpu_buffer_rsrc_t rc = buf_rsrc(g.C, 0xFFFFFFFFu);
            const uint32_t vo = (uint32_t)(4 * h * g.ldc + nbase + r) * 4u;     // lane part; rows: scalar
            const uint32_t ldcb = (uint32_t)g.ldc * 4u;
            // (mbase is wave-uniform but derives from the wave id: say so, or every store becomes a readfirstlane loop)
            const uint32_t mb = (uint32_t)__builtin_amdgcn_readfirstlane(mbase);
            uint32_t pos = 0;
            if constexpr (EPI == EPI_NONE) {        // bias (+ residual) and the store; slab_bytes: the split-K slab of this workgroup
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dr = (i & 3) + 8 * (i >> 2);
                    buf_store_u32(rc, vo, slab_bytes + (mb + dr) * ldcb, __float_as_uint((acc[i] + bias) + aux[i]));
                }
            } else if constexpr (EPI == EPI_RELU_DROP) {
                const uint32_t sbits = __float_as_uint(dc.scale);
                if (dc.on) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int dr = (i & 3) + 8 * (i >> 2);
                        const uint32_t vb = v_and_keep(__float_as_uint(fmaxf(acc[i] + bias, 0.f) * __uint_as_float(sbits)), v_bit_to_mask(keep, i));
                        pos = v_nonzero_bit_or(vb, i, pos);
                        buf_store_u32(rc, vo, (mb + dr) * ldcb, vb);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int dr = (i & 3) + 8 * (i >> 2);
                        const uint32_t vb = __float_as_uint(fmaxf(acc[i] + bias, 0.f));
                        pos = v_nonzero_bit_or(vb, i, pos);
                        buf_store_u32(rc, vo, (mb + dr) * ldcb, vb);
                    }
                }
                if (g.ea.mask_out != nullptr)
                    g.ea.mask_out[((size_t)(mbase >> 5) * g.N + nbase + r) * 2 + h] = (uint16_t)pos;
            } else {
                const uint32_t mword = __float_as_uint(aux[0]);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dr = (i & 3) + 8 * (i >> 2);
                    buf_store_u32(rc, vo, (mb + dr) * ldcb, v_and_keep(__float_as_uint(acc[i] * g.ea.mscale), v_bit_to_mask(mword, i)));
                }
            }
            return;
        }
    }
    float* const cp = g.C + (size_t)(mbase + 4 * h) * g.ldc + nbase + r;
    float* const up = (EPI == EPI_DROP_GELU) ? g.ea.aux_out + (size_t)(mbase + 4 * h) * g.ldc + nbase + r : nullptr;
    const bool bitmask = EPI == EPI_MASK_POS && g.ea.mask_in != nullptr;                  // uniform
    const uint32_t mword = EPI == EPI_MASK_POS ? __float_as_uint(aux[0]) : 0u;
    uint32_t pos = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int dr = (i & 3) + 8 * (i >> 2);
        const float mult = epi_has_dropout<EPI>() ? (((keep >> i) & 1u) ? dc.scale : 0.f) : 1.f;
        float v = acc[i], u = 0.f;
        const float ax = HAS_AUX ? aux[i] : 0.f;
        if (EPI == EPI_NONE) {
            v += bias;
            v += ax;
        } else if (EPI == EPI_RELU_DROP) {
            v = fmaxf(v + bias, 0.f) * mult;
            pos |= (v > 0.f ? 1u : 0u) << i;
        } else if (EPI == EPI_DROP_GELU) {
            u = (v + bias) * mult;
            v = gelu_f(u);
        } else if (EPI == EPI_MASK_POS) {
            const bool on = bitmask ? ((mword >> i) & 1u) != 0u : ax > 0.f;
            v = on ? v * g.ea.mscale : 0.f;
        } else if (EPI == EPI_GELU_BWD_DROP) {
            v = v * mult * gelu_grad_f(ax);
        } else if (EPI == EPI_GELU_BWD) {
            v = v * gelu_grad_f(ax);
        }
        if (EPI == EPI_DROP_GELU) up[(size_t)dr * g.ldc] = u;
        cp[(size_t)dr * g.ldc] = v;
    }
    if (EPI == EPI_RELU_DROP && g.ea.mask_out != nullptr)
        g.ea.mask_out[((size_t)(mbase >> 5) * g.N + nbase + r) * 2 + h] = (uint16_t)pos;
}

template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_store_epilogue(const GemmArgs& g, floatx16 (&acc)[TM][TN], const int mbase, const int nbase,
                                                    const int bz, const int r, const int h, const DropCtx& dc,
                                                    const uint32_t (&keep)[TM][TN]) {
    float aux[TM][TN][16];
    gemm_load_aux<EPI, TM, TN>(g, aux, mbase, nbase, bz, r, h);
    if constexpr (TM == 1 && TN == 1 && (EPI == EPI_NONE || EPI == EPI_RELU_DROP || EPI == EPI_MASK_POS)) {
        // a wave tile that lies wholly inside the matrix (every tile of the hot shapes: M = 3008 = 47 x 64, N a multiple of 64):
        // the straight-line epilogue of the weight-resident kernel — buffer stores with scalar row offsets, no per-element
        // bounds test, the lean dropout / pattern arithmetic — instead of the general one (round 5).  Same values.
        const size_t span = ((size_t)(gridDim.z - 1) * (size_t)g.slab_stride + (size_t)g.M * (size_t)g.ldc) * sizeof(float);
        if (mbase + 32 <= g.M && nbase + 32 <= g.N && span < (size_t(1) << 31) && (EPI == EPI_NONE || bz == 0)) {   // wave-uniform
            float bias = 0.f;
            if (g.ea.bias != nullptr && bz == 0) bias = g.ea.bias[nbase + r];
            const uint32_t slab_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)((size_t)bz * (size_t)g.slab_stride * sizeof(float)));
            gemm_apply_store_full<EPI>(g, acc[0][0], aux[0][0], bias, mbase, nbase, r, h, dc, keep[0][0], slab_bytes);
            return;
        }
    }
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
    const int kbeg = min(bz * g.kchunk, g.K);   // (a split beyond a short problem's range computes and stores zeros)
    const int kend = min(g.K, kbeg + g.kchunk);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // Register prefetch queue: PD FULL tiles in flight.  The loads of tile t+PD are issued at iteration t, the tile
    // written to LDS at iteration t is the one loaded PD-1 iterations ago, so a K loop pays the global-load
    // latency once instead of once per tile (measured: the 1-deep version was load-latency-bound at 1-2 waves/SIMD).
    // Tiles 0 .. nfull-1 are full; the K tail (if any) is tile nfull: it is loaded once, before the loop, by the masked
    // loaders (zero-filled beyond kend) and goes to LDS when its turn comes.  Prefetches beyond the last full tile
    // re-read that tile (uniform clamp of the tile index): valid memory, never consumed.
    constexpr int PD = (BM * BN / (WGM * WGN) >= 64 * 64) ? 2 : 4;   // bigger per-wave tiles: fewer tiles in flight (VGPRs)
    const int nfull = (kend - kbeg) / BK;
    const int nt = nfull + (((kend - kbeg) % BK) ? 1 : 0);
    // One float4 of each operand per thread and tile (64 x 16 floats / 256 threads); the queue slots are plain named
    // vectors (arrays / structs of slots indexed by the unrolled step were left in scratch memory by hipcc).
    static_assert(BM * BK / 4 == NTH && (BN * BK / 4 == NTH || BN * BK / 4 == 2 * NTH),
                  "queue code: one A vector and one or two B vectors per thread and tile");
    constexpr bool B2 = BN * BK / 4 == 2 * NTH;          // BN = 128: a second B vector per thread (qc*)
    float4 qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3, qc0, qc1, qc2, qc3;
    // per-thread element offset inside a tile's rows (clamped at the matrix edge) and LDS slot, loop-invariant
    uint32_t offa, offb, offc = 0;
    int ldsa, ldsb, ldsc = 0;
    if constexpr (MODE == MODE_TN) {          // K-major: tile row = k, thread covers columns 4 * (tid % (BM/4)) ..
        const int kr = tid / (BM / 4), c4 = (tid % (BM / 4)) << 2;
        offa = (uint32_t)kr * (uint32_t)g.lda + (uint32_t)max(min(m0 + c4, g.M - 4), 0);
        ldsa = kr * (BM + 4) + c4;
    } else {                                  // K-contiguous: tile row = matrix row
        const int row = tid / (BK / 4), sl = tid % (BK / 4);
        offa = (uint32_t)min(m0 + row, g.M - 1) * (uint32_t)g.lda + (uint32_t)(sl << 2);
        ldsa = kc_off<BK>(row, sl);
    }
    if constexpr (MODE == MODE_NT) {
        const int row = tid / (BK / 4), sl = tid % (BK / 4);
        offb = (uint32_t)min(n0 + row, g.N - 1) * (uint32_t)g.ldb + (uint32_t)(sl << 2);
        ldsb = kc_off<BK>(row, sl);
        if constexpr (B2) {                   // second vector: tile rows NTH / (BK/4) further down
            constexpr int R2 = NTH / (BK / 4);
            offc = (uint32_t)min(n0 + row + R2, g.N - 1) * (uint32_t)g.ldb + (uint32_t)(sl << 2);
            ldsc = kc_off<BK>(row + R2, sl);
        }
    } else {
        const int kr = tid / (BN / 4), c4 = (tid % (BN / 4)) << 2;
        offb = (uint32_t)kr * (uint32_t)g.ldb + (uint32_t)max(min(n0 + c4, g.N - 4), 0);
        ldsb = kr * (BN + 4) + c4;
        if constexpr (B2) {                   // second vector: k rows NTH / (BN/4) further down, same columns
            constexpr int K2 = NTH / (BN / 4);
            offc = offb + (uint32_t)K2 * (uint32_t)g.ldb;
            ldsc = ldsb + K2 * (BN + 4);
        }
    }
    if (nfull == 0) offa = offb = offc = 0;   // K range shorter than one tile: the (never consumed) queue loads read element 0
    KcTile<BM, BK, NTH> tail_a_kc; KmTile<BM, BK, NTH> tail_a_km;      // the K-tail tile (all zeros when there is none)
    KcTile<BN, BK, NTH> tail_b_kc; KmTile<BN, BK, NTH> tail_b_km;

    // TN bias gradient: column sums of At over k, taken from the registers on their way to LDS (workgroups bx == 0)
    const bool want_colsum = MODE == MODE_TN && g.colsum != nullptr && bx == 0;
    float4 colsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
    static_assert(MODE != MODE_TN || (NTH % (BM / 4) == 0), "colsum: a thread's vectors must share their columns");

    const size_t kstride_a = (MODE == MODE_TN) ? (size_t)g.lda : 1, kstride_b = (MODE == MODE_NT) ? 1 : (size_t)g.ldb;
    // queue loads through buffer descriptors (round 5): lane byte offset loop-invariant, the tile's k offset scalar — no 64-bit
    // address per load (the v_lshl_add_u64 pair per tile sat between the MFMAs, and its temporaries aliased queue registers:
    // generic family 47.5 -> 45.4 us per launch, step -0.6 ms).  Unbounded descriptors: every offset is clamped into the
    // operand; the launchers refuse operands of 4 GiB or more (fits32).
    const uint32_t kbytes_a = (uint32_t)kstride_a * 4u, kbytes_b = (uint32_t)kstride_b * 4u;
    const __amdgpu_buffer_rsrc_t rsQA = buf_rsrc(g.A, 0xFFFFFFFFu), rsQB = buf_rsrc(g.B, 0xFFFFFFFFu);
#define GF_GLOAD(U, T)                                                                                   \
    {                                                                                                    \
        const int k0 = kbeg + max(min((T), nfull - 1), 0) * BK;        /* uniform: scalar ALU */         \
        qa##U = buf_load_f4(rsQA, 4u * offa, (uint32_t)k0 * kbytes_a);                                   \
        qb##U = buf_load_f4(rsQB, 4u * offb, (uint32_t)k0 * kbytes_b);                                   \
        if constexpr (B2) qc##U = buf_load_f4(rsQB, 4u * offc, (uint32_t)k0 * kbytes_b);                 \
    }
#define GF_SSTORE_FULL(U, BUF)                                                                           \
    {                                                                                                    \
        float* const sa = smem + (BUF) * SM::STAGE;                                                      \
        *reinterpret_cast<float4*>(sa + ldsa) = qa##U;                                                   \
        *reinterpret_cast<float4*>(sa + SM::A_FLOATS + ldsb) = qb##U;                                    \
        if constexpr (B2) *reinterpret_cast<float4*>(sa + SM::A_FLOATS + ldsc) = qc##U;                  \
        if (want_colsum) { colsum4.x += qa##U.x; colsum4.y += qa##U.y; colsum4.z += qa##U.z; colsum4.w += qa##U.w; } \
    }
    auto sstore_tail = [&](int buf) __attribute__((always_inline)) {
        float* sa = smem + buf * SM::STAGE;
        float* sb = sa + SM::A_FLOATS;
        if (MODE == MODE_TN) {
            tail_a_km.store(sa, tid);
            if (want_colsum) tail_a_km.add_to(colsum4, tid);
        } else {
            tail_a_kc.store(sa, tid);
        }
        if (MODE == MODE_NT) tail_b_kc.store(sb, tid); else tail_b_km.store(sb, tid);
    };

    // One K tile: ALL operand fragments of the tile are read from LDS first (BK/8 groups x (TM + TN) reads), then the
    // BK/2 x TM x TN MFMAs run back to back.  No run-time condition in here: a K tail is zero-filled by the loaders.
    auto compute = [&](int buf) __attribute__((always_inline)) {
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

    // prologue: the tail tile and PD full tiles in flight, tile 0 to LDS
    {
        const int k0 = kbeg + nfull * BK;      // >= kend when there is no tail: the masked loaders then produce zeros
        if (MODE == MODE_TN) tail_a_km.load(g.A, g.lda, m0, g.M, k0, kend, tid); else tail_a_kc.load(g.A, g.lda, m0, g.M, k0, kend, tid);
        if (MODE == MODE_NT) tail_b_kc.load(g.B, g.ldb, n0, g.N, k0, kend, tid); else tail_b_km.load(g.B, g.ldb, n0, g.N, k0, kend, tid);
    }
    GF_GLOAD(0, 0)
    if constexpr (PD > 1) GF_GLOAD(1, 1)
    if constexpr (PD > 2) GF_GLOAD(2, 2)
    if constexpr (PD > 3) GF_GLOAD(3, 3)
    // the epilogue's dropout keep bits (Philox: ~70 VALU instructions per call, 4 calls per tile) depend on indices only:
    // evaluated here, under the latency of the first global loads, instead of on the critical path after the last MFMA
    DropCtx dc;
    uint32_t keep[TM][TN];
    if constexpr (MODE != MODE_TN) {
        if (epi_has_dropout<EPI>()) dc = make_drop(g.ea.rng, g.ea.rng_add, g.ea.site, g.ea.p, g.ea.train);
        else dc.on = 0;
        gemm_keep_bits<EPI, TM, TN>(g, dc, keep, m0 + wm * WM, n0 + wn * WN, r, h);
    }
    if (nfull > 0) GF_SSTORE_FULL(0, 0) else sstore_tail(0);
    __syncthreads();

    // Steady state, iteration t: issue the global loads of tile t+PD, run tile t from LDS stage t&1, write tile t+1
    // (loaded PD-1 iterations ago) to the other stage, barrier.  The step body is straight-line code; the only branches
    // are the wave-uniform "is there a tile t" around a whole step and "is tile t+1 the tail" around its LDS stores.
#define GF_STEP(U, UN)                                                                  \
    if (t0 + U < nt) {                                                                  \
        GF_GLOAD(U, t0 + U + PD)                                /* slot U is free */    \
        compute(U & 1);                                                                 \
        /* tile t+1 -> LDS: from the queue while it is a full tile, the tail tile right after the last full one */ \
        if (t0 + U + 1 == nfull) sstore_tail((U + 1) & 1);      /* wave-uniform; LDS stores only */ \
        else if (t0 + U + 1 < nfull) GF_SSTORE_FULL(UN, (U + 1) & 1)   /* (nothing beyond the last tile) */ \
        __syncthreads();                                                                \
    }
    for (int t0 = 0; t0 < nt; t0 += PD) {
        if constexpr (PD == 2) {
            GF_STEP(0, 1)
            GF_STEP(1, 0)
        } else {
            GF_STEP(0, 1)
            GF_STEP(1, 2)
            GF_STEP(2, 3)
            GF_STEP(3, 0)
        }
    }
#undef GF_STEP
#undef GF_GLOAD
#undef GF_SSTORE_FULL

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
    // XCD-aware tile order (round 4).  The dispatcher deals workgroups round-robin over the 8 XCDs in linear-id order
    // (x fastest), and every launch of the d_model-512 generator has a tile count along N that is a multiple of 8 (8, 24,
    // 32): each XCD then owned a few N columns of EVERY row tile, i.e. every XCD pulled the whole activation operand through
    // its private L2 — rocprofv3 FETCH_SIZE 4.1x the compulsory bytes over the generator's launch mix
    // (profiles/r04_gemm_traffic.json: 121 MB against 30 MB per launch), worst for the K = 2048 products (24.6 MB of
    // activations read 8 times).  Bijective remap as in gemm_tn_grouped_kernel: the workgroups of one XCD get a contiguous
    // range of the logical (k-split, row tile, column tile) list with the column fastest = a band of row tiles with all
    // their columns: the activation band is read once, only the (smaller) weight is read by all eight.
    const int gx = gridDim.x, gy = gridDim.y, per = gx * gy, nwg = per * gridDim.z;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int q8 = nwg >> 3, r8 = nwg & 7, xl = lin & 7;
    const int b = (xl < r8 ? xl * (q8 + 1) : r8 * (q8 + 1) + (xl - r8) * q8) + (lin >> 3);
    const int bz = b / per, t = b - bz * per, by = t / gx, bx = t - by * gx;
    gemm_body<MODE, BM, BN, BK, EPI, WGM, WGN>(g, bx, by, bz);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-resident short-K GEMM (K = KC = 100: linear1 forward and linear2's dgrad of the d_model-100 feed-forward block,
// the two [T x 100] -> [T x 2048] products with fused epilogues; /root/reference/model.py:1210 -> torch _ff_block).
//
// With the generic kernel this shape is 1504 one-tile workgroups for 1536 resident slots: every workgroup pays its own
// load latency, pulls both operand tiles through L2 (86 MB per launch) and runs in lockstep with all the others.  Here a
// workgroup is PERSISTENT over several 64-token tiles of one 64-column panel of the weight:
//   * the wave's weight fragments (32 columns x K) are loaded from global memory ONCE, straight into the MFMA operand
//     layout, and stay in 50 VGPRs — the weight never passes through LDS and is not re-read per tile;
//   * the token tile (64 x K) is double-buffered in LDS through a register prefetch: tile i+1's global loads are issued
//     before tile i's MFMAs and written to the other LDS buffer after them — one barrier per tile, the load latency is
//     paid once per workgroup, and the epilogue's stores of tile i drain under tile i+1's MFMAs;
//   * K = 100 is 13 groups of 8: 52 MFMAs per wave and tile (the generic BK = 16 loop runs 56), in gemm_body's k order,
//     so both kernels give the same bits;
//   * the grid is exactly the number of workgroups the device holds at once (occupancy query, cached).
// Measured (tools/lab/ffn_gemm_lab.py, T = 3008 / 6016, train mode): 22.0 / 35.8 us against 22.1 / 39.3 us for the generic
// kernel in the step.  What bounds it (ablation in the same lab): launch + operand loads 6.6 us, MFMA chain 7.7 us (= its
// ideal), epilogue VALU + the 24.6 MB store 6.3 us, and these ADD — fp32 MFMAs execute on the SIMD's vector ALU
// (tools/lab/mfma_valu.hip), so Philox / epilogue VALU work cannot hide behind MFMAs of the same or of another wave.
// A software-pipelined variant (previous tile's epilogue and this tile's Philox rounds interleaved between the MFMAs,
// 2 waves per SIMD at 200 VGPRs) was built and measured slower (22.3 / 37.0 us); this is the simple form.
// Same MFMA (v_mfma_f32_32x32x2_f32, exact fp32) and the same epilogue formulas as gemm_body.
GF_LAB_ONLY(unsigned long long* g_wres_stamps = nullptr;)    // lab builds only (make LAB=1): 4 x uint64 of in-kernel time stamps per workgroup

template <int MODE, int EPI, int KC>
__global__ __launch_bounds__(256) void gemm_wres_kernel(GemmArgs g, int mtiles, int wg_per_panel GF_LAB_ONLY(, unsigned long long* stamps)) {
    static_assert(KC % 4 == 0 && KC <= 128, "short K only");
    constexpr int BM = 64, BN = 64;
    constexpr int G8 = (KC + 7) / 8;                              // groups of 8 along k; k >= KC carries zero weights
    constexpr int LD = 4 * ((KC / 4) | 1);                        // LDS row stride: 4 x odd floats (conflict-free b128 reads)
    constexpr int KV = KC / 4, TOTALV = BM * KV, NV = (TOTALV + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int panel = blockIdx.x / wg_per_panel, j0 = blockIdx.x - panel * wg_per_panel;
    const int n0 = panel * BN;
    GF_LAB_ONLY(unsigned long long* const stamp = (stamps && tid == 0) ? stamps + 4 * (size_t)blockIdx.x : nullptr;
                if (stamp) stamp[0] = __builtin_amdgcn_s_memtime();)

    // weight fragments: MFMA j of group gq takes k = 8 gq + 4 h + j — the k order of gemm_body, so this kernel's results
    // are bit-identical to the generic kernel's (a K tail is zero weights against clamped, finite token values)
    // KC = 8 q + 4 (100): the last four k values fill only the h = 0 half of a ninth... thirteenth group of 8 — four MFMAs at
    // half use.  They go into TWO MFMAs instead, lane half h taking k = KC - 4 + 2 j + h at MFMA j: the same fma chain in the
    // same k order (a product with a zero weight adds nothing), 50 MFMAs per tile instead of 52
    constexpr bool TAIL2 = (KC % 8) == 4;
    constexpr int G8F = TAIL2 ? KC / 8 : G8;                       // groups of 8 handled by the 4-MFMA loop
    float wf[G8][4];
    float wtail[2] = {0.f, 0.f};
    {
        const int n = min(n0 + wn * 32 + r, g.N - 1);
        if constexpr (TAIL2) {
            if (MODE == MODE_NT) {
                const float4 q = *reinterpret_cast<const float4*>(g.B + (size_t)n * g.ldb + KC - 4);
                wtail[0] = h ? q.y : q.x;
                wtail[1] = h ? q.w : q.z;
            } else {
                wtail[0] = g.B[(size_t)(KC - 4 + h) * g.ldb + n];
                wtail[1] = g.B[(size_t)(KC - 2 + h) * g.ldb + n];
            }
        }
#pragma unroll
        for (int gq = 0; gq < G8F; ++gq) {
            const int k = 8 * gq + 4 * h;                       // K % 4 == 0: a group half is inside K or outside it
            const float keepw = k < KC ? 1.f : 0.f;
            const int kc = min(k, KC - 4);
            if (MODE == MODE_NT) {
                const float4 q = *reinterpret_cast<const float4*>(g.B + (size_t)n * g.ldb + kc);
                wf[gq][0] = q.x * keepw; wf[gq][1] = q.y * keepw; wf[gq][2] = q.z * keepw; wf[gq][3] = q.w * keepw;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[gq][j] = g.B[(size_t)(kc + j) * g.ldb + n] * keepw;
            }
        }
    }

    // token-tile staging registers: NAMED vectors, not an array — hipcc left a float4[7] that lives across the tile
    // loop in scratch memory (one scratch round trip per load)
    static_assert(NV == 7, "staging code below is written out for 7 vectors per thread (64 x 100 floats, 256 threads)");
    float4 ta0, ta1, ta2, ta3, ta4, ta5, ta6;
#define GF_REP7(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6)
#define GF_WRES_IDX(J)                                                                                   \
    const int ti##J = min(tid + J * 256, TOTALV - 1); /* clamped duplicates rewrite the last vector */   \
    const int trow##J = ti##J / KV, tcol##J = 4 * (ti##J - trow##J * KV), soff##J = trow##J * LD + tcol##J;
    GF_REP7(GF_WRES_IDX)
    // token-tile loads through a buffer descriptor (round 5): the lane's offset inside a tile is loop-invariant, the tile's first
    // row is one 32-bit add, and rows past M are out of the descriptor's range — the hardware returns zeros for them (they feed
    // output rows that are never stored) instead of a per-lane min() + 64-bit address per load
    const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(g.A, (uint32_t)(((size_t)(g.M - 1) * g.lda + KC) * sizeof(float)));
    // (the tile's first row is added to the LANE offset, one v_add per load: the range check covers the lane offset only, a
    //  scalar offset would slip past it)
    const uint32_t ldab = (uint32_t)g.lda * 4u;
#define GF_WRES_VOFF(J) const uint32_t tvo##J = ((uint32_t)trow##J * (uint32_t)g.lda + (uint32_t)tcol##J) * 4u;
    GF_REP7(GF_WRES_VOFF)
#define GF_WRES_GLOAD1(J) ta##J = buf_load_f4(rsA, tvo##J + so_, 0u);
#define GF_WRES_GLOAD(MT) { const uint32_t so_ = (uint32_t)((MT) * BM) * ldab; GF_REP7(GF_WRES_GLOAD1) }
#define GF_WRES_SSTORE1(J) *reinterpret_cast<float4*>(sdst + soff##J) = ta##J;
#define GF_WRES_SSTORE(BUF) { float* const sdst = smem + (BUF) * (BM * LD); GF_REP7(GF_WRES_SSTORE1) }

    DropCtx dc;
    if (epi_has_dropout<EPI>()) dc = make_drop(g.ea.rng, g.ea.rng_add, g.ea.site, g.ea.p, g.ea.train);
    else dc.on = 0;
    const int nbase = n0 + wn * 32;
    float bias = 0.f;                                           // the wave's columns never change: one load per workgroup
    if (EPI == EPI_NONE || EPI == EPI_RELU_DROP || EPI == EPI_DROP_GELU)
        if (g.ea.bias != nullptr) bias = g.ea.bias[min(nbase + r, g.N - 1)];

    int mt = j0;
    if (mt >= mtiles) return;                                   // (the host never launches such a workgroup)
    GF_WRES_GLOAD(mt)
    GF_WRES_SSTORE(0)
    __syncthreads();
    GF_LAB_ONLY(if (stamp) stamp[1] = __builtin_amdgcn_s_memtime();)

    // One tile's 4 * G8 MFMAs with their LDS fragment reads.  (Round 5 also wrote them as inline assembly accumulating in place in
    // ordinary vector registers, to save the 16 v_accvgpr_read of the epilogue: wrong results on the GPU although the emitted
    // sequence reads like hipcc's own — not pursued for 16 instructions per tile; the builtin stays.)
#define GF_WRES_MFMA1(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, ACC, 0, 0, 0);
#define GF_WRES_MFMA(ACC, BUF)                                                                              \
    {                                                                                                       \
        const float* arow = smem + (BUF) * (BM * LD) + (wm * 32 + r) * LD;                                  \
        _Pragma("unroll") for (int gq = 0; gq < G8F; ++gq) {                                                \
            const float4 q = *reinterpret_cast<const float4*>(arow + min(8 * gq + 4 * h, KC - 4));          \
            if (gq == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) ACC[i_] = 0.f; GF_WRES_MFMA1(ACC, q.x, wf[gq][0]) } \
            else GF_WRES_MFMA1(ACC, q.x, wf[gq][0])                                                          \
            GF_WRES_MFMA1(ACC, q.y, wf[gq][1])                                                               \
            GF_WRES_MFMA1(ACC, q.z, wf[gq][2])                                                               \
            GF_WRES_MFMA1(ACC, q.w, wf[gq][3])                                                               \
        }                                                                                                   \
        if constexpr (TAIL2) {                                                                              \
            const float4 q = *reinterpret_cast<const float4*>(arow + KC - 4);                               \
            GF_WRES_MFMA1(ACC, h ? q.y : q.x, wtail[0])                                                      \
            GF_WRES_MFMA1(ACC, h ? q.w : q.z, wtail[1])                                                      \
        }                                                                                                   \
    }
    // one tile from LDS stage BUF (a compile-time constant: the loop below is unrolled by two)
#define GF_WRES_TILE(BUF)                                                                                   \
    {                                                                                                       \
        const int mbase = mt * BM + wm * 32;                                                                \
        const int mload = min(mt + wg_per_panel, mtiles - 1);   /* unconditional (clamped): no load under a branch */ \
        GF_WRES_GLOAD(mload)                                                                                \
        float aux[1][1][16];                                                                                \
        gemm_load_aux<EPI, 1, 1>(g, aux, mbase, nbase, 0, r, h);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        uint32_t keep[1][1];                                                                                \
        gemm_keep_bits<EPI, 1, 1>(g, dc, keep, mbase, nbase, r, h);                                         \
        floatx16 acc[1][1];                                                                                 \
        GF_WRES_MFMA(acc[0][0], BUF)                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        GF_WRES_SSTORE((BUF) ^ 1)                                                                           \
        if (mbase + 32 <= g.M && nbase + 32 <= g.N) {           /* wave-uniform */                          \
            gemm_apply_store_full<EPI>(g, acc[0][0], aux[0][0], bias, mbase, nbase, r, h, dc, keep[0][0]);  \
        } else {                                                                                            \
            gemm_apply_store<EPI, 1, 1>(g, acc, aux, mbase, nbase, 0, r, h, dc, keep);                      \
        }                                                                                                   \
        __syncthreads();                                                                                    \
    }
#pragma unroll 1
    for (; mt < mtiles; mt += wg_per_panel) {
        GF_WRES_TILE(0)
        mt += wg_per_panel;
        if (mt >= mtiles) break;
        GF_WRES_TILE(1)
    }
#undef GF_WRES_TILE
#undef GF_WRES_MFMA1
    GF_LAB_ONLY(if (stamp) { stamp[2] = __builtin_amdgcn_s_memtime(); stamp[3] = (unsigned long long)((mtiles - j0 + wg_per_panel - 1) / wg_per_panel); })
#undef GF_WRES_MFMA
#undef GF_WRES_GLOAD
#undef GF_WRES_SSTORE
#undef GF_WRES_GLOAD1
#undef GF_WRES_SSTORE1
#undef GF_WRES_IDX
#undef GF_WRES_VOFF
#undef GF_REP7
}

template <int MODE, int EPI>
static int launch_wres(const GemmArgs& g, hipStream_t st) {
    constexpr int KC = 100;
    constexpr size_t lds = 2 * 64 * 4 * ((KC / 4) | 1) * sizeof(float);
    const int panels = (g.N + 63) / 64, mtiles = (g.M + 63) / 64;
    GF_TRY((lds_optin<gemm_wres_kernel<MODE, EPI, KC>>(lds, "gemm_wres")));
    // persistent grid = exactly the workgroups the device holds at once (queried once per kernel and device)
    static thread_local int resident[16] = {0};
    int devid = 0;
    GF_HIP(hipGetDevice(&devid));
    int& res = resident[devid & 15];
    if (res == 0) {
        int per_cu = 0, cus = 0;
        GF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_wres_kernel<MODE, EPI, KC>, 256, lds));
        GF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid));
        // (two per CU at most: with the accumulator in ordinary registers three would fit, and three measured slower in
        //  round 3 — fewer tiles per workgroup to amortise the weight fragments and the first load)
        if (per_cu > 2) per_cu = 2;
        res = per_cu * cus > 0 ? per_cu * cus : 512;
    }
    int per = res / panels;
    if (per < 1) per = 1;
    if (per > mtiles) per = mtiles;
    hipLaunchKernelGGL((gemm_wres_kernel<MODE, EPI, KC>), dim3(panels * per), dim3(256), lds, st, g, mtiles, per GF_LAB_ONLY(, g_wres_stamps));
    GF_LAUNCH_CHECK();
    return 0;
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
    long part_off;           // this problem's [M x N | M] partial block inside a split's slab
};
struct TnGroup {
    TnProblem p[MAXP];
    int n;
    float* part;             // partial-slab workspace (nullptr: one owner workgroup per tile adds in place)
    long part_stride;        // floats per split slab
    int splits;
};

// SPLIT = false: no partial-slab code in the kernel at all (with it the kernel needs 100 instead of 96 VGPRs = 4 instead
// of 5 workgroups per CU, and the 1136-tile d_model-100 group no longer fits the chip in one round: 446 -> 470 us)
template <bool SPLIT, int BN = 64>
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
    if constexpr (SPLIT) { g.part = grp.part + q.part_off; g.part_stride = grp.part_stride; }
    // tiles in panels of 8 along N (n fastest inside a panel, then m, then the next panel): an XCD's contiguous
    // range of ~32 tiles is then a 4 x 8 patch — 12 operand panels instead of the 33 of a 1 x 32 strip
    constexpr int PW = 8;
    const int tiles_m = q.tiles_mn / q.tiles_n;
    const int panel = t2 / (PW * tiles_m), rem = t2 - panel * PW * tiles_m;
    const int pw = min(PW, q.tiles_n - panel * PW);
    const int mt = rem / pw, nt = panel * PW + rem - mt * pw;
    gemm_body<MODE_TN, 64, BN, 16, EPI_NONE, 2, 2>(g, nt, mt, bz);
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
    if constexpr (MODE != MODE_TN)
        if (g.K == 100 && g.N >= 1024 && splits == 1 && (g.lda & 3) == 0 && ((MODE == MODE_NT) ? (g.ldb & 3) == 0 : true) &&
            (size_t)g.M * g.lda * 4 < (size_t(1) << 31) && (size_t)g.M * g.ldc * 4 < (size_t(1) << 31))   // 32-bit buffer offsets
            return launch_wres<MODE, EPI>(g, st);
    if (g.K <= 128) return launch_cfg<MODE, 64, 64, 16, EPI, 2, 2, 1>(g, splits, st);
    // (64 x 128 tiles for the N = 2048, K = 512 products: 5-9 % faster in isolation, round 2; in the 3-stream step 35.9 against
    // 35.7 ms with 64 x 64 — fewer, fatter workgroups co-run worse — so they are not used; round 3, tools/lab/mode_ab.py;
    // re-measured in round 4 with tuned streams and the XCD-aware tile order: 34.71 against 34.43 ms, three interleaved runs each)
    return launch_cfg<MODE, 64, 64, 16, EPI>(g, splits, st);
}

// the K loop reads its operands through buffer descriptors with 32-bit byte offsets (round 5): an operand must span < 4 GiB
static inline bool fits32(long rows, long ld) { return (unsigned long long)rows * (unsigned long long)ld * 4ull < (1ull << 32); }

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
    GF_CHECK_ARG(fits32(M, lda) && fits32(N, ldw), "gemm_nt: an operand of 4 GiB or more is not supported");
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
    GF_CHECK_ARG(fits32(M, lda) && fits32(K, ldb), "gemm_nn: an operand of 4 GiB or more is not supported");
    GemmArgs g{A, lda, Bm, ldb, C, ldc, nullptr, M, N, K, K, 0, ea};
    int splits = (splits_io && epi == EPI_NONE) ? *splits_io : 1;
    set_split(g, splits, slab_stride);
    if (splits_io) *splits_io = splits;
    EPI_SWITCH(MODE_NN, g, st)
}

// A one- or two-tile gradient (the discriminator head's fc1 [64 x 100] / fc2 [16 x 64]) used to be cut into K / 64 = 94
// slabs: the 6 us GEMM was followed by a 40 us reduce launch whose 2-7 workgroups walk 94 slabs one batch of loads after
// the other (profiles/r02_*: tn_reduce_kernel 1.4 % of the step).  16 slabs keep the GEMM as short and the reduce at its floor.
constexpr long TN_ACC_MAXSPLIT = 16;

// floats of partial-slab workspace a split TN launch of this shape can use (0: it will not split)
long gemm_tn_part_floats(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    long splits = (768 + tiles - 1) / tiles;             // ~3 blocks per CU measured best (tools/gemm_bench.py)
    const long maxsplits = (K + 63) / 64;                // at least 64 k per block
    if (splits > maxsplits) splits = maxsplits;
    if (splits > TN_ACC_MAXSPLIT) splits = TN_ACC_MAXSPLIT;
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
    GF_CHECK_ARG(fits32(K, lda) && fits32(K, ldb), "gemm_tn: an operand of 4 GiB or more is not supported");
    EpiArgs ea;
    GemmArgs g{At, lda, Bm, ldb, C, ldc, colsum, M, N, K, K, 0, ea};
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    const long per = (long)M * N + M;
    long splits = (768 + tiles - 1) / tiles;
    const long maxsplits = (K + 63) / 64;
    if (splits > maxsplits) splits = maxsplits;
    if (splits > TN_ACC_MAXSPLIT) splits = TN_ACC_MAXSPLIT;
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

// A grouped launch with fewer than TN_GROUP_MIN_TILES output tiles (4 per CU) splits the token range of every tile so that
// about TN_GROUP_TILES workgroups exist.  Measured at 1136 tiles (a whole d_model-100 backward pass, T = 6016): unsplit
// 446 us; split in 3: 424 us + 20 us for the reduce launch — no gain, so such groups stay on the owner-only path; the
// narrow groups this is for are the 1-4-layer gradient buckets of the data-parallel path (142-568 tiles).
constexpr int TN_GROUP_MIN_TILES = 1024;
constexpr int TN_GROUP_TILES = 2560;
constexpr int TN_GROUP_MAXSPLIT = 8;

// C_i += sum_z part[z][i], colsum_i += sum_z part[z][M N + i] for every problem of a split grouped launch, slabs in split
// order; blockIdx.y = problem, blockIdx.x strides over its elements
__global__ __launch_bounds__(256) void tn_reduce_grouped_kernel(TnGroup grp) {
    const TnProblem& q = grp.p[blockIdx.y];
    const float* part = grp.part + q.part_off;
    const long nC = (long)q.M * q.N;
    for (long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i4 < nC; i4 += (long)gridDim.x * 1024) {
        float4 v[TN_GROUP_MAXSPLIT];                             // all slabs' loads in flight together
#pragma unroll
        for (int z = 0; z < TN_GROUP_MAXSPLIT; ++z)
            v[z] = *reinterpret_cast<const float4*>(part + (size_t)min(z, grp.splits - 1) * grp.part_stride + i4);
        const long row = i4 / q.N, col = i4 - row * q.N;       // N % 4 == 0: a float4 stays inside one row
        float* dst = q.C + row * q.ldc + col;
        const float4 old = *reinterpret_cast<const float4*>(dst);
        float4 s = v[0];
#pragma unroll
        for (int z = 1; z < TN_GROUP_MAXSPLIT; ++z)
            if (z < grp.splits) { s.x += v[z].x; s.y += v[z].y; s.z += v[z].z; s.w += v[z].w; }
        *reinterpret_cast<float4*>(dst) = make_float4(old.x + s.x, old.y + s.y, old.z + s.z, old.w + s.w);
    }
    if (q.colsum != nullptr)
        for (int i = blockIdx.x * 256 + threadIdx.x; i < q.M; i += gridDim.x * 256) {
            float s = part[nC + i];
            for (int z = 1; z < grp.splits; ++z) s += part[(size_t)z * grp.part_stride + nC + i];
            q.colsum[i] += s;
        }
}

long gemm_tn_grouped_part_floats() { return (long)(TN_GROUP_TILES + 256) * (64 * 64 + 64) + 4 * MAXP; }

// dW_i[M_i x N_i] += At_i^T B_i for n problems in one launch (see gemm_tn_grouped_kernel).  Groups of >= 1024 tiles (one
// encoder backward pass: 6144 tiles at d_model 512, 1136 at 100): every output tile has one owner workgroup that runs the
// whole token range and adds its result in place.  Narrower groups (a 2-layer gradient bucket: 284 tiles on 256 CUs) with
// a workspace: the token range is split, the partial tiles go to per-split slabs and one reduce launch adds them in split
// order.  Deterministic either way.

int launch_gemm_tn_grouped(const TnDesc* d, int n, hipStream_t st, float* part_ws, long part_floats, TnSlabs* slabs) {
    GF_CHECK_ARG(d && n >= 1 && n <= MAXP, "gemm_tn_grouped: n=%d out of [1,%d]", n, MAXP);
    // every problem 100-wide on one side (a d_model-100 encoder pass): 112-wide 16x16x4 tiles (gemm_tn100.hip)
    if (!mode().tn100_off() && part_ws != nullptr && aligned16(part_ws) && tn100_supported(d, n))
        return launch_gemm_tn100_grouped(d, n, st, part_ws, part_floats, slabs);
    GF_CHECK_ARG(slabs == nullptr, "gemm_tn_grouped: unreduced gradient slabs exist only for the d_model-100 kernel");
    TnGroup grp;
    grp.n = n;
    // 64 x 128 tiles (two accumulators per wave sharing the A fragment: the single-accumulator MFMA chain of the 64 x 64
    // tile is issue-limited) when every problem's N is a multiple of 128 and the group keeps >= 4 such tiles per CU.
    // Measured (tools/lab/tn_wide.py, same bits): the d_model-512 group 1270 -> 1212 us; the d_model-100 group would drop
    // to 568 tiles (N = 100 -> one 128-wide tile) and gets slower (543 -> 639 us), so it stays on 64 x 64.
    bool wide = true;
    long wtiles = 0;
    for (int i = 0; i < n; ++i) {
        GF_CHECK_ARG(fits32(d[i].K, d[i].lda) && fits32(d[i].K, d[i].ldb), "gemm_tn_grouped: an operand of 4 GiB or more is not supported");
        wide = wide && (d[i].N % 128 == 0);
        wtiles += (long)((d[i].M + 63) / 64) * ((d[i].N + 127) / 128);
    }
    wide = wide && wtiles >= TN_GROUP_MIN_TILES;
    const int BNs = wide ? 128 : 64;
    long tiles = 0, per_split = 0;
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        GF_TRY(check_common(d[i].At, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].M, d[i].N, d[i].K));
        GF_CHECK_ARG((d[i].M & 3) == 0 && (d[i].N & 3) == 0, "gemm_tn_grouped: M, N must be multiples of 4");
        tiles += (long)((d[i].M + 63) / 64) * ((d[i].N + BNs - 1) / BNs);
        per_split += (((long)d[i].M * d[i].N + d[i].M) + 3) & ~3L;
        kmax = d[i].K > kmax ? d[i].K : kmax;
        if (!aligned16(d[i].C) || (d[i].ldc & 3) != 0) part_ws = nullptr;     // the reduce adds 16-byte vectors in place
    }
    int splits = 1;
    if (part_ws != nullptr && tiles < TN_GROUP_MIN_TILES) {
        splits = (int)((TN_GROUP_TILES + tiles - 1) / tiles);
        if (splits > TN_GROUP_MAXSPLIT) splits = TN_GROUP_MAXSPLIT;
        if (splits > kmax / 256) splits = kmax / 256;             // >= 256 tokens per workgroup of the longest problem
        if ((long)splits * per_split > part_floats) splits = (int)(part_floats / per_split);
        if (splits < 2) splits = 1;
    }
    grp.splits = splits;
    grp.part = splits > 1 ? part_ws : nullptr;
    grp.part_stride = per_split;
    GF_CHECK_ARG(splits == 1 || aligned16(part_ws), "gemm_tn_grouped: partial-slab workspace must be 16-byte aligned");
    int total = 0;
    long off = 0;
    for (int i = 0; i < n; ++i) {
        TnProblem& q = grp.p[i];
        q.A = d[i].At; q.B = d[i].B; q.C = d[i].C; q.colsum = d[i].colsum;
        q.lda = d[i].lda; q.ldb = d[i].ldb; q.ldc = d[i].ldc; q.M = d[i].M; q.N = d[i].N; q.K = d[i].K;
        // chunk: a multiple of 64 tokens; problems with different K in one group simply get fewer non-empty splits
        q.kchunk = splits > 1 ? (int)((((long)d[i].K + splits - 1) / splits + 63) / 64 * 64) : d[i].K;
        const int tm = (d[i].M + 63) / 64;
        q.tiles_n = (d[i].N + BNs - 1) / BNs;
        q.tiles_mn = tm * q.tiles_n;
        q.block0 = total;
        q.part_off = off;
        off += (((long)d[i].M * d[i].N + d[i].M) + 3) & ~3L;
        total += q.tiles_mn * splits;
    }
    if (wide && splits == 1) {
        constexpr size_t ldsw = Smem<MODE_TN, 64, 128, 16>::TOTAL * sizeof(float);
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, 128>), dim3(total), dim3(256), ldsw, st, grp);
        GF_LAUNCH_CHECK();
        return 0;
    }
    constexpr size_t lds = Smem<MODE_TN, 64, 64, 16>::TOTAL * sizeof(float);
    if (splits > 1) hipLaunchKernelGGL(gemm_tn_grouped_kernel<true>, dim3(total), dim3(256), lds, st, grp);
    else hipLaunchKernelGGL(gemm_tn_grouped_kernel<false>, dim3(total), dim3(256), lds, st, grp);
    GF_LAUNCH_CHECK();
    if (splits > 1) {
        hipLaunchKernelGGL(tn_reduce_grouped_kernel, dim3(64, n), dim3(256), 0, st, grp);
        GF_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace ganffn
