// gemm_tn100.hip — the grouped weight-gradient launch of a d_model-100 encoder backward pass on 112-wide tiles.
//
// Every weight gradient of a d_model-100 layer has a 100-wide dimension: dW2 [100 x 2048], dW1 [2048 x 100], dWo [100 x 100],
// dWin [300 x 100] (dW = dY^T X over the tokens; /root/reference/model.py:1210 -> the backward of torch's
// TransformerEncoderLayer).  On the 64 x 64 tiles of gemm.hip that dimension is 2 tiles = 128: 22 % of every MFMA of the
// launch that dominates the step's GPU time is padding (profiles/r02_*: 447 us per launch, 49 % of the fp32 MFMA peak on
// useful FLOPs although the MFMAs it executes run at 80 %).  Here, as in gemm_n100.hip:
//  * v_mfma_f32_16x16x4_f32 (exact fp32) with the 100-wide dimension on the m axis: 6 tiles of 16 + rows 96..99 on
//    v_mfma_f32_4x4x1_16B_f32 (round 4, TAIL4: 461.6 against 491.3 us per launch at T = 6016, 246.4 against 261.3 us at
//    T = 3008 — tools/lab/tn100_tail4.py; until then a seventh tile: 112 = 10.7 % padding);
//  * a wave owns all 7 of them for 16 columns of the other dimension: the 7 MFMAs of a k-step share one operand;
//  * a workgroup = 4 waves = a 112 x 64 output tile; k = the token axis, read from LDS tiles of 32 tokens (both operands
//    are token-major, so both are read with ds_read_b32; row strides 116 / 68 floats put rows 4 apart 16 banks apart);
//  * the token range is cut into 2..8 chunks so that the launch has several times 768 workgroups (3 resident per CU: the
//    K loop then runs at > 90 % MFMA utilisation, gemm_n100.hip) and the CUs finish within one short workgroup of each
//    other; chunk z writes its partial tile to slab z and one ordered reduce launch adds the slabs to the gradient —
//    no atomics, bit-reproducible.  (Round 4 built the alternative — the LAST workgroup of a tile to arrive, decided by an
//    integer ticket, adds the slabs in chunk order inside tn100_kernel: same bits, tests/test_hip_ops.py — and measured it
//    SLOWER both ways it can be written: with a `__threadfence()` per thread 731 against 490 us at T = 6016 (every fence
//    writes back and invalidates an L2); with the one-lane agent-scope release / acquire of cdna_hip_programming.md 515
//    against 491 us at T = 6016 and 296 against 261 us at T = 3008 (gpurun_out/r4_tn100_lab*.txt): a tile's slabs are
//    3 x 28 KB, and the last arriver reads them serially at the end of a workgroup's life, which costs more than the 21 us
//    reduce launch that reads all of them with the whole chip.  It stays behind ganffn_debug_set_ffn_mode bit 4.)
//    (Round 4 also built 128-column workgroup tiles — two column groups per wave sharing the 28 row-operand registers, 28
//    instead of 20 flop per byte staged into LDS — and measured them SLOWER: 503 against 466 us at T = 6016, 269 against
//    250 us at T = 3008 (20 back-to-back launches, HIP events): 208-228 VGPRs and 63 KB of LDS leave two waves per SIMD instead of three.)
//  * bias gradients (column sums of dY over the tokens) are accumulated from the operand registers the MFMAs read anyway;
//  * workgroup ids are remapped so that one XCD (one L2) gets a contiguous range of the (problem, chunk, tile) list: the
//    tiles of a problem share its 100-wide operand panel ([T x 100], 2.4 MB at T = 6016) through that L2, the wide operand
//    is read exactly once.
#include "common.h"

namespace ganffn {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WE = 100, WT7 = 7, WBK = 32, WBN = 64;
constexpr int LDU = 116, LDV = 68;            // = 4 (mod 8): ds_read_b32 of rows g and g + 1 (4 k apart... see above) conflict-free
constexpr int WMAXP = 40, WMAXSPLIT = 8;

struct W100Problem {
    const float* U; const float* V;            // U: [K x 100] (the 100-wide operand), V: [K x Nn]
    float* C; float* colsum;                   // gradient [M x N] (ldc) and bias gradient [M] (or null)
    int ldu, ldv, ldc, Nn;
    int side;                                  // 0: U = dY (C rows = the 100-wide dim), 1: U = X (C columns = the 100-wide dim)
    int K, kchunk;
    int ntiles, block0;                        // 64-wide tiles over Nn; first workgroup of this problem
    int M, N;                                  // the gradient's shape (partial-slab layout: [M x N] dense, then [M] column sums)
    long part_off;
    int ctr0;                                  // first arrival counter of this problem (one per 64-wide tile)
    long c_off, cs_off;                        // to_slabs: offsets of C / colsum from the gradient slab's base
};
struct W100Group {
    W100Problem p[WMAXP];
    int n, splits;
    float* part; long part_stride;
    int* counters;                             // per (problem, tile) arrival tickets, zero at launch; null: separate reduce launch
    // to_slabs (round 5, single-GPU step): NO reduce launch and NO zeroed gradient — token chunk 0 OVERWRITES the gradient
    // (C / colsum), chunk z >= 1 writes slab z - 1 of `part`, each slab shaped like the gradient slab itself (element at the
    // same offset from its base); the Adam launch adds them in chunk order (elementwise.hip adam_parts_kernel)
    int to_slabs;
};

// TAIL4 (gemm_n100.hip): rows 96..99 of the 100-wide dimension on v_mfma_f32_4x4x1_16B_f32 (8 cycles) instead of a seventh
// 16-row tile that is three quarters padding (32 cycles); bit 23 of ganffn_debug_set_ffn_mode restores the padded tile
template <bool TAIL4>
__global__ __launch_bounds__(256) void tn100_kernel(W100Group grp) {
    constexpr int UT = WBK * LDU, STAGE = UT + WBK * LDV;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 4];
    constexpr int DUMP = 2 * STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    // XCD-aware order (gemm.hip gemm_tn_grouped_kernel): the workgroups of one XCD get a contiguous range of the logical list
    int pi = 0;
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xl = blockIdx.x & 7;
    const int b = (xl < r8 ? xl * (q8 + 1) : r8 * (q8 + 1) + (xl - r8) * q8) + (blockIdx.x >> 3);
#pragma unroll 1
    for (int i = 1; i < grp.n; ++i)
        if (b >= grp.p[i].block0) pi = i;
    const W100Problem& q = grp.p[pi];
    const int local = b - q.block0, z = local / q.ntiles, tile = local - z * q.ntiles;
    const int n0 = tile * WBN;
    const int kbeg = min(z * q.kchunk, q.K), kend = min(q.K, kbeg + q.kchunk);
    const int nt = (kend - kbeg + WBK - 1) / WBK;
    const float* const Ug = q.U;
    const float* const Vg = q.V;
    const int ldu = q.ldu, ldv = q.ldv, Nn = q.Nn;

    // loaders (named scalars: hipcc keeps small arrays that live across the K loop in scratch).  U tile: 32 token rows x 25
    // float4 = 800, 4 per thread (the last partly surplus -> dump slot); V tile: 32 rows x 16 float4 = 512, 2 per thread.
#define GF_W_UIDX(J)                                                                                  \
    const int iu##J = tid + 256 * J;                                                                  \
    const bool oku##J = iu##J < WBK * 25;                                                             \
    const int icu##J = min(iu##J, WBK * 25 - 1), ru##J = icu##J / 25, cu##J = (icu##J - ru##J * 25) << 2; \
    const uint32_t offu##J = (uint32_t)cu##J;                                                         \
    const int ldsu0_##J = oku##J ? ru##J * LDU + cu##J : DUMP, ldsu1_##J = oku##J ? STAGE + ru##J * LDU + cu##J : DUMP;
    GF_W_UIDX(0) GF_W_UIDX(1) GF_W_UIDX(2) GF_W_UIDX(3)
#define GF_W_VIDX(J)                                                                                  \
    const int iv##J = tid + 256 * J, rv##J = iv##J >> 4, cv##J = (iv##J & 15) << 2;                   \
    const uint32_t offv##J = (uint32_t)max(min(n0 + cv##J, Nn - 4), 0);                               \
    const int ldsv##J = UT + rv##J * LDV + cv##J;
    GF_W_VIDX(0) GF_W_VIDX(1)
    float4 qu0, qu1, qu2, qu3, qv0, qv1;
    // Global loads.  A FULL tile (all 32 token rows below kend — every tile but possibly the last) is read through a uniform
    // base pointer (scalar ALU) + loop-invariant 32-bit lane offsets: no vector ALU work per tile.  fp32 MFMAs execute on
    // the SIMD's vector ALU, so every VALU instruction in this loop is paid for in MFMA time (round 5 counters,
    // profiles/r05_wgrad_pmc.json: 3.05 VALU instructions per MFMA, MFMA pipe 57 % busy, before this form).  Rows beyond kend
    // (the tail tile only) are clamped onto the last valid token and zeroed when they go to LDS (wave-uniform tail test).
    const uint32_t lou0 = (uint32_t)ru0 * (uint32_t)ldu + offu0, lou1 = (uint32_t)ru1 * (uint32_t)ldu + offu1,
                   lou2 = (uint32_t)ru2 * (uint32_t)ldu + offu2, lou3 = (uint32_t)ru3 * (uint32_t)ldu + offu3;
    const uint32_t lov0 = (uint32_t)rv0 * (uint32_t)ldv + offv0, lov1 = (uint32_t)rv1 * (uint32_t)ldv + offv1;
    // (buffer descriptors: the tile's token offset goes into the SCALAR offset of the load — no per-lane 64-bit pointer to advance;
    //  unbounded: a full tile lies inside the operands by construction; operands of 4 GiB or more are refused by the launcher)
    const __amdgpu_buffer_rsrc_t rsU = buf_rsrc(Ug, 0xFFFFFFFFu);
    const __amdgpu_buffer_rsrc_t rsV = buf_rsrc(Vg, 0xFFFFFFFFu);
#define GF_W_GLOAD(TT)                                                                                \
    {                                                                                                 \
        const int k0 = kbeg + min((TT), nt - 1) * WBK, kl = kend - 1;                                 \
        if (k0 + WBK <= kend) {                             /* wave-uniform */                          \
            const uint32_t su_ = (uint32_t)k0 * (uint32_t)ldu * 4u, sv_ = (uint32_t)k0 * (uint32_t)ldv * 4u;   /* scalar */ \
            qu0 = buf_load_f4(rsU, 4u * lou0, su_);                                                     \
            qu1 = buf_load_f4(rsU, 4u * lou1, su_);                                                     \
            qu2 = buf_load_f4(rsU, 4u * lou2, su_);                                                     \
            qu3 = buf_load_f4(rsU, 4u * lou3, su_);                                                     \
            qv0 = buf_load_f4(rsV, 4u * lov0, sv_);                                                     \
            qv1 = buf_load_f4(rsV, 4u * lov1, sv_);                                                     \
        } else {                                                                                      \
            qu0 = *reinterpret_cast<const float4*>(Ug + (size_t)min(k0 + ru0, kl) * ldu + offu0);     \
            qu1 = *reinterpret_cast<const float4*>(Ug + (size_t)min(k0 + ru1, kl) * ldu + offu1);     \
            qu2 = *reinterpret_cast<const float4*>(Ug + (size_t)min(k0 + ru2, kl) * ldu + offu2);     \
            qu3 = *reinterpret_cast<const float4*>(Ug + (size_t)min(k0 + ru3, kl) * ldu + offu3);     \
            qv0 = *reinterpret_cast<const float4*>(Vg + (size_t)min(k0 + rv0, kl) * ldv + offv0);     \
            qv1 = *reinterpret_cast<const float4*>(Vg + (size_t)min(k0 + rv1, kl) * ldv + offv1);     \
        }                                                                                             \
    }
#define GF_W_MASK(Q, ROW) { const float f_ = (ROW) < kv ? 1.f : 0.f; Q.x *= f_; Q.y *= f_; Q.z *= f_; Q.w *= f_; }
#define GF_W_SSTORE(BUF, TT)                                                                          \
    {                                                                                                 \
        const int kv = kend - (kbeg + (TT) * WBK);          /* valid token rows of this tile */         \
        if (kv < WBK) { GF_W_MASK(qu0, ru0) GF_W_MASK(qu1, ru1) GF_W_MASK(qu2, ru2) GF_W_MASK(qu3, ru3) GF_W_MASK(qv0, rv0) GF_W_MASK(qv1, rv1) } \
        float* const sdst = smem + (BUF) * STAGE;                                                     \
        *reinterpret_cast<float4*>(smem + ((BUF) ? ldsu1_0 : ldsu0_0)) = qu0;                         \
        *reinterpret_cast<float4*>(smem + ((BUF) ? ldsu1_1 : ldsu0_1)) = qu1;                         \
        *reinterpret_cast<float4*>(smem + ((BUF) ? ldsu1_2 : ldsu0_2)) = qu2;                         \
        *reinterpret_cast<float4*>(smem + ((BUF) ? ldsu1_3 : ldsu0_3)) = qu3;                         \
        *reinterpret_cast<float4*>(sdst + ldsv0) = qv0;                                               \
        *reinterpret_cast<float4*>(sdst + ldsv1) = qv1;                                               \
    }

    // The accumulators are updated IN PLACE by MFMAs written as inline assembly: with the builtin, hipcc let the destination
    // of an MFMA differ from its accumulator input and restored the assignment with 88 v_accvgpr_read / write / mov per K
    // tile (for 56 MFMAs).  Hazards the compiler no longer sees: an accumulator is touched again 7 MFMAs (>= 200 cycles)
    // later — far beyond any MFMA -> MFMA wait-state requirement; the A / B operands come straight from ds_read (s_waitcnt is
    // still inserted for inline-assembly operands), never from a VALU instruction right before; the epilogue's first VALU
    // read of an accumulator comes after the explicit s_nop block below.
    floatx4 acc[WT7];
#pragma unroll
    for (int m = 0; m < WT7; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of dY over the tokens.  side 0: dY = U (its 100 columns): wave 0 of the tile-0 workgroup
    // sums the A operand values it reads; side 1: dY = V: every wave sums the B operand values of its 16 columns.
    // (wave-uniform by construction; readfirstlane tells the compiler, which otherwise runs the sums of the one wave that
    //  owns them under an exec mask in EVERY wave: 32 dead VALU instructions per K tile)
    const bool cs_u = __builtin_amdgcn_readfirstlane((int)(q.colsum != nullptr && q.side == 0 && tile == 0 && wave == 0)) != 0;
    const bool cs_v = q.colsum != nullptr && q.side == 1;
    float csu[WT7], csv = 0.f;
#pragma unroll
    for (int m = 0; m < WT7; ++m) csu[m] = 0.f;

    // columns 100 .. 111 of the U tile are never written by the loader: zero them once (both stages)
    for (int i = tid; i < 2 * WBK * 12; i += 256) {
        const int st = i / (WBK * 12), r = (i / 12) % WBK, cc = i % 12;
        smem[st * STAGE + r * LDU + WE + cc] = 0.f;
    }
    if (nt > 0) {
        GF_W_GLOAD(0)
        GF_W_SSTORE(0, 0)
    }
    __syncthreads();

    // 8 k-steps of 4 tokens per tile: at the j-th MFMA of a half, lane group g takes token 16 half + 4 g + j (rows of the lane
    // groups 4 apart: conflict-free ds_read_b32 with the 116 / 68 row strides).  The stage is a compile-time constant of the
    // step (the loop is unrolled by two): every LDS address is one loop-invariant VGPR + an immediate offset.
    const float* const sv_lane = smem + UT + wave * 16 + c + 4 * g * LDV;
    const float* const su_lane = smem + c + 4 * g * LDU;
    const float* const s4_lane = smem + 96 + (c & 3) + 4 * g * LDU;
    // LDS reads run one group of 2 k-steps (16 dwords per lane) AHEAD of the MFMAs: the reads of group i + 1 are issued before
    // the 14 MFMAs of group i, so only the first group of a tile waits for LDS (k order unchanged: 0 1 | 2 3 | 16 17 | 18 19 + 4 g)
#define GF_W_LOADG(A, B, BUF, GI)                                                                     \
    {                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
            const int kr = 16 * ((GI) >> 1) + 2 * ((GI) & 1) + j;                                     \
            B[j] = sv_lane[(BUF) * STAGE + kr * LDV];                                                 \
            _Pragma("unroll") for (int m = 0; m < WT7; ++m)                                           \
                A[j][m] = (TAIL4 && m == WT7 - 1) ? s4_lane[(BUF) * STAGE + kr * LDU] : su_lane[(BUF) * STAGE + kr * LDU + 16 * m]; \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define GF_W_MFMAG(A, B)                                                                              \
    {                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
            _Pragma("unroll") for (int m = 0; m < WT7; ++m) {                                         \
                if (TAIL4 && m == WT7 - 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(A[j][m]), "v"(B[j])); \
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(A[j][m]), "v"(B[j])); \
            }                                                                                         \
        if (cs_v) { csv += B[0]; csv += B[1]; }                                                       \
        if (cs_u) {                                                                                   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                _Pragma("unroll") for (int m = 0; m < WT7; ++m) csu[m] += A[j][m];                    \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define GF_W_COMPUTE(BUF)                                                                             \
    {                                                                                                 \
        float a0[2][WT7], b0[2], a1[2][WT7], b1[2];                                                   \
        GF_W_LOADG(a0, b0, BUF, 0)                                                                    \
        GF_W_LOADG(a1, b1, BUF, 1)                                                                    \
        GF_W_MFMAG(a0, b0)                                                                            \
        GF_W_LOADG(a0, b0, BUF, 2)                                                                    \
        GF_W_MFMAG(a1, b1)                                                                            \
        GF_W_LOADG(a1, b1, BUF, 3)                                                                    \
        GF_W_MFMAG(a0, b0)                                                                            \
        GF_W_MFMAG(a1, b1)                                                                            \
    }
#define GF_W_STEP(BUF, TT)                                                                            \
    {                                                                                                 \
        GF_W_GLOAD((TT) + 1)                                                                          \
        __builtin_amdgcn_sched_barrier(0);       /* the next tile's loads stay ahead of this tile's MFMAs */ \
        GF_W_COMPUTE(BUF)                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if ((TT) + 1 < nt) GF_W_SSTORE((BUF) ^ 1, (TT) + 1)                                           \
        __syncthreads();                                                                              \
    }
    for (int t = 0; t < nt; t += 2) {
        GF_W_STEP(0, t)
        if (t + 1 < nt) GF_W_STEP(1, t + 1)
    }
    // (inline-assembly MFMAs: the wait states before a VALU / store instruction reads an accumulator are ours to provide)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#undef GF_W_STEP
#undef GF_W_COMPUTE
#undef GF_W_LOADG
#undef GF_W_MFMAG
#undef GF_W_GLOAD
#undef GF_W_SSTORE
#undef GF_W_MASK
#undef GF_W_UIDX
#undef GF_W_VIDX

    // ---------------- epilogue: lane (c, g) holds m = 16 mt + 4 g + r (the 100-wide dim), n = n0 + 16 wave + c ----------------
    if constexpr (TAIL4) {
        // rows 96..99: register r of every lane group holds the partial sum over ITS tokens of (row 96 + r, column c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[WT7 - 1][r];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[WT7 - 1][r] = v;
        }
    }
    const int n = n0 + wave * 16 + c;
    const bool nok = n < Nn;
    if (grp.to_slabs) {
        // plain stores, every chunk: chunk 0 into the gradient itself, chunk z into gradient-shaped slab z - 1
        float* const Cb = z == 0 ? q.C : grp.part + (size_t)(z - 1) * grp.part_stride + q.c_off;
        float* const Sb = z == 0 ? q.colsum : grp.part + (size_t)(z - 1) * grp.part_stride + q.cs_off;
        if (q.side == 0) {
#pragma unroll
            for (int m = 0; m < WT7; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = 16 * m + 4 * g + r;
                    if (mm < WE && nok) Cb[(size_t)mm * q.ldc + n] = acc[m][r];
                }
        } else if (nok) {
#pragma unroll
            for (int m = 0; m < WT7; ++m) {
                const int mm = 16 * m + 4 * g;
                if (mm < WE) *reinterpret_cast<float4*>(Cb + (size_t)n * q.ldc + mm) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
            }
        }
        if (cs_v) {
            csv += __shfl_xor(csv, 16, 64);
            csv += __shfl_xor(csv, 32, 64);
            if (g == 0 && nok) Sb[n] = csv;
        }
        if (cs_u) {
#pragma unroll
            for (int m = 0; m < WT7; ++m) {
                float v = csu[m];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int mm = 16 * m + c;
                if (g == 0 && mm < WE) Sb[mm] = v;
            }
        }
        return;
    }
    float* const slab = grp.splits > 1 ? grp.part + (size_t)z * grp.part_stride + q.part_off : nullptr;
    if (q.side == 0) {
        // gradient [100 x Nn]: element (m, n)
#pragma unroll
        for (int m = 0; m < WT7; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = 16 * m + 4 * g + r;
                if (mm < WE && nok) {
                    if (slab) slab[(size_t)mm * q.N + n] = acc[m][r];
                    else q.C[(size_t)mm * q.ldc + n] += acc[m][r];
                }
            }
    } else {
        // gradient [Nn x 100]: element (n, m): 4 consecutive m = one 16-byte access
        if (nok) {
#pragma unroll
            for (int m = 0; m < WT7; ++m) {
                const int mm = 16 * m + 4 * g;
                if (mm < WE) {
                    if (slab) {
                        *reinterpret_cast<float4*>(slab + (size_t)n * q.N + mm) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
                    } else {
                        float* dst = q.C + (size_t)n * q.ldc + mm;
                        const float4 o = *reinterpret_cast<const float4*>(dst);
                        *reinterpret_cast<float4*>(dst) = make_float4(o.x + acc[m][0], o.y + acc[m][1], o.z + acc[m][2], o.w + acc[m][3]);
                    }
                }
            }
        }
    }
    if (cs_v) {            // wave-uniform
        csv += __shfl_xor(csv, 16, 64);
        csv += __shfl_xor(csv, 32, 64);
        if (g == 0 && nok) {
            if (slab) slab[(size_t)q.M * q.N + n] = csv;
            else q.colsum[n] += csv;
        }
    }
    if (cs_u) {
#pragma unroll
        for (int m = 0; m < WT7; ++m) {
            float v = csu[m];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int mm = 16 * m + c;
            if (g == 0 && mm < WE) {
                if (slab) slab[(size_t)q.M * q.N + mm] = v;
                else q.colsum[mm] += v;
            }
        }
    }
    if (slab == nullptr || grp.counters == nullptr) return;

    // ---------------- the last chunk of this tile to arrive adds the tile's slabs to the gradient, in chunk order ----------------
    // release: this workgroup's slab stores are visible device-wide before its ticket is; acquire on the other side.  The
    // ticket decides only WHO adds — every slab is read back from memory and summed z = 0, 1, ... exactly as
    // tn100_reduce_kernel does, so the result does not depend on the arrival order (and equals round 3's bits).
    // (cdna_hip_programming.md, in-launch split-K reduction: every wave drains its stores, the workgroup meets, ONE lane
    // releases at agent scope and draws the ticket; the last arriver's one lane acquires, then the workgroup reads the slabs
    // with plain loads)
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(&grp.counters[q.ctr0 + tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == grp.splits - 1 ? 1 : 0;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!s_last) return;
    const float* const part = grp.part + q.part_off;
    const int ns = grp.splits;
    if (q.side == 0) {
#pragma unroll
        for (int m = 0; m < WT7; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = 16 * m + 4 * g + r;
                if (mm < WE && nok) {
                    const size_t e = (size_t)mm * q.N + n;
                    float sacc = part[e];
                    for (int zz = 1; zz < ns; ++zz) sacc += part[(size_t)zz * grp.part_stride + e];
                    q.C[(size_t)mm * q.ldc + n] += sacc;
                }
            }
    } else if (nok) {
#pragma unroll
        for (int m = 0; m < WT7; ++m) {
            const int mm = 16 * m + 4 * g;
            if (mm < WE) {
                const size_t e = (size_t)n * q.N + mm;
                float4 sacc = *reinterpret_cast<const float4*>(part + e);
                for (int zz = 1; zz < ns; ++zz) {
                    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)zz * grp.part_stride + e);
                    sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
                }
                float* dst = q.C + (size_t)n * q.ldc + mm;
                const float4 o = *reinterpret_cast<const float4*>(dst);
                *reinterpret_cast<float4*>(dst) = make_float4(o.x + sacc.x, o.y + sacc.y, o.z + sacc.z, o.w + sacc.w);
            }
        }
    }
    const size_t nC = (size_t)q.M * q.N;
    if (cs_v && g == 0 && nok) {
        float sacc = part[nC + n];
        for (int zz = 1; zz < ns; ++zz) sacc += part[(size_t)zz * grp.part_stride + nC + n];
        q.colsum[n] += sacc;
    }
    if (cs_u && g == 0) {
#pragma unroll
        for (int m = 0; m < WT7; ++m) {
            const int mm = 16 * m + c;
            if (mm < WE) {
                float sacc = part[nC + mm];
                for (int zz = 1; zz < ns; ++zz) sacc += part[(size_t)zz * grp.part_stride + nC + mm];
                q.colsum[mm] += sacc;
            }
        }
    }
}

// C_i += sum_z part[z][i], colsum_i += sum_z part[z][M N + i], slabs in chunk order; blockIdx.y = problem
__global__ __launch_bounds__(256) void tn100_reduce_kernel(W100Group grp) {
    const W100Problem& q = grp.p[blockIdx.y];
    const float* part = grp.part + q.part_off;
    const long nC = (long)q.M * q.N;
    for (long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i4 < nC; i4 += (long)gridDim.x * 1024) {
        float4 v[WMAXSPLIT];                                    // all slabs' loads in flight together
#pragma unroll
        for (int z = 0; z < WMAXSPLIT; ++z)
            v[z] = *reinterpret_cast<const float4*>(part + (size_t)min(z, grp.splits - 1) * grp.part_stride + i4);
        const long row = i4 / q.N, col = i4 - row * q.N;      // N % 4 == 0: a float4 stays inside one row
        float* dst = q.C + row * q.ldc + col;
        const float4 old = *reinterpret_cast<const float4*>(dst);
        float4 s = v[0];
#pragma unroll
        for (int z = 1; z < WMAXSPLIT; ++z)
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

}  // namespace

// lab knobs of the mode word (common.h `Mode`): bit 23 = the padded seventh tile, bits 16..19 force the token-chunk count,
// bit 4 = add the partial slabs in the last-arriving workgroup of a tile (measured slower)
constexpr long TN100_COUNTERS = 4096;      // ints reserved at the end of the partial-slab workspace for the arrival tickets

// every problem has a 100-wide dimension, 16-byte aligned dense operands
bool tn100_supported(const TnDesc* d, int n) {
    if (n < 1 || n > WMAXP) return false;
    for (int i = 0; i < n; ++i) {
        if (d[i].M != WE && d[i].N != WE) return false;
        if ((d[i].M & 3) || (d[i].N & 3) || (d[i].lda & 3) || (d[i].ldb & 3) || (d[i].ldc & 3)) return false;
        if (!aligned16(d[i].At) || !aligned16(d[i].B) || !aligned16(d[i].C)) return false;
    }
    return true;
}

long tn100_part_floats(const TnDesc* d, int n) {
    long per = 0;
    for (int i = 0; i < n; ++i) per += (((long)d[i].M * d[i].N + d[i].M) + 3) & ~3L;
    return per;
}

// slabs (optional): the caller wants the weight gradients UNREDUCED (TnSlabs, common.h): no reduce launch, chunk 0 overwrites the
// gradient slab, chunks 1.. go to gradient-shaped slabs at the head of part_ws
int launch_gemm_tn100_grouped(const TnDesc* d, int n, hipStream_t st, float* part_ws, long part_floats, TnSlabs* slabs) {
    GF_CHECK_ARG(tn100_supported(d, n), "gemm_tn100_grouped: unsupported group");
    for (int i = 0; i < n; ++i)
        GF_CHECK_ARG((unsigned long long)d[i].K * (unsigned long long)(d[i].lda > d[i].ldb ? d[i].lda : d[i].ldb) * 4ull < (1ull << 32),
                     "gemm_tn100_grouped: an operand of 4 GiB or more is not supported");
    const Mode md = mode();
    W100Group grp;
    grp.n = n;
    grp.to_slabs = 0;
    long tiles = 0, per_split = tn100_part_floats(d, n);
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        const int Nn = d[i].M == WE ? d[i].N : d[i].M;
        tiles += (Nn + WBN - 1) / WBN;
        kmax = d[i].K > kmax ? d[i].K : kmax;
    }
    // chunks of the token range: aim at ~6.6 workgroups per CU (1700 for a whole 8-layer pass = 568 tiles x 3: the CUs then
    // finish within 5 % of each other), at least 256 tokens per workgroup, within the workspace
    int splits = 1;
    int* counters = nullptr;
    if (part_ws != nullptr && aligned16(part_ws) && md.tn100_in_kernel_sum() && tiles <= TN100_COUNTERS && part_floats > 2 * TN100_COUNTERS) {
        part_floats -= TN100_COUNTERS;
        counters = reinterpret_cast<int*>(part_ws + part_floats);
    }
    if (part_ws != nullptr && aligned16(part_ws)) {
        splits = (int)((1700 + tiles - 1) / tiles);
        if (splits > WMAXSPLIT) splits = WMAXSPLIT;
        if (splits > kmax / 256) splits = kmax / 256;
        if (md.tn100_force_splits() > 0) splits = md.tn100_force_splits() < WMAXSPLIT ? md.tn100_force_splits() : WMAXSPLIT;
        if (slabs != nullptr) {
            if ((long)(splits - 1) * slabs->range_floats > part_floats) splits = 1 + (int)(part_floats / slabs->range_floats);
        } else if ((long)splits * per_split > part_floats) splits = (int)(part_floats / per_split);
        if (splits < 2) splits = 1;
    }
    grp.splits = splits;
    grp.part = splits > 1 ? part_ws : nullptr;
    grp.part_stride = per_split;
    grp.counters = splits > 1 ? counters : nullptr;
    if (slabs != nullptr) {
        GF_CHECK_ARG(slabs->grad_base && (slabs->range_floats & 3) == 0, "gemm_tn100_grouped: bad gradient-slab description");
        for (int i = 0; i < n; ++i) {
            const long co = d[i].C - slabs->grad_base, so = d[i].colsum ? d[i].colsum - slabs->grad_base : 0;
            GF_CHECK_ARG(co >= 0 && co + (long)d[i].M * d[i].N <= slabs->range_floats && so >= 0 && so + d[i].M <= slabs->range_floats &&
                             d[i].ldc == d[i].N, "gemm_tn100_grouped: problem %d does not lie densely inside the gradient slab", i);
        }
        grp.to_slabs = 1;
        grp.part_stride = slabs->range_floats;
        grp.counters = nullptr;
        slabs->n_parts = splits;
        slabs->part = splits > 1 ? part_ws : nullptr;
        slabs->part_stride = slabs->range_floats;
    }
    int total = 0, ctr = 0;
    long off = 0;
    for (int i = 0; i < n; ++i) {
        W100Problem& q = grp.p[i];
        const bool side0 = d[i].M == WE;                 // (a 100 x 100 gradient: dY is the U operand)
        q.U = side0 ? d[i].At : d[i].B;
        q.V = side0 ? d[i].B : d[i].At;
        q.ldu = side0 ? d[i].lda : d[i].ldb;
        q.ldv = side0 ? d[i].ldb : d[i].lda;
        q.Nn = side0 ? d[i].N : d[i].M;
        q.side = side0 ? 0 : 1;
        q.C = d[i].C; q.colsum = d[i].colsum; q.ldc = d[i].ldc; q.M = d[i].M; q.N = d[i].N; q.K = d[i].K;
        q.kchunk = splits > 1 ? (int)((((long)d[i].K + splits - 1) / splits + WBK - 1) / WBK * WBK) : d[i].K;
        q.ntiles = (q.Nn + WBN - 1) / WBN;
        q.block0 = total;
        q.part_off = off;
        q.c_off = slabs ? d[i].C - slabs->grad_base : 0;
        q.cs_off = (slabs && d[i].colsum) ? d[i].colsum - slabs->grad_base : 0;
        q.ctr0 = ctr;
        ctr += q.ntiles;
        off += (((long)d[i].M * d[i].N + d[i].M) + 3) & ~3L;
        total += q.ntiles * splits;
    }
    if (grp.counters != nullptr) GF_HIP(hipMemsetAsync(grp.counters, 0, (size_t)ctr * sizeof(int), st));
    if (md.n100_pad7()) hipLaunchKernelGGL(tn100_kernel<false>, dim3(total), dim3(256), 0, st, grp);
    else hipLaunchKernelGGL(tn100_kernel<true>, dim3(total), dim3(256), 0, st, grp);
    GF_LAUNCH_CHECK();
    if (splits > 1 && grp.counters == nullptr && !grp.to_slabs) {
        hipLaunchKernelGGL(tn100_reduce_kernel, dim3(64, n), dim3(256), 0, st, grp);
        GF_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace ganffn
