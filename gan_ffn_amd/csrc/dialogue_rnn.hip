// dialogue_rnn.hip — the DialogueRNN recurrence of BASELINE.json configs[4] (GAN_FFN_DialogueRNN head) on gfx950.
//
// Replaces DialogueRNNCell.forward / DialogueRNN.forward (/root/reference/model.py:828-972) in the configuration
// train_IEMOCAP_DialogueRNN.py runs (:586 --attention general, :595 listener off; dims :634-642: D_m = 100,
// D_g = D_p = 500, D_e = 100): per utterance step t and dialogue b
//     g_t  = drop(GRU_g([U_t, q_{t-1}[spk]], g_{t-1}))                    global state
//     c_t  = sum_j softmax_j(<W_a U_t, g_j>) g_j,  j < t                   "general" attention over the global history
//     q_t[spk] = m ? drop(GRU_p([U_t, c_t], q_{t-1}[spk])) : q_{t-1}[spk]  speaker's party state (listener unchanged)
//     e_t  = drop(GRU_e(q_t[spk], e_{t-1}))                                emotion state
// (m = qmask[t,b,spk]: 0 on padded steps, where the reference's blend keeps both party states.)
//
// Why not one kernel per dialogue: a step needs every cell's full weight matrix (12.7 MB per direction); one workgroup
// per dialogue would stream that through ONE CU 94 times.  Instead the dialogues are the N axis (<= 32 per tile) of
// skinny matrix products that spread a cell's weight rows over ~100-200 workgroups, and the S steps are a chain of small
// launches (2 per step forward, 2 backward, both directions of BiModel in the same launches; independent pieces of a step
// share a launch: within step t the global and the party cell do not depend on each other — the party cell reads the
// context over g_0 .. g_{t-1}, not g_t — so their four products are one launch and their gate kernels another, with the
// context attention of the next step behind the global cell's gate math in the workgroup that owns the dialogue; round 2
// ran the cells one after the other, 4 + 4 launches per step):
//   * everything that depends on U alone is hoisted out of the recurrence into three ordinary GEMMs over all steps
//     (x-parts of the g / p cells incl. b_ih, and the attention query W_a U_t);
//   * skinny_nt / skinny_nn: C[B x N] = A[B x K] W^T resp. A W on v_mfma_f32_16x16x4_f32 (exact fp32): a workgroup owns
//     16 weight rows (columns) and all dialogues, its waves split K and are summed through LDS in a fixed order;
//   * gate kernels (elementwise GRU math + Philox dropout + party select/update), attention kernels (one workgroup
//     per dialogue, history in L2);
//   * the emotion cell feeds nothing back into the recurrence: its input product for all steps is one GEMM after the
//     main loop and its own recurrence (independent per dialogue, 300 x 100 weight) runs as one persistent workgroup per
//     (dialogue, direction) — drnn_echain_fwd / bwd_kernel;
//   * backward mirrors it step by step in reverse; all weight gradients are deferred: the per-step gate gradients are
//     kept and ONE grouped TN GEMM launch (gemm.hip, owner-accumulated, no atomics) computes the 9 products at the end.
// Everything is deterministic (no atomics).  Dropout follows the Philox contract of common.h with rows t*B + b.
#include "common.h"

namespace ganffn {

typedef float floatx4 __attribute__((ext_vector_type(4)));

enum : uint32_t { SITE_DRNN_G = 8, SITE_DRNN_P = 9, SITE_DRNN_E = 10 };   // + 4 for the second direction

// ------------------------------------------------------------------------------------------
// skinny products: M <= 32 rows of A (dialogues) against a weight matrix
// ------------------------------------------------------------------------------------------
// (SkinnyProb / SkinnyGroup: common.h — lstm.hip runs its recurrent products through the same kernels)

// NT: C[b][n] = sum_k A[b][k] W[n][k] (+ Cin[b][n] + bias[n]).  Workgroup = 16 weight rows x 32 dialogues, NW waves split K
// (4 for the K = 500 products of the forward step, 12 for the K = 1500 ones of the backward step: 125 -> 128 k each).
// MFMA tile D[m = weight row][n' = dialogue]: A-operand = W rows, B-operand = A rows; both are float4 loads along k
// (k block of 16 per 4 MFMAs: step i contracts k = kb + 4g + i, identically on both operands).
// A launch is one link of a serial chain, so what counts is its latency: every load of a chunk of SK_NT k-blocks (a
// wave's whole K range in both uses) is issued before the first MFMA — one memory round trip (10.9 -> 9.9 us in-step).
// (Round 4: two weight tiles per workgroup — 4 loads per 4 MFMA groups instead of 3 per 2, half the workgroups, a third
// less L2 traffic — measured SLOWER: configuration 5 15.07 against 14.80 ms, three interleaved runs each.  The launch is not
// L2-bandwidth-sized; fewer, fatter waves lengthen the one round trip it consists of.)
constexpr int SK_NT = 8;
template <int NW>
__global__ __launch_bounds__(64 * NW) void skinny_nt_kernel(SkinnyGroup grp) {
    __shared__ __attribute__((aligned(16))) float red[NW][2][4][64];
    const SkinnyProb& q = grp.p[blockIdx.z];
    const int n0 = blockIdx.x * 16;
    if (n0 >= q.N) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int KQ = (((q.K + NW - 1) / NW) + 15) / 16 * 16;     // k range per wave, multiple of 16
    const int kbeg = w * KQ, kend = min(q.K, kbeg + KQ);
    const float* wrow = q.W + (size_t)min(n0 + c, q.N - 1) * q.ldw;
    const float* a0 = q.A + (size_t)min(c, q.M - 1) * q.lda;
    const float* a1 = q.A + (size_t)min(16 + c, q.M - 1) * q.lda;
    // the epilogue's addends (waves 0 and 1 write the tile) are fetched with the operands, not after the reduction
    float e_cin[4] = {0.f, 0.f, 0.f, 0.f}, e_cin2[4] = {0.f, 0.f, 0.f, 0.f}, e_bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 128) {
        const int eb = min(16 * (int)(threadIdx.x >> 6) + c, q.M - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int en = min(n0 + 4 * g + r, q.N - 1);
            if (q.Cin) e_cin[r] = q.Cin[(size_t)eb * q.ldcin + en];
            if (q.Cin2) e_cin2[r] = q.Cin2[(size_t)eb * q.ldc + en];
            if (q.bias) e_bias[r] = q.bias[en];
        }
    }
    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int kb = kbeg; kb < kend; kb += 16 * SK_NT) {
        float4 wv[SK_NT], x0[SK_NT], x1[SK_NT];
        float f[SK_NT];
#pragma unroll
        for (int u = 0; u < SK_NT; ++u) {
            const int k = kb + 16 * u + 4 * g;             // K % 4 == 0: a float4 never straddles the end
            f[u] = k < kend ? 1.f : 0.f;
            const int kc = max(min(k, q.K - 4), 0);
            wv[u] = *reinterpret_cast<const float4*>(wrow + kc);
            x0[u] = *reinterpret_cast<const float4*>(a0 + kc);
            x1[u] = *reinterpret_cast<const float4*>(a1 + kc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SK_NT; ++u) {
            const float wx = wv[u].x * f[u], wy = wv[u].y * f[u], wz = wv[u].z * f[u], ww = wv[u].w * f[u];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wx, x0[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wx, x1[u].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wy, x0[u].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wy, x1[u].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz, x0[u].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz, x1[u].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ww, x0[u].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ww, x1[u].w, acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[w][0][r][lane] = acc0[r];
        red[w][1][r][lane] = acc1[r];
    }
    __syncthreads();
    // thread -> (dialogue tile mt, lane): sums the waves in order; lane holds C[b = 16 mt + c][n0 + 4g .. 4g+3]
    if (threadIdx.x < 128) {
        const int mt = threadIdx.x >> 6;
        const int b = 16 * mt + c, n = n0 + 4 * g;
        if (b < q.M && n < q.N) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = red[0][mt][r][lane];
#pragma unroll
                for (int ww = 1; ww < NW; ++ww) v[r] += red[ww][mt][r][lane];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < q.N) {
                    float o = v[r];
                    if (q.Cin) o += e_cin[r];
                    if (q.Cin2) o += e_cin2[r];
                    if (q.bias) o += e_bias[r];
                    q.C[(size_t)b * q.ldc + n + r] = o;
                }
            }
        }
    }
}

// out[c][r] = in[r][c] for up to 8 matrices [rows x cols] (leading dimension ld_in) -> [cols x rows]: the recurrent
// weights in the orientation the backward step's products read row-wise (once per backward call)
struct TrGroup { const float* in[8]; float* out[8]; int ld_in[8]; int rows, cols; };
__global__ __launch_bounds__(256) void drnn_transpose_kernel(TrGroup t) {
    __shared__ float tile[32][33];
    const int z = blockIdx.z, c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < t.rows && c < t.cols) ? t.in[z][(size_t)r * t.ld_in[z] + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < t.rows && c < t.cols) t.out[z][(size_t)c * t.rows + r] = tile[tx][ty + 8 * i];
    }
}

// NN: C[b][n] = sum_k A[b][k] W[k][n] (+ Cin).  Workgroup = 16 output columns x 32 dialogues, 8 waves split K.
// MFMA tile D[m = dialogue][n' = column]: A-operand = A rows (float4 along k), B-operand = W[k][n0 + n'] (one dword per
// step: 16 consecutive columns of a row = 64 contiguous bytes).
__global__ __launch_bounds__(512) void skinny_nn_kernel(SkinnyGroup grp) {
    __shared__ __attribute__((aligned(16))) float red[8][2][4][64];
    const SkinnyProb& q = grp.p[blockIdx.z];
    const int n0 = blockIdx.x * 16;
    if (n0 >= q.N) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int KQ = (((q.K + 7) / 8) + 15) / 16 * 16;
    const int kbeg = w * KQ, kend = min(q.K, kbeg + KQ);
    const float* a0 = q.A + (size_t)min(c, q.M - 1) * q.lda;
    const float* a1 = q.A + (size_t)min(16 + c, q.M - 1) * q.lda;
    const float* wcol = q.W + min(n0 + c, q.N - 1);
    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int kb = kbeg; kb < kend; kb += 16) {
        const int k = kb + 4 * g;
        const float f = k < kend ? 1.f : 0.f;
        const int kc = max(min(k, q.K - 4), 0);
        const float4 x0 = *reinterpret_cast<const float4*>(a0 + kc);
        const float4 x1 = *reinterpret_cast<const float4*>(a1 + kc);
        float wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = wcol[(size_t)(kc + i) * q.ldw] * f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, wv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, wv[0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, wv[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, wv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, wv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, wv[2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, wv[3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, wv[3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[w][0][r][lane] = acc0[r];
        red[w][1][r][lane] = acc1[r];
    }
    __syncthreads();
    // accumulator register r of tile mt at lane (c, g) = C[b = 16 mt + 4g + r][n0 + c]
    if (threadIdx.x < 128) {
        const int mt = threadIdx.x >> 6;
        const int n = n0 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = 16 * mt + 4 * g + r;
            if (b < q.M && n < q.N) {
                float o = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) o += red[ww][mt][r][lane];
                if (q.Cin) o += q.Cin[(size_t)b * q.ldcin + n];
                if (q.Cin2) o += q.Cin2[(size_t)b * q.ldc + n];
                q.C[(size_t)b * q.ldc + n] = o;
            }
        }
    }
}

int launch_skinny(const SkinnyGroup& grp, int nprob, bool nn, hipStream_t st) {
    int maxN = 0, maxK = 0;
    for (int i = 0; i < nprob; ++i) {
        const SkinnyProb& q = grp.p[i];
        GF_CHECK_ARG(q.M >= 1 && q.M <= 32 && (q.K & 3) == 0 && (q.lda & 3) == 0 && aligned16(q.A) && (nn || ((q.ldw & 3) == 0 && aligned16(q.W))),
                     "drnn skinny product: M=%d K=%d lda=%d ldw=%d unsupported", q.M, q.K, q.lda, q.ldw);
        maxN = q.N > maxN ? q.N : maxN;
        maxK = q.K > maxK ? q.K : maxK;
    }
    if (nn) hipLaunchKernelGGL(skinny_nn_kernel, dim3((maxN + 15) / 16, 1, nprob), dim3(512), 0, st, grp);
    else if (maxK > 4 * 16 * SK_NT) hipLaunchKernelGGL(skinny_nt_kernel<12>, dim3((maxN + 15) / 16, 1, nprob), dim3(768), 0, st, grp);
    else hipLaunchKernelGGL(skinny_nt_kernel<4>, dim3((maxN + 15) / 16, 1, nprob), dim3(256), 0, st, grp);
    GF_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// GRU gate math (torch.nn.GRUCell: r, z, n row blocks; n = tanh(i_n + r * h_n); h' = (1 - z) n + z h)
// ------------------------------------------------------------------------------------------
struct GateDir {
    const float* GI; const float* GH;     // [B x 3H] pre-activations (biases included)
    const float* hprev;                   // [B x H]
    float* R; float* Z; float* N; float* HN;   // saved gates of this step [B x H]
    float* hout;                          // g / e cells: state after dropout [B x H]
    // party cell only:
    const int* spk; const int* spk_next;  // [B] speaker of this / the next step (spk_next NULL on the last step)
    const float* mval;                    // [B] 1 on valid steps, 0 on padding
    const float* Qprev; float* Qnext;     // [B x 2 x H] party states before / after this step
    float* QN; float* QSnext;             // [B x H] q_t[spk_t]; q_t[spk_{t+1}] (next step's g / p cell input)
    uint32_t site;
};
struct GateArgs {
    GateDir d[2];
    int B, H, row0;        // row0 = t * B: dropout row of dialogue 0
    float p; int train;
    const uint64_t* rng; uint64_t add;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int PARTY>
__device__ __forceinline__ void gru_gate_fwd_body(const GateArgs& a, const GateDir& d, const int idx) {
    if (idx >= a.B * a.H) return;
    const int b = idx / a.H, u = idx - b * a.H, H3 = 3 * a.H;
    const float* gi = d.GI + (size_t)b * H3;
    const float* gh = d.GH + (size_t)b * H3;
    const float h = d.hprev[idx];
    const float r = sigm(gi[u] + gh[u]);
    const float z = sigm(gi[a.H + u] + gh[a.H + u]);
    const float hn = gh[2 * a.H + u];
    const float n = tanhf(gi[2 * a.H + u] + r * hn);
    float hnew = (1.0f - z) * n + z * h;
    d.R[idx] = r; d.Z[idx] = z; d.N[idx] = n; d.HN[idx] = hn;
    const DropCtx dc = make_drop(a.rng, a.add, d.site, a.p, a.train);
    hnew *= drop_mult1(dc, (uint32_t)(a.row0 + b), (uint32_t)a.H, (uint32_t)u);
    if (!PARTY) {
        d.hout[idx] = hnew;
    } else {
        // q_t[spk] = m ? qs : q_{t-1}[spk]; the other party keeps its state (listener_state False, model.py:888-893)
        const int s = d.spk[b];
        const float m = d.mval[b];
        const float qs = m != 0.f ? hnew : h;
        const float other = d.Qprev[((size_t)b * 2 + (1 - s)) * a.H + u];
        d.Qnext[((size_t)b * 2 + s) * a.H + u] = qs;
        d.Qnext[((size_t)b * 2 + (1 - s)) * a.H + u] = other;
        d.QN[idx] = qs;
        if (d.spk_next) d.QSnext[idx] = d.spk_next[b] == s ? qs : other;
    }
}
template <int PARTY>
__global__ __launch_bounds__(256) void gru_gate_fwd_kernel(GateArgs a) {
    gru_gate_fwd_body<PARTY>(a, a.d[blockIdx.z], blockIdx.x * 256 + threadIdx.x);
}
struct GateBwdDir {
    const float* dh;        // [B x H] gradient wrt the cell output AFTER dropout (g / e cells) or wrt Q[t+1][spk] (party cell)
    const float* dh2;       // optional second addend (e cell: upstream dE_out[t])
    const float* R; const float* Z; const float* N; const float* HN;
    const float* hprev;
    float* dGI; float* dGH; // [B x 3H] gate gradients (kept for the weight-gradient GEMMs)
    float* dhdir;           // [B x H] direct path to hprev: dh' * z (+ (1 - m) dq for the party cell)
    const float* mval;      // party cell: [B]
    const int* spk;         // party cell: [B]; dh is then the [B x 2 x H] gradient wrt Q[t+1], read at party spk[b]
    uint32_t site;
    // party cell, optional: the previous (later-in-time) step's party-gradient assembly done here instead of in its own
    // launch — dQ[t+1][b][p] = p == spk_{t+1}[b] ? dQSp + dQSg : dQ[t+2][b][p]; written to pg_out (= dh) and used directly
    const float* pg_dQ = nullptr; const float* pg_dQSp = nullptr; const float* pg_dQSg = nullptr; const int* pg_spk = nullptr;
    float* pg_out = nullptr;
};
struct GateBwdArgs {
    GateBwdDir d[2];
    int B, H, row0;
    float p; int train;
    const uint64_t* rng; uint64_t add;
};

template <int PARTY>
__device__ __forceinline__ void gru_gate_bwd_body(const GateBwdArgs& a, const GateBwdDir& d, const int idx) {
    if (idx >= a.B * a.H) return;
    const int b = idx / a.H, u = idx - b * a.H, H3 = 3 * a.H;
    float dout;
    if (PARTY && d.pg_out != nullptr) {
        const int sn = d.pg_spk[b];
        const size_t o0 = ((size_t)b * 2) * a.H + u, o1 = o0 + a.H;
        const float own = d.pg_dQSp[idx] + d.pg_dQSg[idx];
        const float v0 = sn == 0 ? own : d.pg_dQ[o0];
        const float v1 = sn == 1 ? own : d.pg_dQ[o1];
        d.pg_out[o0] = v0;
        d.pg_out[o1] = v1;
        dout = d.spk[b] ? v1 : v0;
    } else {
        dout = PARTY ? d.dh[((size_t)b * 2 + d.spk[b]) * a.H + u] : d.dh[idx];
    }
    if (d.dh2) dout += d.dh2[idx];
    float pass = 0.f;
    if (PARTY) {
        const float m = d.mval[b];
        pass = m != 0.f ? 0.f : dout;      // padded step: q_t[spk] = q_{t-1}[spk]
        dout = m != 0.f ? dout : 0.f;
    }
    const DropCtx dc = make_drop(a.rng, a.add, d.site, a.p, a.train);
    const float dhn = dout * drop_mult1(dc, (uint32_t)(a.row0 + b), (uint32_t)a.H, (uint32_t)u);   // grad wrt h' (pre-dropout)
    const float r = d.R[idx], z = d.Z[idx], n = d.N[idx], hn = d.HN[idx], h = d.hprev[idx];
    const float dn = dhn * (1.0f - z);
    const float dz = dhn * (h - n);
    const float dnp = dn * (1.0f - n * n);
    const float drp = dnp * hn * r * (1.0f - r);
    const float dzp = dz * z * (1.0f - z);
    float* gi = d.dGI + (size_t)b * H3;
    float* gh = d.dGH + (size_t)b * H3;
    gi[u] = drp; gi[a.H + u] = dzp; gi[2 * a.H + u] = dnp;
    gh[u] = drp; gh[a.H + u] = dzp; gh[2 * a.H + u] = dnp * r;
    d.dhdir[idx] = dhn * z + pass;
}
template <int PARTY>
__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(GateBwdArgs a) {
    gru_gate_bwd_body<PARTY>(a, a.d[blockIdx.z], blockIdx.x * 256 + threadIdx.x);
}
// ------------------------------------------------------------------------------------------
// "general" context attention of step t over the global history g_0 .. g_{t-1} (model.py:161-164,193)
// one workgroup per (dialogue, direction); G rows are (S+1) blocks of B: block j+1 = g_j
// ------------------------------------------------------------------------------------------
constexpr int DR_MAXS = 112;
struct AttnDir {
    const float* XA;     // [B x H] query W_a U_t of this step
    const float* G;      // [(S+1) B x H]
    float* CT;           // [B x H] context
    float* alpha;        // [B x S x S], row t
};
struct AttnArgs { AttnDir d[2]; int B, H, S, t; };

// 1024 threads per (dialogue, direction).  The work is tiny (t x H multiply-adds) and latency-bound, so every phase keeps
// many independent loads in flight: 16 waves take 2 history steps each per pass for the scores, and the pooling splits
// the history in two halves per column with 8 loads in flight per thread.
constexpr int DR_AT = 1024;
__device__ __forceinline__ void drnn_scores(const float* __restrict__ q, const float* __restrict__ G, int B, int H, int b, int t,
                                            float* __restrict__ out, int lane, int w) {
    float qr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qr[i] = lane + 64 * i < H ? q[lane + 64 * i] : 0.f;
    for (int j0 = w; j0 < t; j0 += 32) {
        const int j1 = min(j0 + 16, t - 1);
        const float* g0 = G + ((size_t)(j0 + 1) * B + b) * H;
        const float* g1 = G + ((size_t)(j1 + 1) * B + b) * H;
        float v0[8], v1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = min(lane + 64 * i, H - 1);
            v0[i] = g0[k];
            v1[i] = g1[k];
        }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 += qr[i] * v0[i]; s1 += qr[i] * v1[i]; }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) {
            out[j0] = s0;
            if (j0 + 16 < t) out[j0 + 16] = s1;
        }
    }
}

__device__ __forceinline__ void drnn_attn_fwd_body(const AttnArgs& a, const AttnDir& d, const int b) {
    __shared__ float sc[DR_MAXS + 16];
    __shared__ float red[2];
    __shared__ float part[512];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, t = a.t;
    drnn_scores(d.XA + (size_t)b * a.H, d.G, a.B, a.H, b, t, sc, lane, w);     // H <= 512 (checked on the host)
    __syncthreads();
    // softmax over t <= 112 scores: the first two waves
    float v = tid < t ? sc[tid] : -INFINITY;
    float m = wave_max(v);
    if (tid < 128 && lane == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(red[0], red[1]);
    __syncthreads();
    float e = tid < t ? __expf(v - m) : 0.f;
    const float es = wave_sum(e);
    if (tid < 128 && lane == 0) red[w] = es;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1]);
    if (tid < t) {
        const float al = e * inv;
        sc[tid] = al;
        d.alpha[((size_t)b * a.S + t) * a.S + tid] = al;
    }
    __syncthreads();
    // context: c_k = sum_j alpha_j g_j[k]; thread (k, half) sums its half of the history, 8 loads in flight
    const int k = tid & 511, half = tid >> 9, jmid = (t + 1) >> 1;
    const int jb = half ? jmid : 0, je = half ? t : jmid;
    float c = 0.f;
    if (k < a.H) {
        const float* gk = d.G + ((size_t)a.B + b) * a.H + k;          // g_0[k]; step j adds j * B * H
        const size_t st = (size_t)a.B * a.H;
        for (int j = jb; j < je; j += 8) {
            float g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = gk[(size_t)min(j + u, je - 1) * st];
#pragma unroll
            for (int u = 0; u < 8; ++u) c += (j + u < je ? sc[j + u] : 0.f) * g[u];
        }
    }
    if (half) part[k] = c;
    __syncthreads();
    if (!half && k < a.H) d.CT[(size_t)b * a.H + k] = c + part[k];
}
__global__ __launch_bounds__(DR_AT) void drnn_attn_fwd_kernel(AttnArgs a) { drnn_attn_fwd_body(a, a.d[blockIdx.z], blockIdx.x); }

// Second launch of a forward step: both cells' gate math and, behind the global cell's, the context attention of the NEXT
// step.  Blocks [0, party_blocks) run the party gate body (1024 elements each); block party_blocks + b owns dialogue b:
// its first H threads produce g_t[b], then the whole workgroup pools g_0 .. g_t[b] for step t + 1 (the newest history row
// was written by this very workgroup: a workgroup-scope fence + barrier orders it).
struct GateAttnArgs { GateArgs g, p; AttnArgs at; int party_blocks, has_attn; };
__global__ __launch_bounds__(DR_AT) void drnn_gates_attn_fwd_kernel(GateAttnArgs a) {
    if ((int)blockIdx.x < a.party_blocks) {
        gru_gate_fwd_body<1>(a.p, a.p.d[blockIdx.z], blockIdx.x * DR_AT + threadIdx.x);
        return;
    }
    const int b = blockIdx.x - a.party_blocks;
    if ((int)threadIdx.x < a.g.H) gru_gate_fwd_body<0>(a.g, a.g.d[blockIdx.z], b * a.g.H + threadIdx.x);
    if (!a.has_attn) return;
    __threadfence_block();
    __syncthreads();
    drnn_attn_fwd_body(a.at, a.at.d[blockIdx.z], b);
}

struct AttnBwdDir {
    const float* dCT;    // [B x H]
    const float* XA;     // [B x H]
    const float* G;      // [(S+1) B x H]
    const float* alpha;  // [B x S x S]
    float* dG;           // [(S+1) B x H] accumulated gradient wrt g_j (block j+1)
    float* dXA;          // [B x H]
};
struct AttnBwdArgs { AttnBwdDir d[2]; int B, H, S, t; };

__device__ __forceinline__ void drnn_attn_bwd_body(const AttnBwdArgs& a, const AttnBwdDir& d, const int b) {
    __shared__ float da[DR_MAXS + 16];
    __shared__ float al[DR_MAXS + 16];
    __shared__ float red[2];
    __shared__ float part[512];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, t = a.t;
    const float* dc = d.dCT + (size_t)b * a.H;
    drnn_scores(dc, d.G, a.B, a.H, b, t, da, lane, w);                 // d alpha_j = <dc, g_j>
    if (tid < t) al[tid] = d.alpha[((size_t)b * a.S + t) * a.S + tid];
    __syncthreads();
    float dot = tid < t ? al[tid] * da[tid] : 0.f;
    dot = wave_sum(dot);
    if (tid < 128 && lane == 0) red[w] = dot;
    __syncthreads();
    const float tot = red[0] + red[1];
    __syncthreads();
    if (tid < t) da[tid] = al[tid] * (da[tid] - tot);                  // d score_j
    __syncthreads();
    // dXA_k = sum_j dscore_j g_j[k];  dG[j][k] += alpha_j dc_k + dscore_j x_k.  Thread (k, half) owns column k of its half of
    // the history: no other thread touches those dG elements (no race, no atomics).
    const int k = tid & 511, half = tid >> 9, jmid = (t + 1) >> 1;
    const int jb = half ? jmid : 0, je = half ? t : jmid;
    float dx = 0.f;
    if (k < a.H) {
        const float dck = dc[k], xk = d.XA[(size_t)b * a.H + k];
        const size_t st = (size_t)a.B * a.H, base = ((size_t)a.B + b) * a.H + k;
        for (int j = jb; j < je; j += 8) {
            float g[8], og[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t o = base + (size_t)min(j + u, je - 1) * st;
                g[u] = d.G[o];
                og[u] = d.dG[o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j + u < je) {
                    dx += da[j + u] * g[u];
                    d.dG[base + (size_t)(j + u) * st] = og[u] + al[j + u] * dck + da[j + u] * xk;
                }
            }
        }
    }
    if (half) part[k] = dx;
    __syncthreads();
    if (!half && k < a.H) d.dXA[(size_t)b * a.H + k] = dx + part[k];
}
__global__ __launch_bounds__(DR_AT) void drnn_attn_bwd_kernel(AttnBwdArgs a) { drnn_attn_bwd_body(a, a.d[blockIdx.z], blockIdx.x); }

// First launch of a backward step t: the attention backward of step t + 1 (it completes dG row t + 1 of its dialogue),
// then the global cell's gate gradient of step t on that row, in the workgroup that owns the dialogue; the party cell's
// gate gradient (independent of both) in blocks [0, party_blocks).
struct GateAttnBwdArgs { GateBwdArgs g, p; AttnBwdArgs at; int party_blocks, has_attn; };
__global__ __launch_bounds__(DR_AT) void drnn_gates_attn_bwd_kernel(GateAttnBwdArgs a) {
    if ((int)blockIdx.x < a.party_blocks) {
        gru_gate_bwd_body<1>(a.p, a.p.d[blockIdx.z], blockIdx.x * DR_AT + threadIdx.x);
        return;
    }
    const int b = blockIdx.x - a.party_blocks;
    if (a.has_attn) {
        drnn_attn_bwd_body(a.at, a.at.d[blockIdx.z], b);
        __threadfence_block();
        __syncthreads();
    }
    if ((int)threadIdx.x < a.g.H) gru_gate_bwd_body<0>(a.g, a.g.d[blockIdx.z], b * a.g.H + threadIdx.x);
}

// ------------------------------------------------------------------------------------------
// Emotion cell as ONE chain per dialogue (round 2).  e_t = drop(GRU_e(q_t[spk], e_{t-1})) takes q_t from the party cell
// but feeds nothing back into the recurrence, so: (1) its input product GI_e = QN W_ih^T + b_ih for ALL steps is one
// ordinary GEMM after the main loop; (2) what remains is a recurrence over e alone, independent per dialogue, with a
// 300 x 100 recurrent weight — one 512-thread workgroup per (dialogue, direction) runs all S steps with that weight in
// registers and no grid-level synchronisation.  The backward mirrors it (e chain first, then dQN for all steps as one
// GEMM).  Replaces 4 launches per step (2 skinny products, 2 gate kernels) by 4 launches per call.  D_e <= 128.
// ------------------------------------------------------------------------------------------
constexpr int EC_MAXHE = 128;
constexpr int EC_NT = 512;
struct EchainDir {
    const float* GI;          // [T x 3He] input pre-activations of all steps (b_ih included)
    const float* whh;         // [3He x He]
    const float* bhh;         // [3He]
    float* E;                 // [(S+1) B x He], block 0 = zeros
    float* R; float* Z; float* N; float* HN;      // [T x He] saved gates
    uint32_t site;
};
struct EchainArgs {
    EchainDir d[2];
    int S, B, He;
    float p; int train;
    const uint64_t* rng; uint64_t add;
};

__global__ __launch_bounds__(EC_NT) void drnn_echain_fwd_kernel(EchainArgs a) {
    const EchainDir& d = a.d[blockIdx.z];
    const int b = blockIdx.x, tid = threadIdx.x, He = a.He, H3 = 3 * He;
    __shared__ __attribute__((aligned(16))) float e_s[EC_MAXHE];
    __shared__ float gh_s[3 * EC_MAXHE];
    float w[EC_MAXHE];
    float bj = 0.f;
    if (tid < H3) {
        bj = d.bhh[tid];
#pragma unroll
        for (int k = 0; k < EC_MAXHE; ++k) w[k] = k < He ? d.whh[(size_t)tid * He + k] : 0.f;
    } else {
#pragma unroll
        for (int k = 0; k < EC_MAXHE; ++k) w[k] = 0.f;
    }
    if (tid < EC_MAXHE) e_s[tid] = 0.f;
    const DropCtx dc = make_drop(a.rng, a.add, d.site, a.p, a.train);
    // input pre-activations of the next 4 steps are always in flight (a step is ~1 us, an L2 / HBM round trip up to 2)
    constexpr int PF = 4;
    float gq[PF][3];
    const int uu = min(tid, He - 1);
    auto fetch = [&](int t, float (&g)[3]) __attribute__((always_inline)) {
        const float* gi = d.GI + ((size_t)min(t, a.S - 1) * a.B + b) * H3;
        g[0] = gi[uu]; g[1] = gi[He + uu]; g[2] = gi[2 * He + uu];
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fetch(i, gq[i]);
    for (int t = 0; t < a.S; ++t) {
        const size_t row = (size_t)t * a.B + b;
        const float gi0 = gq[0][0], gi1 = gq[0][1], gi2 = gq[0][2];
#pragma unroll
        for (int i = 0; i + 1 < PF; ++i) { gq[i][0] = gq[i + 1][0]; gq[i][1] = gq[i + 1][1]; gq[i][2] = gq[i + 1][2]; }
        fetch(t + PF, gq[PF - 1]);
        __syncthreads();                                   // e_s holds e_{t-1}
        {
            float acc = bj;
#pragma unroll
            for (int k4 = 0; k4 < EC_MAXHE / 4; ++k4) {
                if (4 * k4 < He) {                         // uniform
                    const float4 e4 = *reinterpret_cast<const float4*>(e_s + 4 * k4);
                    acc += e4.x * w[4 * k4] + e4.y * w[4 * k4 + 1] + e4.z * w[4 * k4 + 2] + e4.w * w[4 * k4 + 3];
                }
            }
            if (tid < H3) gh_s[tid] = acc;
        }
        __syncthreads();
        if (tid < He) {
            const float h = e_s[tid];
            const float r = sigm(gi0 + gh_s[tid]);
            const float z = sigm(gi1 + gh_s[He + tid]);
            const float hn = gh_s[2 * He + tid];
            const float n = tanhf(gi2 + r * hn);
            float hnew = (1.0f - z) * n + z * h;
            const size_t o = row * He + tid;
            d.R[o] = r; d.Z[o] = z; d.N[o] = n; d.HN[o] = hn;
            hnew *= drop_mult1(dc, (uint32_t)row, (uint32_t)He, (uint32_t)tid);
            d.E[((size_t)(t + 1) * a.B + b) * He + tid] = hnew;
            // (written after every thread has read e_s for this step's matvec: second barrier above)
            e_s[tid] = hnew;
        }
    }
}

struct EchainBwdDir {
    const float* d_e;         // [T x He] upstream gradient wrt every e_t
    const float* whh;         // [3He x He]
    const float* E;           // [(S+1) B x He]
    const float* R; const float* Z; const float* N; const float* HN;
    float* dGI; float* dGH;   // [T x 3He] gate gradients of all steps (kept for dQN and the weight gradients)
    uint32_t site;
};
struct EchainBwdArgs {
    EchainBwdDir d[2];
    int S, B, He;
    float p; int train;
    const uint64_t* rng; uint64_t add;
};

__global__ __launch_bounds__(EC_NT) void drnn_echain_bwd_kernel(EchainBwdArgs a) {
    const EchainBwdDir& d = a.d[blockIdx.z];
    const int b = blockIdx.x, tid = threadIdx.x, He = a.He, H3 = 3 * He;
    __shared__ float carry_s[EC_MAXHE];                   // gradient wrt e_t arriving from step t+1
    __shared__ float dgh_s[3 * EC_MAXHE];
    __shared__ float dir_s[EC_MAXHE];
    __shared__ float part_s[4][EC_MAXHE];
    // thread (k, part): column k of W_hh restricted to a quarter of its 3He rows
    const int k = tid & (EC_MAXHE - 1), part = tid >> 7, jper = (H3 + 3) / 4, j0 = part * jper;
    float wT[(3 * EC_MAXHE + 3) / 4];
#pragma unroll
    for (int i = 0; i < (3 * EC_MAXHE + 3) / 4; ++i) wT[i] = (k < He && i < jper && j0 + i < H3) ? d.whh[(size_t)(j0 + i) * He + k] : 0.f;
    if (tid < EC_MAXHE) carry_s[tid] = 0.f;
    const DropCtx dc = make_drop(a.rng, a.add, d.site, a.p, a.train);
    // the saved gates / states / upstream gradients of the next 4 steps are always in flight
    constexpr int PF = 4;
    float sq[PF][6];
    const int uu = min(tid, He - 1);
    auto fetch = [&](int t, float (&v)[6]) __attribute__((always_inline)) {
        const size_t o = ((size_t)max(t, 0) * a.B + b) * He + uu;
        v[0] = d.R[o]; v[1] = d.Z[o]; v[2] = d.N[o]; v[3] = d.HN[o]; v[4] = d.E[o]; v[5] = d.d_e[o];
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fetch(a.S - 1 - i, sq[i]);
    for (int t = a.S - 1; t >= 0; --t) {
        const size_t row = (size_t)t * a.B + b;
        const float r = sq[0][0], z = sq[0][1], n = sq[0][2], hn = sq[0][3], h = sq[0][4], de = sq[0][5];
#pragma unroll
        for (int i = 0; i + 1 < PF; ++i)
#pragma unroll
            for (int q = 0; q < 6; ++q) sq[i][q] = sq[i + 1][q];
        fetch(t - PF, sq[PF - 1]);
        __syncthreads();                                   // carry_s complete
        if (tid < He) {
            const float dout = carry_s[tid] + de;
            const float dhn = dout * drop_mult1(dc, (uint32_t)row, (uint32_t)He, (uint32_t)tid);
            const float dn = dhn * (1.0f - z);
            const float dz = dhn * (h - n);
            const float dnp = dn * (1.0f - n * n);
            const float drp = dnp * hn * r * (1.0f - r);
            const float dzp = dz * z * (1.0f - z);
            float* gi = d.dGI + row * H3;
            float* gh = d.dGH + row * H3;
            gi[tid] = drp; gi[He + tid] = dzp; gi[2 * He + tid] = dnp;
            gh[tid] = drp; gh[He + tid] = dzp; gh[2 * He + tid] = dnp * r;
            dgh_s[tid] = drp; dgh_s[He + tid] = dzp; dgh_s[2 * He + tid] = dnp * r;
            dir_s[tid] = dhn * z;
        }
        __syncthreads();
        {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < (3 * EC_MAXHE + 3) / 4; ++i)
                if (i < jper) acc += dgh_s[min(j0 + i, H3 - 1)] * wT[i];     // (rows beyond H3 carry zero weights)
            part_s[part][k] = acc;
        }
        __syncthreads();
        if (tid < He) carry_s[tid] = dir_s[tid] + ((part_s[0][tid] + part_s[1][tid]) + (part_s[2][tid] + part_s[3][tid]));
    }
}

// ------------------------------------------------------------------------------------------
// layouts of the saved-for-backward block and of the workspace (floats; every region 16-byte aligned)
// ------------------------------------------------------------------------------------------
struct DrnnSaved {
    int64_t XG, XP, XA, G, Q, E, QS, CT, QN, Rg, Zg, Ng, HNg, Rp, Zp, Np, HNp, Re, Ze, Ne, HNe, total;
};
static DrnnSaved drnn_saved(const ganffn_drnn_cfg* c) {
    DrnnSaved s;
    const int64_t T = (int64_t)c->S * c->B, T1 = (int64_t)(c->S + 1) * c->B, H = c->H, He = c->He;
    int64_t p = 0;
    auto take = [&](int64_t n) { int64_t r = p; p += (n + 3) & ~int64_t(3); return r; };
    s.XG = take(T * 3 * H); s.XP = take(T * 3 * H); s.XA = take(T * H);
    s.G = take(T1 * H); s.Q = take(T1 * 2 * H); s.E = take(T1 * He);
    s.QS = take(T1 * H); s.CT = take(T * H); s.QN = take(T * H);
    s.Rg = take(T * H); s.Zg = take(T * H); s.Ng = take(T * H); s.HNg = take(T * H);
    s.Rp = take(T * H); s.Zp = take(T * H); s.Np = take(T * H); s.HNp = take(T * H);
    s.Re = take(T * He); s.Ze = take(T * He); s.Ne = take(T * He); s.HNe = take(T * He);
    s.total = p;
    return s;
}
struct DrnnWs {
    int64_t GI, GH, GIp, GHp, dGIg, dGHg, dGIp, dGHp, dGIe, dGHe, dXA, dCT, dG, dQa, dQb, dEa, dEb, dQsel, dQSp, dQSg, dhdir, dhdirG, dQN, GIe, dQNall, WT, total;
};
static DrnnWs drnn_ws(const ganffn_drnn_cfg* c) {
    DrnnWs w;
    const int64_t T = (int64_t)c->S * c->B, T1 = (int64_t)(c->S + 1) * c->B, B = c->B, H = c->H, He = c->He;
    int64_t p = 0;
    auto take = [&](int64_t n) { int64_t r = p; p += (n + 3) & ~int64_t(3); return r; };
    w.GI = take(B * 3 * H); w.GH = take(B * 3 * H); w.GIp = take(B * 3 * H); w.GHp = take(B * 3 * H);
    w.dGIg = take(T * 3 * H); w.dGHg = take(T * 3 * H); w.dGIp = take(T * 3 * H); w.dGHp = take(T * 3 * H);
    w.dGIe = take(T * 3 * He); w.dGHe = take(T * 3 * He);
    w.dXA = take(T * H); w.dCT = take(B * H); w.dG = take(T1 * H);
    w.dQa = take(B * 2 * H); w.dQb = take(B * 2 * H); w.dEa = take(B * He); w.dEb = take(B * He);
    w.dQsel = take(B * H); w.dQSp = take(B * H); w.dQSg = take(B * H); w.dhdir = take(B * H); w.dhdirG = take(B * H); w.dQN = take(B * H);
    w.GIe = take(T * 3 * He); w.dQNall = take(T * H);       // emotion chain: input pre-activations / dQN of all steps
    w.WT = take(4 * H * 3 * H);                              // transposed recurrent weights (backward): 4 x [H x 3H]
    w.total = p;
    return w;
}

static int check_drnn(const ganffn_drnn_cfg* c, int ndir) {
    GF_CHECK_ARG(c, "null drnn cfg");
    GF_CHECK_ARG(ndir == 1 || ndir == 2, "drnn: ndir=%d", ndir);
    GF_CHECK_ARG(c->S >= 1 && c->S <= DR_MAXS && c->B >= 1 && c->B <= 32, "drnn: S=%d (<= %d), B=%d (<= 32)", c->S, DR_MAXS, c->B);
    GF_CHECK_ARG(c->Dm >= 4 && (c->Dm & 3) == 0 && c->H >= 4 && (c->H & 3) == 0 && c->He >= 4 && (c->He & 3) == 0,
                 "drnn: D_m=%d, D_g=D_p=%d, D_e=%d must be multiples of 4", c->Dm, c->H, c->He);
    GF_CHECK_ARG(c->H <= 512, "drnn: D_g = D_p = %d > 512 (attention kernels hold one column per thread)", c->H);
    GF_CHECK_ARG(c->p >= 0.f && c->p < 1.f, "drnn: dropout p out of [0,1)");
    return 0;
}

static int memset_f(float* p, int64_t n, hipStream_t st) {
    hipError_t e = hipMemsetAsync(p, 0, (size_t)n * sizeof(float), st);
    if (e != hipSuccess) return fail((int)e, "drnn: memset failed: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace ganffn

using namespace ganffn;

// test / measurement hook: one skinny product C[M x N] = A[M x K] W^T (nn = 0, W [N x K]) or A W (nn = 1, W [K x N]),
// replicated `copies` (<= 8) times in one launch like a step of the recurrence (2 directions x 2 cells x 2 products)
extern "C" int ganffn_drnn_skinny(int nn, int copies, const float* A, const float* W, float* C, int M, int N, int K, void* stream) {
    GF_CHECK_ARG(A && W && C && copies >= 1 && copies <= 8, "drnn_skinny: bad arguments");
    SkinnyGroup sg;
    for (int i = 0; i < copies; ++i)
        sg.p[i] = SkinnyProb{A, K, W + (size_t)i * N * K, nn ? N : K, nullptr, 0, nullptr, nullptr, C + (size_t)i * M * N, N, M, N, K};
    return launch_skinny(sg, copies, nn != 0, (hipStream_t)stream);
}

extern "C" int64_t ganffn_drnn_saved_floats(const ganffn_drnn_cfg* c) { return check_drnn(c, 1) ? -1 : drnn_saved(c).total; }
extern "C" int64_t ganffn_drnn_workspace_floats(const ganffn_drnn_cfg* c) { return check_drnn(c, 1) ? -1 : drnn_ws(c).total; }

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
extern "C" int ganffn_drnn_fwd(const ganffn_drnn_cfg* c, int ndir, const float* const* U, const int32_t* const* spk,
                               const float* const* mval, const ganffn_drnn_params* prm, float* const* e_out,
                               float* const* alpha, float* const* saved, float* const* workspace, const uint64_t* rng,
                               uint64_t add, void* stream) {
    GF_TRY(check_drnn(c, ndir));
    GF_CHECK_ARG(U && spk && mval && prm && e_out && alpha && saved && workspace, "drnn_fwd: null pointer");
    GF_CHECK_ARG(!(c->train && c->p > 0.f) || rng, "drnn_fwd: rng required in train mode");
    hipStream_t st = (hipStream_t)stream;
    const int S = c->S, B = c->B, Dm = c->Dm, H = c->H, He = c->He, T = S * B;
    const DrnnSaved so = drnn_saved(c);
    const DrnnWs wo = drnn_ws(c);
    for (int z = 0; z < ndir; ++z) {
        GF_CHECK_ARG(U[z] && spk[z] && mval[z] && e_out[z] && alpha[z] && saved[z] && workspace[z] && aligned16(U[z]) &&
                     aligned16(saved[z]) && aligned16(workspace[z]), "drnn_fwd: direction %d: null or misaligned buffer", z);
        float* sv = saved[z];
        // U-only terms for all steps: XG = U Wih_g[:, :Dm]^T + bih_g, XP likewise, XA = U Watt^T
        EpiArgs e;
        e.bias = prm[z].g_bih;
        GF_TRY(launch_gemm_nt(U[z], Dm, prm[z].g_wih, Dm + H, sv + so.XG, 3 * H, T, 3 * H, Dm, EPI_NONE, e, st));
        e.bias = prm[z].p_bih;
        GF_TRY(launch_gemm_nt(U[z], Dm, prm[z].p_wih, Dm + H, sv + so.XP, 3 * H, T, 3 * H, Dm, EPI_NONE, e, st));
        e.bias = nullptr;
        GF_TRY(launch_gemm_nt(U[z], Dm, prm[z].att_w, Dm, sv + so.XA, H, T, H, Dm, EPI_NONE, e, st));
        // zero initial states: G[0], Q[0], E[0], QS[0], CT[0]; alpha (entries j >= t stay zero)
        GF_TRY(memset_f(sv + so.G, (int64_t)B * H, st));
        GF_TRY(memset_f(sv + so.Q, (int64_t)B * 2 * H, st));
        GF_TRY(memset_f(sv + so.E, (int64_t)B * He, st));
        GF_TRY(memset_f(sv + so.QS, (int64_t)B * H, st));
        GF_TRY(memset_f(sv + so.CT, (int64_t)B * H, st));
        GF_TRY(memset_f(alpha[z], (int64_t)B * S * S, st));
    }
    const dim3 gH((B * H + 255) / 256, 1, ndir), gHe((B * He + 255) / 256, 1, ndir);
    const bool echain = He <= EC_MAXHE;          // emotion cell as one chain per dialogue after the main loop
    // A step is two launches: the four products of the global and the party cell (the party cell needs c_t but not g_t, the
    // global cell neither: they are independent within a step), then both cells' gate math with the context attention of
    // step t + 1 behind the global cell's (c_0 = 0: step 0 needs none).
    for (int t = 0; t < S; ++t) {
        const int64_t r0 = (int64_t)t * B, r1 = (int64_t)(t + 1) * B;
        SkinnyGroup sg;
        // ---- global cell: GI = XG[t] + QS[t] Wih_g[:, Dm:]^T ; GH = G[t] Whh_g^T + bhh_g
        // ---- party cell (speaker): GI = XP[t] + CT[t] Wih_p[:, Dm:]^T ; GH = QS[t] Whh_p^T + bhh_p
        for (int z = 0; z < ndir; ++z) {
            float* sv = saved[z]; float* ws = workspace[z];
            sg.p[4 * z] = SkinnyProb{sv + so.QS + r0 * H, H, prm[z].g_wih + Dm, Dm + H, sv + so.XG + r0 * 3 * H, 3 * H, nullptr, nullptr, ws + wo.GI, 3 * H, B, 3 * H, H};
            sg.p[4 * z + 1] = SkinnyProb{sv + so.G + r0 * H, H, prm[z].g_whh, H, nullptr, 0, nullptr, prm[z].g_bhh, ws + wo.GH, 3 * H, B, 3 * H, H};
            sg.p[4 * z + 2] = SkinnyProb{sv + so.CT + r0 * H, H, prm[z].p_wih + Dm, Dm + H, sv + so.XP + r0 * 3 * H, 3 * H, nullptr, nullptr, ws + wo.GIp, 3 * H, B, 3 * H, H};
            sg.p[4 * z + 3] = SkinnyProb{sv + so.QS + r0 * H, H, prm[z].p_whh, H, nullptr, 0, nullptr, prm[z].p_bhh, ws + wo.GHp, 3 * H, B, 3 * H, H};
        }
        GF_TRY(launch_skinny(sg, 4 * ndir, false, st));
        GateArgs ga, gp;
        ga.B = B; ga.H = H; ga.row0 = (int)r0; ga.p = c->p; ga.train = c->train; ga.rng = rng; ga.add = add;
        gp = ga;
        for (int z = 0; z < ndir; ++z) {
            float* sv = saved[z]; float* ws = workspace[z];
            ga.d[z] = GateDir{ws + wo.GI, ws + wo.GH, sv + so.G + r0 * H, sv + so.Rg + r0 * H, sv + so.Zg + r0 * H, sv + so.Ng + r0 * H,
                              sv + so.HNg + r0 * H, sv + so.G + r1 * H, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              SITE_DRNN_G + 4u * z};
            gp.d[z] = GateDir{ws + wo.GIp, ws + wo.GHp, sv + so.QS + r0 * H, sv + so.Rp + r0 * H, sv + so.Zp + r0 * H, sv + so.Np + r0 * H,
                              sv + so.HNp + r0 * H, nullptr, spk[z] + r0, t + 1 < S ? spk[z] + r1 : nullptr, mval[z] + r0,
                              sv + so.Q + r0 * 2 * H, sv + so.Q + r1 * 2 * H, sv + so.QN + r0 * H, sv + so.QS + r1 * H,
                              SITE_DRNN_P + 4u * z};
        }
        GateAttnArgs gaa;
        gaa.g = ga; gaa.p = gp;
        gaa.party_blocks = (B * H + DR_AT - 1) / DR_AT;
        gaa.has_attn = t + 1 < S;
        gaa.at.B = B; gaa.at.H = H; gaa.at.S = S; gaa.at.t = t + 1;
        for (int z = 0; z < ndir; ++z)
            gaa.at.d[z] = AttnDir{saved[z] + so.XA + r1 * H, saved[z] + so.G, saved[z] + so.CT + r1 * H, alpha[z]};
        hipLaunchKernelGGL(drnn_gates_attn_fwd_kernel, dim3(gaa.party_blocks + B, 1, ndir), dim3(DR_AT), 0, st, gaa);
        GF_LAUNCH_CHECK();
        if (echain) continue;
        // ---- emotion cell: GI = QN[t] Wih_e^T + bih_e ; GH = E[t] Whh_e^T + bhh_e
        for (int z = 0; z < ndir; ++z) {
            float* sv = saved[z]; float* ws = workspace[z];
            sg.p[2 * z] = SkinnyProb{sv + so.QN + r0 * H, H, prm[z].e_wih, H, nullptr, 0, nullptr, prm[z].e_bih, ws + wo.GI, 3 * He, B, 3 * He, H};
            sg.p[2 * z + 1] = SkinnyProb{sv + so.E + r0 * He, He, prm[z].e_whh, He, nullptr, 0, nullptr, prm[z].e_bhh, ws + wo.GH, 3 * He, B, 3 * He, He};
        }
        GF_TRY(launch_skinny(sg, 2 * ndir, false, st));
        ga.H = He;
        for (int z = 0; z < ndir; ++z) {
            float* sv = saved[z]; float* ws = workspace[z];
            ga.d[z] = GateDir{ws + wo.GI, ws + wo.GH, sv + so.E + r0 * He, sv + so.Re + r0 * He, sv + so.Ze + r0 * He, sv + so.Ne + r0 * He,
                              sv + so.HNe + r0 * He, sv + so.E + r1 * He, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              SITE_DRNN_E + 4u * z};
        }
        hipLaunchKernelGGL(gru_gate_fwd_kernel<0>, gHe, dim3(256), 0, st, ga);
        GF_LAUNCH_CHECK();
    }
    if (echain) {
        EchainArgs ea;
        ea.S = S; ea.B = B; ea.He = He; ea.p = c->p; ea.train = c->train; ea.rng = rng; ea.add = add;
        for (int z = 0; z < ndir; ++z) {
            float* sv = saved[z]; float* ws = workspace[z];
            EpiArgs e;
            e.bias = prm[z].e_bih;                 // GI_e of every step: one GEMM over all T rows of q_t[spk]
            GF_TRY(launch_gemm_nt(sv + so.QN, H, prm[z].e_wih, H, ws + wo.GIe, 3 * He, T, 3 * He, H, EPI_NONE, e, st));
            ea.d[z] = EchainDir{ws + wo.GIe, prm[z].e_whh, prm[z].e_bhh, sv + so.E, sv + so.Re, sv + so.Ze, sv + so.Ne, sv + so.HNe,
                                SITE_DRNN_E + 4u * z};
        }
        hipLaunchKernelGGL(drnn_echain_fwd_kernel, dim3(B, 1, ndir), dim3(EC_NT), 0, st, ea);
        GF_LAUNCH_CHECK();
    }
    for (int z = 0; z < ndir; ++z) {
        hipError_t er = hipMemcpyAsync(e_out[z], saved[z] + so.E + (int64_t)B * He, (size_t)T * He * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (er != hipSuccess) return fail((int)er, "drnn_fwd: memcpy failed: %s", hipGetErrorString(er));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
extern "C" int ganffn_drnn_bwd(const ganffn_drnn_cfg* c, int ndir, const float* const* d_e, const float* const* U,
                               const int32_t* const* spk, const float* const* mval, const ganffn_drnn_params* prm,
                               const ganffn_drnn_grads* grd, float* const* dU, const float* const* alpha,
                               const float* const* saved, float* const* workspace, const uint64_t* rng, uint64_t add,
                               void* stream) {
    GF_TRY(check_drnn(c, ndir));
    GF_CHECK_ARG(d_e && U && spk && mval && prm && grd && dU && alpha && saved && workspace, "drnn_bwd: null pointer");
    GF_CHECK_ARG(!(c->train && c->p > 0.f) || rng, "drnn_bwd: rng required in train mode");
    hipStream_t st = (hipStream_t)stream;
    const int S = c->S, B = c->B, Dm = c->Dm, H = c->H, He = c->He, T = S * B;
    const DrnnSaved so = drnn_saved(c);
    const DrnnWs wo = drnn_ws(c);
    for (int z = 0; z < ndir; ++z) {
        GF_CHECK_ARG(d_e[z] && U[z] && dU[z] && saved[z] && workspace[z] && alpha[z], "drnn_bwd: direction %d: null buffer", z);
        float* ws = workspace[z];
        GF_TRY(memset_f(ws + wo.dG, (int64_t)(S + 1) * B * H, st));
        GF_TRY(memset_f(ws + wo.dQa, (int64_t)B * 2 * H, st));
        GF_TRY(memset_f(ws + wo.dEa, (int64_t)B * He, st));
        GF_TRY(memset_f(ws + wo.dXA, (int64_t)B * H, st));          // step 0 has no attention: dXA[0] = 0
    }
    const dim3 gH((B * H + 255) / 256, 1, ndir), gHe((B * He + 255) / 256, 1, ndir);
    const bool echain = He <= EC_MAXHE;
    {
        // the step products contract over the 3H gate rows: transposed copies [H x 3H] make them row-wise reads (the NT
        // kernel, 16.6 -> see DESIGN.md) — order per direction: p_wih[:, Dm:], p_whh, g_wih[:, Dm:], g_whh
        TrGroup tr;
        tr.rows = 3 * H; tr.cols = H;
        for (int z = 0; z < ndir; ++z) {
            float* wt = workspace[z] + wo.WT;
            const float* src[4] = {prm[z].p_wih + Dm, prm[z].p_whh, prm[z].g_wih + Dm, prm[z].g_whh};
            const int ld[4] = {Dm + H, H, Dm + H, H};
            for (int i = 0; i < 4; ++i) { tr.in[4 * z + i] = src[i]; tr.ld_in[4 * z + i] = ld[i]; tr.out[4 * z + i] = wt + (int64_t)i * H * 3 * H; }
        }
        hipLaunchKernelGGL(drnn_transpose_kernel, dim3((H + 31) / 32, (3 * H + 31) / 32, 4 * ndir), dim3(256), 0, st, tr);
        GF_LAUNCH_CHECK();
    }
    if (echain) {
        // emotion chain first (it depends on nothing else): gate gradients of all steps, then dQN of all steps in one GEMM
        EchainBwdArgs eb;
        eb.S = S; eb.B = B; eb.He = He; eb.p = c->p; eb.train = c->train; eb.rng = rng; eb.add = add;
        for (int z = 0; z < ndir; ++z) {
            const float* sv = saved[z]; float* ws = workspace[z];
            eb.d[z] = EchainBwdDir{d_e[z], prm[z].e_whh, sv + so.E, sv + so.Re, sv + so.Ze, sv + so.Ne, sv + so.HNe, ws + wo.dGIe,
                                   ws + wo.dGHe, SITE_DRNN_E + 4u * z};
        }
        hipLaunchKernelGGL(drnn_echain_bwd_kernel, dim3(B, 1, ndir), dim3(EC_NT), 0, st, eb);
        GF_LAUNCH_CHECK();
        for (int z = 0; z < ndir; ++z) {
            float* ws = workspace[z];
            EpiArgs e0;
            GF_TRY(launch_gemm_nn(ws + wo.dGIe, 3 * He, prm[z].e_wih, H, ws + wo.dQNall, H, T, H, 3 * He, EPI_NONE, e0, st));
        }
    }
    for (int t = S - 1; t >= 0; --t) {
        const int64_t r0 = (int64_t)t * B, r1 = (int64_t)(t + 1) * B;
        const bool even = ((S - 1 - t) & 1) == 0;            // ping-pong of the recurrent gradients
        const int64_t dQin = even ? wo.dQa : wo.dQb;      // gradient wrt Q[t+1] (assembled by the party gate kernel below)
        const int64_t dEin = even ? wo.dEa : wo.dEb, dEout = even ? wo.dEb : wo.dEa;
        SkinnyGroup sg;
        GateBwdArgs gb;
        gb.B = B; gb.row0 = (int)r0; gb.p = c->p; gb.train = c->train; gb.rng = rng; gb.add = add;
        // ---- emotion cell (per step only when the chain kernels do not apply)
        gb.H = He;
        if (!echain) {
        for (int z = 0; z < ndir; ++z) {
            const float* sv = saved[z]; float* ws = workspace[z];
            gb.d[z] = GateBwdDir{ws + dEin, d_e[z] + r0 * He, sv + so.Re + r0 * He, sv + so.Ze + r0 * He, sv + so.Ne + r0 * He,
                                 sv + so.HNe + r0 * He, sv + so.E + r0 * He, ws + wo.dGIe + r0 * 3 * He, ws + wo.dGHe + r0 * 3 * He,
                                 ws + wo.dhdir, nullptr, nullptr, SITE_DRNN_E + 4u * z};
        }
        hipLaunchKernelGGL(gru_gate_bwd_kernel<0>, gHe, dim3(256), 0, st, gb);
        GF_LAUNCH_CHECK();
        for (int z = 0; z < ndir; ++z) {
            float* ws = workspace[z];
            // dQN[t] = dGI_e Wih_e ; dE_rec = dGH_e Whh_e + dhdir
            sg.p[2 * z] = SkinnyProb{ws + wo.dGIe + r0 * 3 * He, 3 * He, prm[z].e_wih, H, nullptr, 0, nullptr, nullptr, ws + wo.dQN, H, B, H, 3 * He};
            sg.p[2 * z + 1] = SkinnyProb{ws + wo.dGHe + r0 * 3 * He, 3 * He, prm[z].e_whh, He, ws + wo.dhdir, He, nullptr, nullptr, ws + dEout, He, B, He, 3 * He};
        }
        GF_TRY(launch_skinny(sg, 2 * ndir, true, st));
        }
        // ---- gate gradients of both cells in one launch.  Party cell: gradient wrt Q[t+1][spk] = dQ[t+1][spk] + dQN (selected
        // inside the kernel).  Global cell: dg_t = dG[t+1] — the attention uses of g_t at later steps and the recurrent path
        // were all added by the steps already done.
        gb.H = H;
        GateBwdArgs gg = gb;
        for (int z = 0; z < ndir; ++z) {
            const float* sv = saved[z]; float* ws = workspace[z];
            gb.d[z] = GateBwdDir{ws + dQin, echain ? ws + wo.dQNall + r0 * H : ws + wo.dQN, sv + so.Rp + r0 * H, sv + so.Zp + r0 * H, sv + so.Np + r0 * H, sv + so.HNp + r0 * H,
                                 sv + so.QS + r0 * H, ws + wo.dGIp + r0 * 3 * H, ws + wo.dGHp + r0 * 3 * H, ws + wo.dhdir, mval[z] + r0,
                                 spk[z] + r0, SITE_DRNN_P + 4u * z};
            if (t < S - 1) {      // assemble dQ[t+1] here (the previous iteration's party-gradient launch is gone)
                const bool even1 = ((S - 2 - t) & 1) == 0;
                gb.d[z].pg_dQ = ws + (even1 ? wo.dQa : wo.dQb);
                gb.d[z].pg_out = ws + dQin;
                gb.d[z].pg_dQSp = ws + wo.dQSp; gb.d[z].pg_dQSg = ws + wo.dQSg; gb.d[z].pg_spk = spk[z] + r1;
            }
            gg.d[z] = GateBwdDir{ws + wo.dG + r1 * H, nullptr, sv + so.Rg + r0 * H, sv + so.Zg + r0 * H, sv + so.Ng + r0 * H, sv + so.HNg + r0 * H,
                                 sv + so.G + r0 * H, ws + wo.dGIg + r0 * 3 * H, ws + wo.dGHg + r0 * 3 * H, ws + wo.dhdirG, nullptr,
                                 nullptr, SITE_DRNN_G + 4u * z};
        }
        GateAttnBwdArgs gab;
        gab.g = gg; gab.p = gb;
        gab.party_blocks = (B * H + DR_AT - 1) / DR_AT;
        gab.has_attn = t + 1 < S;             // attention backward of step t + 1: needs its dCT (the products of step t + 1)
        gab.at.B = B; gab.at.H = H; gab.at.S = S; gab.at.t = t + 1;
        for (int z = 0; z < ndir; ++z)
            gab.at.d[z] = AttnBwdDir{workspace[z] + wo.dCT, saved[z] + so.XA + r1 * H, saved[z] + so.G, alpha[z], workspace[z] + wo.dG,
                                     workspace[z] + wo.dXA + r1 * H};
        hipLaunchKernelGGL(drnn_gates_attn_bwd_kernel, dim3(gab.party_blocks + B, 1, ndir), dim3(DR_AT), 0, st, gab);
        GF_LAUNCH_CHECK();
        // ---- the four dgrad products of the step in one launch
        for (int z = 0; z < ndir; ++z) {
            float* ws = workspace[z];
            const float* wt = ws + wo.WT;
            const int64_t WM = (int64_t)H * 3 * H;
            // dCT[t] = dGI_p Wih_p[:, Dm:] ; dQS_p = dGH_p Whh_p + dhdir
            sg.p[4 * z] = SkinnyProb{ws + wo.dGIp + r0 * 3 * H, 3 * H, wt, 3 * H, nullptr, 0, nullptr, nullptr, ws + wo.dCT, H, B, H, 3 * H};
            sg.p[4 * z + 1] = SkinnyProb{ws + wo.dGHp + r0 * 3 * H, 3 * H, wt + WM, 3 * H, ws + wo.dhdir, H, nullptr, nullptr, ws + wo.dQSp, H, B, H, 3 * H};
            // dQS_g = dGI_g Wih_g[:, Dm:] ; dG[t] += dGH_g Whh_g + dhdir (in place: second addend = the output block itself)
            sg.p[4 * z + 2] = SkinnyProb{ws + wo.dGIg + r0 * 3 * H, 3 * H, wt + 2 * WM, 3 * H, nullptr, 0, nullptr, nullptr, ws + wo.dQSg, H, B, H, 3 * H};
            sg.p[4 * z + 3] = SkinnyProb{ws + wo.dGHg + r0 * 3 * H, 3 * H, wt + 3 * WM, 3 * H, ws + wo.dhdirG, H, ws + wo.dG + r0 * H, nullptr,
                                         ws + wo.dG + r0 * H, H, B, H, 3 * H};
        }
        GF_TRY(launch_skinny(sg, 4 * ndir, false, st));
    }
    // ---- dU and the deferred weight gradients (all steps at once)
    for (int z = 0; z < ndir; ++z) {
        const float* sv = saved[z]; float* ws = workspace[z];
        const ganffn_drnn_grads& g = grd[z];
        EpiArgs e0, e1;
        GF_TRY(launch_gemm_nn(ws + wo.dGIg, 3 * H, prm[z].g_wih, Dm + H, dU[z], Dm, T, Dm, 3 * H, EPI_NONE, e0, st));
        e1.aux_in = dU[z];
        GF_TRY(launch_gemm_nn(ws + wo.dGIp, 3 * H, prm[z].p_wih, Dm + H, dU[z], Dm, T, Dm, 3 * H, EPI_NONE, e1, st));
        GF_TRY(launch_gemm_nn(ws + wo.dXA, H, prm[z].att_w, Dm, dU[z], Dm, T, Dm, H, EPI_NONE, e1, st));
        if (g.g_wih) {
            TnDesc tn[12];
            int n = 0;
            tn[n++] = TnDesc{ws + wo.dGIg, 3 * H, U[z], Dm, g.g_wih, Dm + H, g.g_bih, 3 * H, Dm, T};
            tn[n++] = TnDesc{ws + wo.dGIg, 3 * H, sv + so.QS, H, g.g_wih + Dm, Dm + H, nullptr, 3 * H, H, T};
            tn[n++] = TnDesc{ws + wo.dGHg, 3 * H, sv + so.G, H, g.g_whh, H, g.g_bhh, 3 * H, H, T};
            tn[n++] = TnDesc{ws + wo.dGIp, 3 * H, U[z], Dm, g.p_wih, Dm + H, g.p_bih, 3 * H, Dm, T};
            tn[n++] = TnDesc{ws + wo.dGIp, 3 * H, sv + so.CT, H, g.p_wih + Dm, Dm + H, nullptr, 3 * H, H, T};
            tn[n++] = TnDesc{ws + wo.dGHp, 3 * H, sv + so.QS, H, g.p_whh, H, g.p_bhh, 3 * H, H, T};
            tn[n++] = TnDesc{ws + wo.dGIe, 3 * He, sv + so.QN, H, g.e_wih, H, g.e_bih, 3 * He, H, T};
            tn[n++] = TnDesc{ws + wo.dGHe, 3 * He, sv + so.E, He, g.e_whh, He, g.e_bhh, 3 * He, He, T};
            tn[n++] = TnDesc{ws + wo.dXA, H, U[z], Dm, g.att_w, Dm, nullptr, H, Dm, T};
            GF_TRY(launch_gemm_tn_grouped(tn, n, st));
        }
    }
    return 0;
}
