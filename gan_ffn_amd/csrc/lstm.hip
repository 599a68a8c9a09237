// lstm.hip — one BIDIRECTIONAL LSTM layer, forward and backward: the recurrence of
// `nn.LSTM(input_size, hidden_size, num_layers=4, bidirectional=True, dropout=p)` inside MELDLSTMModel
// (/root/reference/model.py:520-562, constructed at /root/reference/train_MELD.py:147-151 with D_m = 600, D_e = 300) on padded
// (seq_len, batch, features) tensors — the reference hands the LSTM the padded batch as it is (model.py:546: no packing, no
// mask), so every dialogue runs all S steps.
//
// torch's cell (gate rows of the weights in the order i, f, g, o):
//     G_t = x_t W_ih^T + b_ih + h_{t-1} W_hh^T + b_hh;   i, f, o = sigmoid(G_i, G_f, G_o),  g = tanh(G_g)
//     c_t = f c_{t-1} + i g;   h_t = o tanh(c_t);   the reverse direction runs t = S-1 .. 0; out[t] = [h_t fwd | h_t rev]
//
// Built from the pieces the DialogueRNN recurrence (dialogue_rnn.hip) already has, the same way and for the same reason — a
// step is a chain of tiny dependent products, so what counts is launches and round trips, not FLOPs:
//   * everything that depends on x alone is hoisted out of the chain: xg_d = x W_ih_d^T + b_ih_d for ALL steps is one GEMM per
//     direction (gemm.hip, fp32 MFMA);
//   * a step = ONE skinny MFMA product launch for BOTH directions (skinny_nt_kernel: [B x H] x [4H x H]^T + xg[t] + b_hh, the
//     dialogues on the n axis of v_mfma_f32_16x16x4_f32, K split over the waves and summed in wave order) + ONE gate launch for
//     both directions (sigmoid / tanh, the cell update, h_t straight into its half of out[t]): 2 launches per step;
//   * backward: per step one gate-gradient launch (dG_t and dc_{t-1} of both directions) and one skinny NN product
//     (dh_{t-1} = d_out[t-1] + dG_t W_hh, the weight read row-wise as stored); every weight gradient is DEFERRED: dG of all steps
//     is kept and four TN GEMMs after the loop compute dW_ih = dG^T x and dW_hh = dG^T h_prev over all tokens (owner-accumulated or
//     slab + ordered reduce: no atomics), the bias gradients are fixed-order column sums, dx = sum_d dG_d W_ih_d two NN GEMMs.
// Deterministic: no atomics anywhere.  B <= 32 per call (the skinny kernels' dialogue tile); the Python side chunks larger batches.
#include "common.h"

namespace ganffn {

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

struct LstmGateDir {
    const float* G;        // [B x 4H] pre-activations of this step (everything added: xg, recurrent product, both biases)
    const float* c_prev;   // [B x H] (null at the first step: zeros)
    float* c;              // [B x H]
    float* h_out;          // h_t into out[t][:, d H .. (d + 1) H): leading dimension 2H
    float* gates;          // [B x 4H] activated i | f | g | o, kept for the backward
};
struct LstmGateArgs { LstmGateDir d[2]; int B, H; };

__global__ __launch_bounds__(256) void lstm_gate_fwd_kernel(LstmGateArgs a) {
    const LstmGateDir& q = a.d[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.B * a.H) return;
    const int b = idx / a.H, j = idx - b * a.H, H = a.H;
    const float* G = q.G + (size_t)b * 4 * H;
    const float i = sigmoidf_(G[j]), f = sigmoidf_(G[H + j]), g = tanhf(G[2 * H + j]), o = sigmoidf_(G[3 * H + j]);
    const float cp = q.c_prev ? q.c_prev[(size_t)b * H + j] : 0.f;
    const float c = f * cp + i * g;
    q.c[(size_t)b * H + j] = c;
    q.h_out[(size_t)b * 2 * H + j] = o * tanhf(c);
    float* S = q.gates + (size_t)b * 4 * H;
    S[j] = i; S[H + j] = f; S[2 * H + j] = g; S[3 * H + j] = o;
}

struct LstmGateBwdDir {
    const float* dh; int ld_dh;   // gradient wrt h_t: d_out[t]'s half (ld 2H) at the first backward step, else the product's output (ld H)
    float* dc;                    // [B x H] in: dL/dc_t from the later step (zeros at the first backward step: dc_zero), out: dL/dc_{t-1}
    int dc_zero;
    const float* gates;           // activated i | f | g | o of this step
    const float* c; const float* c_prev;   // c_t, c_{t-1} (null: zeros)
    float* dG;                    // [B x 4H] gradient wrt the pre-activations of this step
};
struct LstmGateBwdArgs { LstmGateBwdDir d[2]; int B, H; };

__global__ __launch_bounds__(256) void lstm_gate_bwd_kernel(LstmGateBwdArgs a) {
    const LstmGateBwdDir& q = a.d[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.B * a.H) return;
    const int b = idx / a.H, j = idx - b * a.H, H = a.H;
    const float* S = q.gates + (size_t)b * 4 * H;
    const float i = S[j], f = S[H + j], g = S[2 * H + j], o = S[3 * H + j];
    const float tc = tanhf(q.c[(size_t)b * H + j]);
    const float cp = q.c_prev ? q.c_prev[(size_t)b * H + j] : 0.f;
    const float dh = q.dh[(size_t)b * q.ld_dh + j];
    const float dc = dh * o * (1.f - tc * tc) + (q.dc_zero ? 0.f : q.dc[(size_t)b * H + j]);
    float* dG = q.dG + (size_t)b * 4 * H;
    dG[j] = dc * g * i * (1.f - i);
    dG[H + j] = dc * cp * f * (1.f - f);
    dG[2 * H + j] = dc * i * (1.f - g * g);
    dG[3 * H + j] = dh * tc * o * (1.f - o);
    q.dc[(size_t)b * H + j] = dc * f;
}

// out[n] += sum over rows of X[row][n], rows in order (the two bias gradients of a direction are this same sum)
__global__ __launch_bounds__(256) void lstm_colsum2_kernel(const float* __restrict__ X, int rows, int N, float* __restrict__ o1, float* __restrict__ o2) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};                 // four interleaved chains, joined in a fixed order
    int r = 0;
    for (; r + 4 <= rows; r += 4) {
        s[0] += X[(size_t)r * N + n]; s[1] += X[(size_t)(r + 1) * N + n]; s[2] += X[(size_t)(r + 2) * N + n]; s[3] += X[(size_t)(r + 3) * N + n];
    }
    for (; r < rows; ++r) s[0] += X[(size_t)r * N + n];
    const float t = (s[0] + s[1]) + (s[2] + s[3]);
    if (o1) o1[n] += t;
    if (o2) o2[n] += t;
}

struct LstmOff { int64_t gates, c, total; };             // saved: gates [2][T x 4H] | c [2][T x H]
LstmOff lstm_saved(const ganffn_lstm_cfg* c) {
    const int64_t T = (int64_t)c->S * c->B, H = c->H;
    LstmOff o;
    o.gates = 0; o.c = 2 * T * 4 * H; o.total = o.c + 2 * T * H;
    return o;
}
// workspace: forward xg [2][T x 4H] | G [2][B x 4H];  backward dG [2][T x 4H] | dh [2][B x H] | dc [2][B x H] | TN partial slabs
int64_t lstm_ws(const ganffn_lstm_cfg* c) {
    const int64_t T = (int64_t)c->S * c->B, H = c->H, B = c->B;
    const int64_t fwd = 2 * T * 4 * H + 2 * B * 4 * H;
    const int64_t part = gemm_tn_part_floats(4 * c->H, c->In > c->H ? c->In : c->H, (int)T);
    const int64_t bwd = 2 * T * 4 * H + 4 * B * H + part + 64;
    return (fwd > bwd ? fwd : bwd) + 64;
}

int check_lstm(const ganffn_lstm_cfg* c) {
    GF_CHECK_ARG(c, "null lstm cfg");
    GF_CHECK_ARG(c->S >= 1 && c->B >= 1 && c->B <= 32, "lstm: S=%d B=%d (B <= 32 per call)", c->S, c->B);
    GF_CHECK_ARG(c->In >= 4 && (c->In & 3) == 0 && c->H >= 4 && (c->H & 3) == 0, "lstm: In=%d H=%d must be multiples of 4", c->In, c->H);
    return 0;
}

}  // namespace
}  // namespace ganffn

using namespace ganffn;

extern "C" int64_t ganffn_lstm_saved_floats(const ganffn_lstm_cfg* c) { return check_lstm(c) ? -1 : lstm_saved(c).total; }
extern "C" int64_t ganffn_lstm_workspace_floats(const ganffn_lstm_cfg* c) { return check_lstm(c) ? -1 : lstm_ws(c); }

extern "C" int ganffn_lstm_layer_fwd(const ganffn_lstm_cfg* c, const float* x, const float* const* w_ih, const float* const* w_hh,
                                     const float* const* b_ih, const float* const* b_hh, float* out, float* saved, float* workspace,
                                     void* stream) {
    GF_TRY(check_lstm(c));
    GF_CHECK_ARG(x && w_ih && w_hh && b_ih && b_hh && out && saved && workspace, "lstm_layer_fwd: null pointer");
    GF_CHECK_ARG(aligned16(x) && aligned16(out) && aligned16(saved) && aligned16(workspace), "lstm_layer_fwd: buffers must be 16-byte aligned");
    for (int d = 0; d < 2; ++d)
        GF_CHECK_ARG(w_ih[d] && w_hh[d] && b_ih[d] && b_hh[d] && aligned16(w_ih[d]) && aligned16(w_hh[d]), "lstm_layer_fwd: direction %d: null / unaligned weights", d);
    hipStream_t st = (hipStream_t)stream;
    const int S = c->S, B = c->B, In = c->In, H = c->H;
    const int64_t T = (int64_t)S * B;
    const LstmOff so = lstm_saved(c);
    float* xg = workspace;                          // [2][T x 4H]
    float* G = xg + 2 * T * 4 * H;                  // [2][B x 4H]
    for (int d = 0; d < 2; ++d) {
        EpiArgs e;
        e.bias = b_ih[d];
        GF_TRY(launch_gemm_nt(x, In, w_ih[d], In, xg + d * T * 4 * H, 4 * H, (int)T, 4 * H, In, EPI_NONE, e, st));
    }
    const dim3 ggrid((B * H + 255) / 256, 2);
    for (int t = 0; t < S; ++t) {
        SkinnyGroup sg;
        LstmGateArgs ga;
        ga.B = B; ga.H = H;
        for (int d = 0; d < 2; ++d) {
            const int64_t tm = d == 0 ? t : S - 1 - t, tp = d == 0 ? tm - 1 : tm + 1;      // this step's / the previous step's time index
            float* Gd = G + (int64_t)d * B * 4 * H;
            float* cS = saved + so.c + d * T * H;
            float* gS = saved + so.gates + d * T * 4 * H;
            const float* xgt = xg + d * T * 4 * H + tm * B * 4 * H;
            // (first step: h_{-1} = 0 — the product runs on this step's own, still unwritten, c slot, zeroed below)
            sg.p[d] = SkinnyProb{t == 0 ? cS + tm * B * H : out + tp * B * 2 * H + d * H, t == 0 ? H : 2 * H, w_hh[d], H, xgt, 4 * H, nullptr,
                                 b_hh[d], Gd, 4 * H, B, 4 * H, H};
            ga.d[d] = LstmGateDir{Gd, t == 0 ? nullptr : cS + tp * B * H, cS + tm * B * H, out + tm * B * 2 * H + d * H, gS + tm * B * 4 * H};
        }
        if (t == 0) {
            // the first step's h_prev operand: this step's own (not yet written) c slot, zeroed — h_{-1} = 0 without a special kernel
            for (int d = 0; d < 2; ++d) {
                const int64_t tm = d == 0 ? 0 : S - 1;
                GF_HIP(hipMemsetAsync(saved + so.c + d * T * H + tm * B * H, 0, (size_t)B * H * sizeof(float), st));
            }
        }
        GF_TRY(launch_skinny(sg, 2, false, st));
        hipLaunchKernelGGL(lstm_gate_fwd_kernel, ggrid, dim3(256), 0, st, ga);
        GF_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int ganffn_lstm_layer_bwd(const ganffn_lstm_cfg* c, const float* d_out, const float* x, const float* out,
                                     const float* const* w_ih, const float* const* w_hh, float* dx, float* const* gw_ih,
                                     float* const* gw_hh, float* const* gb_ih, float* const* gb_hh, const float* saved,
                                     float* workspace, void* stream) {
    GF_TRY(check_lstm(c));
    GF_CHECK_ARG(d_out && x && out && w_ih && w_hh && saved && workspace, "lstm_layer_bwd: null pointer");
    GF_CHECK_ARG(aligned16(d_out) && aligned16(x) && aligned16(out) && aligned16(saved) && aligned16(workspace) && (!dx || aligned16(dx)),
                 "lstm_layer_bwd: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int S = c->S, B = c->B, In = c->In, H = c->H;
    const int64_t T = (int64_t)S * B;
    const LstmOff so = lstm_saved(c);
    float* dG = workspace;                          // [2][T x 4H]
    float* dh = dG + 2 * T * 4 * H;                 // [2][B x H]
    float* dc = dh + 2 * B * H;                     // [2][B x H]
    float* part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(dc + 2 * B * H) + 15) & ~(uintptr_t)15);
    const long part_floats = gemm_tn_part_floats(4 * H, In > H ? In : H, (int)T);
    const dim3 ggrid((B * H + 255) / 256, 2);
    for (int t = S - 1; t >= 0; --t) {              // step index of the forward loop, walked backwards
        LstmGateBwdArgs gb;
        gb.B = B; gb.H = H;
        for (int d = 0; d < 2; ++d) {
            const int64_t tm = d == 0 ? t : S - 1 - t, tp = d == 0 ? tm - 1 : tm + 1;
            const float* cS = saved + so.c + d * T * H;
            const float* gS = saved + so.gates + d * T * 4 * H;
            const bool first = t == S - 1;          // first backward step: no later step feeds dh / dc
            gb.d[d] = LstmGateBwdDir{first ? d_out + tm * B * 2 * H + d * H : dh + (int64_t)d * B * H, first ? 2 * H : H, dc + (int64_t)d * B * H,
                                     first ? 1 : 0, gS + tm * B * 4 * H, cS + tm * B * H, t == 0 ? nullptr : cS + tp * B * H,
                                     dG + d * T * 4 * H + tm * B * 4 * H};
        }
        hipLaunchKernelGGL(lstm_gate_bwd_kernel, ggrid, dim3(256), 0, st, gb);
        GF_LAUNCH_CHECK();
        if (t > 0) {
            // dh_{t-1} = d_out[t-1]'s half + dG_t W_hh   ([B x 4H] x [4H x H], the weight row-wise as stored)
            SkinnyGroup sg;
            for (int d = 0; d < 2; ++d) {
                const int64_t tm = d == 0 ? t : S - 1 - t, tp = d == 0 ? tm - 1 : tm + 1;
                sg.p[d] = SkinnyProb{dG + d * T * 4 * H + tm * B * 4 * H, 4 * H, w_hh[d], H, d_out + tp * B * 2 * H + d * H, 2 * H, nullptr, nullptr,
                                     dh + (int64_t)d * B * H, H, B, H, 4 * H};
            }
            GF_TRY(launch_skinny(sg, 2, true, st));
        }
    }
    // deferred weight gradients over all tokens (accumulated: the caller zeroes), bias gradients, input gradient
    for (int d = 0; d < 2; ++d) {
        const float* dGd = dG + d * T * 4 * H;
        if (gw_ih && gw_ih[d]) GF_TRY(launch_gemm_tn_acc(dGd, 4 * H, x, In, gw_ih[d], In, nullptr, 4 * H, In, (int)T, st, part, part_floats));
        if (gw_hh && gw_hh[d] && S > 1) {
            // h_prev of time tm: forward direction out[tm - 1], reverse direction out[tm + 1] (zero at the direction's first step)
            const float* dGs = d == 0 ? dGd + (int64_t)B * 4 * H : dGd;
            const float* hp = d == 0 ? out + d * H : out + (int64_t)B * 2 * H + d * H;
            GF_TRY(launch_gemm_tn_acc(dGs, 4 * H, hp, 2 * H, gw_hh[d], H, nullptr, 4 * H, H, (int)(T - B), st, part, part_floats));
        }
        float* b1 = gb_ih ? gb_ih[d] : nullptr;
        float* b2 = gb_hh ? gb_hh[d] : nullptr;
        if (b1 || b2) {
            hipLaunchKernelGGL(lstm_colsum2_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, st, dGd, (int)T, 4 * H, b1, b2);
            GF_LAUNCH_CHECK();
        }
    }
    if (dx) {
        EpiArgs e0;
        GF_TRY(launch_gemm_nn(dG, 4 * H, w_ih[0], In, dx, In, (int)T, In, 4 * H, EPI_NONE, e0, st));
        EpiArgs e1;
        e1.aux_in = dx;                              // + the reverse direction's share (each element read and written by one thread)
        GF_TRY(launch_gemm_nn(dG + T * 4 * H, 4 * H, w_ih[1], In, dx, In, (int)T, In, 4 * H, EPI_NONE, e1, st));
    }
    return 0;
}
