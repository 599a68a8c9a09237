#!/usr/bin/env python3
"""bench.py — utterances/sec per GAN train step (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 60 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full batch of train_GAN's inner loop (/root/reference/train_IEMOCAP.py:322-382): the 12
sub-steps (6x train_disc, 6x train_gen) with train-mode dropout, BCE, backward and Adam, on a synthetic
IEMOCAP-schema batch of 32 dialogues per GPU padded to S = 94 (BASELINE.json configs[1]).  Weak scaling:
every rank owns its own 32 dialogues; gradients are all-reduced (RCCL) per sub-step.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
L2_SERVED_GBS = 16800.0         # same guide, "Indexed rows: gather into LDS": rows served from the XCDs' L2s, 16.8-18.8 TB/s chip-wide (lower bound)
FP32_MFMA_PEAK = 157.3e12      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
# algorithmic FLOPs of one iteration per padded token at S = 94: fwd 409.5 + bwd 544.8 MFLOP
FLOP_PER_TOKEN_ITER = 954.3e6


def flops_per_token(S, config="iemocap"):
    """reference-equivalent FLOPs per padded token per iteration (fwd + 2x bwd), SURVEY.md §8d formulas."""
    F = 2048

    def layer(E):
        return 8 * E * E + 4 * S * E + 4 * E * F
    D = 8 * layer(100) + 2 * (100 * 64 + 64 * 16 + 16)
    if config == "meld":
        # bi-modal schedule (4 sub-steps): per train_disc 2 D fwd + 1 G fwd, 2 D bwd; per train_gen G fwd + D fwd, D bwd + G bwd
        G600 = 8 * layer(600) + 2 * (600 * 1024 + 1024 * 100)
        G300 = 8 * layer(300) + 2 * (300 * 512 + 512 * 100)
        OBJ = 2 * 600 * 100 + 2 * 300 * 100          # one `object` pass of each discriminator per iteration
        fwd = 6 * D + 2 * G600 + 2 * G300 + OBJ
        bwd = 2 * (6 * D + G600 + G300 + OBJ)
        return fwd + bwd
    G100 = 8 * layer(100) + 2 * (100 * 512 + 512 * 100)
    G512 = 8 * layer(512) + 2 * (512 * 1024 + 1024 * 100)
    OBJ = 2 * 512 * 100
    fwd = 18 * D + 8 * G100 + 4 * G512 + 2 * OBJ
    bwd = 2 * (18 * D + 4 * G100 + 2 * G512 + 2 * OBJ)
    return fwd + bwd


def frozen_wgrad_flops_per_token(config="iemocap"):
    """FLOPs per padded token per iteration the reference spends on the frozen discriminator's weight gradients inside
    train_gen (computed by autograd, zeroed unused: train_IEMOCAP.py:216,245) — the engine does not execute them.
    6 (iemocap) / 2 (meld) discriminator backward passes: 8 layers x 2 x (linear1 + linear2 + out-proj + in-proj weights) + head"""
    per_pass = 8 * 2.0 * (100 * 2048 * 2 + 100 * 100 + 300 * 100) + 2.0 * (100 * 64 + 64 * 16 + 16)
    return (2 if config == "meld" else 6) * per_pass


def wgrad_groups(S, B, config="iemocap"):
    """The dominant kernel `gemm_tn_grouped_kernel` = the deferred weight-gradient GEMMs of one encoder backward pass
    (8 layers x {linear2, linear1, out_proj, in_proj}: dW[M x N] += dY^T[M x K] X[K x N], K = tokens; one owner workgroup
    per output tile over the whole token range adds in place — no atomics —, bias gradients folded in) in one launch.  One iteration issues 6 launches for the discriminators' batched
    [real | fake] pass (2B dialogues, d=100), 4 for the 100-d generators and 2 for the 512-d generator.
    Returns [(launches per iteration, [(M, N, K)] x 32)]."""
    T1, T2 = S * B, S * 2 * B
    e100 = [(100, 2048), (2048, 100), (100, 100), (300, 100)]
    e512 = [(512, 2048), (2048, 512), (512, 512), (1536, 512)]
    if config == "meld":
        e600 = [(600, 2048), (2048, 600), (600, 600), (1800, 600)]
        e300 = [(300, 2048), (2048, 300), (300, 300), (900, 300)]
        return [(2, [(m, n, T2) for _ in range(8) for (m, n) in e100]),
                (1, [(m, n, T1) for _ in range(8) for (m, n) in e600]),
                (1, [(m, n, T1) for _ in range(8) for (m, n) in e300])]
    return [(6, [(m, n, T2) for _ in range(8) for (m, n) in e100]),
            (4, [(m, n, T1) for _ in range(8) for (m, n) in e100]),
            (2, [(m, n, T1) for _ in range(8) for (m, n) in e512])]


TRAFFIC_FILE = "profiles/r05_wgrad_traffic.json"
GEMM_TRAFFIC_FILE = "profiles/r05_gemm_traffic.json"


def wgrad_algorithmic_bytes(S, B, config="iemocap"):
    """average compulsory bytes of one launch: every dY and X operand read once, every dW (and db) written once"""
    tot, n = 0.0, 0
    for cnt, probs in wgrad_groups(S, B, config):
        b = sum(4.0 * (k * m + k * n_ + m * n_ + m) for (m, n_, k) in probs)
        tot += cnt * b
        n += cnt
    return tot / n


def generic_gemm_algorithmic_bytes(S, B):
    """average compulsory bytes of one launch of time_generic_gemm's mix: both operands read once, the output written once
    (the mask epilogue also reads the saved activation)"""
    T, E, F = S * B, 512, 2048
    mix = [(32, T, 3 * E, E, 0), (32, T, E, E, 0), (32, T, F, E, 0), (32, T, E, F, 0), (16, T, F, E, 1), (16, T, E, F, 0), (14, T, E, 3 * E, 0),
           (16, T, E, E, 0)]
    tot = sum(c * 4.0 * (M * K + N * K + M * N * (1 + aux)) for (c, M, N, K, aux) in mix)
    return tot / sum(m[0] for m in mix)


def committed_traffic(S, B, which=None):
    """HBM-side bytes per launch of a kernel family from the committed PMC passes (tools/traffic_pmc.sh), or None
    when the file is absent or was taken at another problem size"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), which or TRAFFIC_FILE)
    try:
        d = json.load(open(path))
        if d.get("seq_len") == S and d.get("dialogues_per_gpu") == B:
            return d["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def _ramp(one_iteration, seconds=0.5):
    """replay the launch mix until `seconds` of wall time have passed: the shader clock idles at ~100 MHz and takes ~0.4 s of
    load to reach its sustained 2.39 GHz (tools/lab/clock_wgrad.py, profiles/r05_ab_logs.txt section 9); a three-replay
    measurement taken cold reads ~10 % slow (weight-gradient family 541 against 485 us per launch).  The training step runs
    with the clock up (bench pre-roll), so its kernels are priced in that state too."""
    t0 = time.time()
    one_iteration()
    torch.cuda.synchronize()
    while time.time() - t0 < seconds:
        one_iteration()
        torch.cuda.synchronize()


def time_dominant_kernel(S, B, reps=3, config="iemocap"):
    """Live HIP-event timing, on the stream the kernel is launched on (torch's current stream), of one iteration's
    launches of the dominant kernel (see wgrad_groups), replayed back to back in isolation.
    Returns (average seconds per launch, average algorithmic flops per launch, launches per iteration)."""
    from gan_ffn_amd import _lib, ops
    groups = wgrad_groups(S, B, config)
    st = ops._stream()
    calls = []
    keep = []
    for cnt, probs in groups:
        n = len(probs)
        bufs = {}
        for (M, N, K) in probs:
            if (M, N, K) not in bufs:
                bufs[(M, N, K)] = (torch.rand(K, M, device="cuda") - 0.5, torch.rand(K, N, device="cuda") - 0.5)
        outs = [(torch.zeros(M, N, device="cuda"), torch.zeros(M, device="cuda")) for (M, N, K) in probs]
        keep.append((bufs, outs))
        PA = (C.c_void_p * n)(*[bufs[p][0].data_ptr() for p in probs])
        PB = (C.c_void_p * n)(*[bufs[p][1].data_ptr() for p in probs])
        PC = (C.c_void_p * n)(*[o[0].data_ptr() for o in outs])
        PS = (C.c_void_p * n)(*[o[1].data_ptr() for o in outs])
        Ms = (C.c_int * n)(*[p[0] for p in probs])
        Ns = (C.c_int * n)(*[p[1] for p in probs])
        Ks = (C.c_int * n)(*[p[2] for p in probs])
        flops = sum(2.0 * m * n_ * k for (m, n_, k) in probs)
        calls.append((cnt, (n, PA, PB, PC, PS, Ms, Ns, Ks), flops))

    nws = int(_lib.load().ganffn_gemm_tn_grouped_workspace_floats())
    ws = torch.empty(nws, device="cuda")         # as inside the encoder backward: narrow groups split their token range

    def one_iteration():
        for cnt, a, _ in calls:
            for _i in range(cnt):
                _lib.call("ganffn_gemm_tn_grouped", a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], ops._ptr(ws), nws, st)
    _ramp(one_iteration)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        one_iteration()
    e1.record()
    torch.cuda.synchronize()
    n = sum(c[0] for c in calls)
    flops = sum(c[0] * c[2] for c in calls) / n
    return e0.elapsed_time(e1) * 1e-3 / (reps * n), flops, n


def time_k100_family(S, B, reps=3):
    """Live HIP-event timing (launch stream = torch's current stream) of the K = 100 -> 2048 family `gemm_wres_kernel`, one
    iteration's launch mix exactly as the encoder stack issues it (measurement hook ganffn_ffn_k100_hook):
    linear1 forward `<0,1,100>` ([T x 100] x [2048 x 100]^T + bias + ReLU + dropout, writing the 1-bit ReLU pattern when the
    pass is saved for a backward): at T1 = S*B 32 train-mode saved launches (4 generator passes x 8 layers), 48 eval-mode
    saved (6 frozen-discriminator passes) and 32 eval-mode unsaved (the generators inside train_disc); at T2 = 2*S*B 48
    train-mode saved (the six [real | fake] discriminator passes);
    linear2 dgrad `<1,3,100>` ([T x 100] x [100 x 2048], masked by the pattern bits): 80 at T1, 48 at T2.
    Returns (avg seconds per launch, avg algorithmic flops per launch, launches)."""
    from gan_ffn_amd import _lib, ops
    st, P = ops._stream(), ops._ptr
    E, F = 100, 2048
    rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
    calls = []
    for T, mix in ((S * B, ((0, 1, 1, 32), (0, 0, 1, 48), (0, 0, 0, 32), (1, 1, 1, 80))), (2 * S * B, ((0, 1, 1, 48), (1, 1, 1, 48)))):
        x = torch.rand(T, E, device="cuda") - 0.5
        w1, b1 = (torch.rand(F, E, device="cuda") - 0.5) * 0.2, torch.zeros(F, device="cuda")
        w2 = (torch.rand(E, F, device="cuda") - 0.5) * 0.2
        h = torch.empty(T, F, device="cuda")
        hmask = torch.zeros(((T + 31) // 32) * F, device="cuda")
        for which, train, saved, cnt in mix:
            def fn(which=which, train=train, saved=saved, x=x, w1=w1, b1=b1, w2=w2, h=h, hmask=hmask, T=T):
                return lambda: _lib.call("ganffn_ffn_k100_hook", which, P(x), P(w2 if which else w1), None if which else P(b1), P(h),
                                         P(hmask) if saved else None, None, T, C.c_float(0.1), C.c_uint32(18), P(rng), C.c_uint64(0), train, st)
            calls.append((cnt, fn(), 2.0 * T * E * F))
    return _time_mix(calls, reps)


def time_n100_kernel(S, B, reps=3):
    """Live HIP-event timing (launch stream) of the [T x 2048] x [2048 x 100] family `gemm_n100_kernel` (csrc/gemm_n100.hip):
    linear2 forward (weight rows of K; 112 launches at T = S*B and 48 at T = 2*S*B per iteration) and the linear1 dgrad
    (K-major weight; 80 and 48).  Returns (avg seconds per launch, avg algorithmic flops per launch, launches)."""
    from gan_ffn_amd import _lib, ops
    st = ops._stream()
    K = 2048
    calls = []
    for T, c_nt, c_km in ((S * B, 112, 80), (2 * S * B, 48, 48)):
        A = torch.rand(T, K, device="cuda") - 0.5
        Wn, Wk = (torch.rand(100, K, device="cuda") - 0.5) * 0.05, (torch.rand(K, 100, device="cuda") - 0.5) * 0.05
        b, slabs, n = torch.zeros(100, device="cuda"), torch.empty(16, T, 100, device="cuda"), C.c_int(0)
        calls.append((c_nt, T, (A, Wn, 0, b, slabs, n)))
        calls.append((c_km, T, (A, Wk, 1, None, slabs, n)))

    def one_iteration():
        for cnt, T, (A, W, km, b, slabs, n) in calls:
            for _ in range(cnt):
                _lib.call("ganffn_gemm_n100", ops._ptr(A), ops._ptr(W), km, ops._ptr(b), ops._ptr(slabs), C.c_int64(T * 100), T, K, 16,
                          C.byref(n), st)
    _ramp(one_iteration)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        one_iteration()
    e1.record()
    torch.cuda.synchronize()
    nl = sum(c[0] for c in calls)
    flops = sum(c[0] * 2.0 * c[1] * 100 * K for c in calls) / nl
    return e0.elapsed_time(e1) * 1e-3 / (reps * nl), flops, nl


def _time_mix(calls, reps=3):
    """HIP-event timing (events recorded on the launch stream = torch's current stream) of `reps` replays of one
    iteration's launch mix: calls = [(count, fn, algorithmic work per launch)].  -> (avg s per launch, avg work per launch, launches)"""
    def one_iteration():
        for cnt, fn, _ in calls:
            for _i in range(cnt):
                fn()
    _ramp(one_iteration)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        one_iteration()
    e1.record()
    torch.cuda.synchronize()
    n = sum(c[0] for c in calls)
    return e0.elapsed_time(e1) * 1e-3 / (reps * n), sum(c[0] * c[2] for c in calls) / n, n


def time_generic_gemm(S, B, reps=3):
    """Live timing of the generic 64 x 64-tile GEMM family (`gemm_kernel<NT|NN, ...>`): one iteration's launch mix of the
    d_model-512 generator — 4 forward stack passes (in-proj, out-proj, linear1 with bias + ReLU + dropout on the two
    train-mode passes, linear2 as split-K slabs) and 2 backward passes (linear2 dgrad with the ReLU/dropout mask, linear1 and
    in-proj dgrad as split-K slabs, out-proj dgrad), 8 layers each — through the measurement hook ganffn_gemm_hook, which
    launches the kernel exactly as the encoder stack does.  (The generator heads and `object`, ~3 % of the family's
    FLOPs, are not replayed.)"""
    from gan_ffn_amd import _lib, ops
    st, P = ops._stream(), ops._ptr
    T, E, F = S * B, 512, 2048
    rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
    r = lambda *sh: torch.rand(*sh, device="cuda") - 0.5
    x, xq, xf = r(T, E), r(T, 3 * E), r(T, F)
    hsave = torch.relu(r(T, F))
    w_in, w_out, w1, w2 = r(3 * E, E) * 0.1, r(E, E) * 0.1, r(F, E) * 0.1, r(E, F) * 0.1
    b3, b1, bE, bF = torch.zeros(3 * E, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(F, device="cuda")
    o3, oE, oF, slabs = torch.empty(T, 3 * E, device="cuda"), torch.empty(T, E, device="cuda"), torch.empty(T, F, device="cuda"), torch.empty(8, T, E, device="cuda")
    n = C.c_int(0)

    def hook(mode, epi, A, W, bias, aux, out, M, N, K, train, max_slabs):
        return lambda: _lib.call("ganffn_gemm_hook", mode, epi, P(A), P(W), P(bias) if bias is not None else None,
                                 P(aux) if aux is not None else None, P(out), C.c_int64(M * N), M, N, K, C.c_float(0.1), C.c_uint32(18),
                                 P(rng), C.c_uint64(0), train, max_slabs, C.byref(n), st)
    fl = lambda M, N, K: 2.0 * M * N * K
    calls = [
        (32, hook(0, 0, x, w_in, b3, None, o3, T, 3 * E, E, 0, 1), fl(T, 3 * E, E)),        # in-proj
        (32, hook(0, 0, x, w_out, b1, None, oE, T, E, E, 0, 1), fl(T, E, E)),              # out-proj
        (16, hook(0, 1, x, w1, bF, None, oF, T, F, E, 1, 1), fl(T, F, E)),                 # linear1, train mode
        (16, hook(0, 1, x, w1, bF, None, oF, T, F, E, 0, 1), fl(T, F, E)),                 # linear1, eval mode
        (32, hook(0, 0, xf, w2, bE, None, slabs, T, E, F, 0, 8), fl(T, E, F)),             # linear2 (split-K slabs)
        (16, hook(1, 3, x, w2, None, hsave, oF, T, F, E, 1, 1), fl(T, F, E)),              # linear2 dgrad: [T x 512] x [512 x 2048], masked
        (16, hook(1, 0, xf, w1, None, None, slabs, T, E, F, 0, 8), fl(T, E, F)),           # linear1 dgrad: [T x 2048] x [2048 x 512]
        (14, hook(1, 0, xq, w_in, None, None, slabs, T, E, 3 * E, 0, 8), fl(T, E, 3 * E)),  # in-proj dgrad: [T x 1536] x [1536 x 512]
        (16, hook(1, 0, x, w_out, None, None, oE, T, E, E, 0, 1), fl(T, E, E)),            # out-proj dgrad
    ]
    return _time_mix(calls, reps)


def time_attention(S, B, reps=3):
    """Live timing of the attention family: one iteration's launch mix — head_dim 10 (d_model 100, 10 heads): forward 112
    launches over B dialogues (32 of them train mode) and 48 over 2B (train), backward 80 over B (32 train) and 48 over 2B
    (train); head_dim 64 (d_model 512, 8 heads): forward 32 (16 train), backward 16 (train) — through the C ABI pair that hands
    the dropout keep words from forward to backward, as the encoder stack does.  -> (avg s, avg algorithmic flops, launches)"""
    from gan_ffn_amd import _lib, ops
    st, P = ops._stream(), ops._ptr
    rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
    lib = _lib.load()
    calls = []
    for (Bx, E, H, nf_tr, nf_ev, nb_tr, nb_ev) in ((B, 100, 10, 32, 80, 32, 48), (2 * B, 100, 10, 48, 0, 48, 0), (B, 512, 8, 16, 16, 16, 0)):
        qkv, do = torch.randn(S, Bx, 3 * E, device="cuda"), torch.randn(S, Bx, E, device="cuda")
        o, lse, dq = torch.empty(S, Bx, E, device="cuda"), torch.zeros(Bx * H, S, device="cuda"), torch.empty(S, Bx, 3 * E, device="cuda")
        keep = torch.zeros(int(lib.ganffn_attention_keep_words(Bx, H)), dtype=torch.int32, device="cuda")
        ffl = 4.0 * S * E * S * Bx
        def fwd(p, qkv=qkv, o=o, lse=lse, keep=keep, Bx=Bx, E=E, H=H):
            return lambda: _lib.call("ganffn_attention_fwd_keep", P(qkv), P(o), P(lse), P(keep), S, Bx, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
        def bwd(p, qkv=qkv, o=o, lse=lse, keep=keep, do=do, dq=dq, Bx=Bx, E=E, H=H):
            return lambda: _lib.call("ganffn_attention_bwd_keep", P(qkv), P(o), P(lse), P(do), P(keep), P(dq), S, Bx, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
        fwd(0.1)()                      # (the backward reads the forward's output, log-sum-exp and keep words)
        for cnt, fn, w in ((nf_tr, fwd(0.1), ffl), (nf_ev, fwd(0.0), ffl), (nb_tr, bwd(0.1), 2.5 * ffl), (nb_ev, bwd(0.0), 2.5 * ffl)):
            if cnt:
                calls.append((cnt, fn, w))
    return _time_mix(calls, reps)


IN_STEP_FILE = "profiles/r05_bench_streams1_by_launch_shape.txt"


def profile_is_current(rel_path):
    """does the committed rocprofv3 summary carry the hash of the kernel sources now in the tree?  (an in-step
    figure taken on older kernels must not be mixed with live numerators)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import roofline_model as RM
    path = os.path.join(ROOT, rel_path)
    try:
        return RM.summary_sha(path) == RM.csrc_sha16()
    except Exception:
        return False


def in_step_kernel_us(symbol_prefix, grid=None):
    """launch-weighted average duration of the (kernel, grid) rows of the committed single-stream rocprofv3 summary whose
    symbol starts with symbol_prefix (grid: one row only), or None"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), IN_STEP_FILE)
    tot, n = 0.0, 0
    try:
        for line in open(path):
            f = line.split()
            if len(f) >= 5 and f[0].endswith("%") and " ".join(f[4:]).startswith(symbol_prefix) and (grid is None or f[3] == grid):
                tot += float(f[2]) * int(f[1])
                n += int(f[1])
    except Exception:
        pass
    return tot / n if n else None


def in_step_wgrad_us():
    """per logical weight-gradient launch, inside the single-stream step: (tn100_kernel + tn100_reduce_kernel) and
    gemm_tn_grouped_kernel rows of the committed summary, weighted by launches"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), IN_STEP_FILE)
    tot, n = 0.0, 0
    try:
        for line in open(path):
            f = line.split()
            if len(f) >= 5 and f[0].endswith("%"):
                name = " ".join(f[4:])
                if name.startswith("tn100_kernel") or name.startswith("gemm_tn_grouped_kernel"):
                    tot += float(f[2]) * int(f[1]); n += int(f[1])
                elif name.startswith("tn100_reduce_kernel"):
                    tot += float(f[2]) * int(f[1])
    except Exception:
        pass
    return round(tot / n, 1) if n else None


def cpu_baseline(S, B_sample, threads, dropout=True, config="iemocap"):
    """Stock-PyTorch CPU execution (oracle/stock_modules.py: nn.TransformerEncoder stacks, train-mode dropout,
    the reference's sub-step logic) of ONE full 12-sub-step iteration on a bounded sample: B_sample dialogues of
    the same S.  dropout=False: every dropout probability 0 (the CPU run is dominated by bernoulli_ mask generation,
    SURVEY.md §6 — the dropout-free time keeps the speed-up from being overstated).
    Returns (utterances/s, seconds, real utterances)."""
    from gan_ffn_amd import data as D
    from oracle import stock_modules as SM
    from oracle.ganffn_oracle import SCHEDULE
    torch.set_num_threads(threads)
    torch.manual_seed(3407)
    meld = config == "meld"
    gens, discs, opts = SM.build_stock(nets=SM.MELD_NETS if meld else None)
    schedule = [s_ for s_ in SCHEDULE if "visual" not in s_[1:]] if meld else SCHEDULE
    if not dropout:
        for m in list(gens.values()) + list(discs.values()):
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
                if isinstance(mod, torch.nn.MultiheadAttention):
                    mod.dropout = 0.0
    batch = make_batch(config, B_sample, S, 3407, "cpu")
    # tiny warm-up (thread pools, allocator): one D sub-step on 2 dialogues
    wb = {k: batch[k][:, :2].contiguous() for k in gens}
    SM.stock_gan_iteration(gens, discs, opts, wb, schedule[:1])
    t0 = time.perf_counter()
    for i, step in enumerate(schedule):            # one sub-step at a time so progress is visible
        SM.stock_gan_iteration(gens, discs, opts, batch, [step])
        print("[bench] cpu_baseline%s sub-step %d/%d done at %.1f s" % ("" if dropout else " (dropout-free)", i + 1, len(schedule), time.perf_counter() - t0), file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    utts = float(batch["umask"].sum())
    return utts / dt, dt, utts


def make_batch(config, B, S, seed, device):
    """synthetic batch of the named workload (dataset pickles are absent: .MISSING_LARGE_BLOBS)"""
    from gan_ffn_amd import data as D
    if config == "meld":
        # MELD feature widths (text 600, audio 300; no visual), 7 emotion classes (train_MELD.py:139,143)
        return D.synthetic_batch(B=B, S_max=S, seed=seed, device=device, n_classes=7, dims=D.MELD_DIMS, lo=2, mean=10)
    return D.synthetic_batch(B=B, S_max=S, seed=seed, device=device)


def build_workload(config, dev):
    """the networks of the named workload, randomly initialised under the reference's seed"""
    from gan_ffn_amd import engine, model
    if config == "meld":
        torch.manual_seed(3407)
        discs = {"acoustic": model.MELDAudioDiscriminator(100, dropout=0.2), "text": model.MELDTextDiscriminator(100, dropout=0.2)}
        gens = {"acoustic": model.MELDAudioGenerator(100, dropout=0.2), "text": model.MELDTextGenerator(100, dropout=0.2)}
        for d in (gens, discs):
            for k in d:
                d[k] = d[k].to(dev)
        return gens, discs
    return engine.build_networks(device=dev, seed=3407)       # random init of the reference architecture


DRNN_IN_STEP_FILE = "profiles/r05_drnn_by_launch_shape.txt"


def time_skinny_kernel(B, reps=20):
    """the recurrence's skinny product at the shape of configuration 5's BACKWARD step (both cells' two dgrad products in
    both directions in one launch: 8 problems of [B x 1500] x [1500 x 500] on the transposed weights, row-wise reads), timed
    with HIP events on the launch stream.  Returns (avg seconds per launch, algorithmic bytes per launch)."""
    from gan_ffn_amd import _lib, ops
    M, N, K = B, 500, 1500
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(8, N, K, device="cuda")
    C = torch.empty(8, M, N, device="cuda")
    run = lambda: _lib.call("ganffn_drnn_skinny", 0, 8, ops._ptr(A), ops._ptr(W), ops._ptr(C), M, N, K, ops._stream())
    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # launches go to torch's current stream
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, 4.0 * (8 * K * N + 8 * M * K + 8 * M * N)


def drnn_in_step_us(symbol_prefix):
    """launch-weighted average duration of a kernel inside the configuration-5 step, from the committed rocprofv3 summary (or
    None; None too when that summary was taken on other kernel sources than the ones in the tree)"""
    if not profile_is_current(DRNN_IN_STEP_FILE):
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), DRNN_IN_STEP_FILE)
    tot, n = 0.0, 0
    try:
        for line in open(path):
            f = line.split()
            if len(f) >= 5 and f[0].endswith("%") and " ".join(f[4:]).startswith(symbol_prefix):
                tot += float(f[2]) * int(f[1])
                n += int(f[1])
    except Exception:
        pass
    return round(tot / n, 2) if n else None


def run_drnn(args, dev, pg, rank, world):
    """BASELINE.json configs[4] on the GPUs given: GAN_FFN_DialogueRNN (three generators -> sum -> bidirectional
    DialogueRNN -> matching attention -> 6-class head; model.py:1485-1534) forward + MaskedNLLLoss + backward + Adam at
    the reference's batch of 30 dialogues (train_IEMOCAP_DialogueRNN.py:580) on the C-ABI step runner
    (engine.DrnnEngine).  Data parallel = one batch per rank, gradients all-reduced in buckets that overlap the
    generators' backward (no reference counterpart: the reference is single-device)."""
    from gan_ffn_amd import model as M, ops
    torch.manual_seed(3407)
    net = M.GAN_FFN_DialogueRNN(M.AcousticGenerator(100), M.VisualGenerator(100), M.TextGenerator(100), 100, 500, 500, 100, 100,
                                100, n_classes=6, listener_state=False, context_attention="general", dropout_rec=0.1,
                                dropout=0.6).to(dev).train()          # train_IEMOCAP_DialogueRNN.py:556-607, 705-721 defaults
    ops.manual_seed(3407 + 1000 * rank, dev)
    batch = make_batch("iemocap", args.batch, args.seq, 3407 + rank, dev)
    S, B = batch["text"].shape[:2]
    from gan_ffn_amd import engine
    if pg is not None:
        import torch.distributed as dist
        for p in net.parameters():
            dist.broadcast(p.data, src=0)
    # the step runner on the C ABI (engine.DrnnEngine): no autograd graph, fused Adam (lr 1e-4, weight decay 1e-5,
    # train_IEMOCAP_DialogueRNN.py:746) on flat slabs, bucketed all-reduce
    eng = engine.DrnnEngine(net, lr=1e-4, weight_decay=1e-5, process_group=pg,
                            n_streams=int(os.environ.get("GANFFN_DRNN_STREAMS", "1")))

    def step():
        return eng.step(batch, train=True)[0]

    def sync():
        torch.cuda.synchronize()
        if pg is not None:
            dist.barrier()
            torch.cuda.synchronize()

    n_warm = max(args.warmup, 60)
    for _ in range(n_warm):      # latency-sized kernels: the clocks need about a second of load to settle (a cold run reads 23-35 ms)
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    utts = float(batch["umask"].sum())
    if pg is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        u = torch.tensor([utts], device=dev, dtype=torch.float64)
        dist.all_reduce(u)
        dt, utts = float(t), float(u)
    if rank == 0:
        kt, kbytes = time_skinny_kernel(B)
        in_step = drnn_in_step_us("skinny_nt_kernel<12>")
        print(json.dumps({
            "metric": "utterances/sec per phase-2 train step, IEMOCAP GAN-FFN + DialogueRNN", "value": round(utts * args.steps / dt, 2),
            "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "IEMOCAP GAN-FFN + DialogueRNN (BASELINE.json configs[4]): 3 generators -> bidirectional "
                                   "DialogueRNN (D_g = D_p = 500, D_e = 100, general attention) -> matching attention -> "
                                   "6-class head, forward + MaskedNLLLoss + backward + Adam, batch=%d per GPU fp32 on "
                                   "%dxMI355X" % (B, world),
                       "dialogues_per_gpu": B, "seq_len": S, "parallelism": "dp%d" % world, "launch": "eager",
                       "last_loss": round(float(loss.detach()), 4)},
            "roofline": {"bound": "hbm", "kernel": "skinny_nt_kernel<12> (the four dgrad products of one backward step of the "
                                                   "recurrence, both directions: 8 x ([B x 1500] x [1500 x 500]) on transposed "
                                                   "weights; the skinny products are the kernel family with the largest share of "
                                                   "GPU time in " + DRNN_IN_STEP_FILE + ")",
                         "achieved": round(kbytes / kt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(kbytes / kt / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": round(kbytes), "avg_kernel_us": round(kt * 1e6, 2),
                         "in_step_avg_us": in_step, "frac_in_step": round(kbytes / (in_step * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if in_step else None,
                         "l2_served_peak": L2_SERVED_GBS, "frac_of_l2_served_peak": round(kbytes / kt / 1e9 / L2_SERVED_GBS, 4),
                         "frac_of_l2_served_peak_in_step": round(kbytes / (in_step * 1e-6) / 1e9 / L2_SERVED_GBS, 4) if in_step else None,
                         "how": "HIP events around 20 back-to-back launches on the launch stream (in_step_avg_us: the same "
                                "kernel inside the step, from the committed rocprofv3 summary).  The 24 MB of recurrent weights are "
                                "re-read every step by the same workgroup ids, i.e. each XCD re-reads its own 3 MB eighth out of its "
                                "4 MiB L2: the roof that applies is the L2 -> CU path (MI355X_MICROARCH.md 'Indexed rows': 16.8-18.8 "
                                "TB/s chip-wide for rows served from L2) — frac_of_l2_served_peak; `frac` against the HBM peak is "
                                "what the contract's bound key offers.  Either way the launch is latency-sized (one link of a "
                                "serial chain of ~400)"}}), flush=True)


def host_threads():
    """CPU threads this process may really use: the affinity mask, capped by the cgroup CPU quota (a 1-GPU box exposes
    many cores but grants a share of them).  GANFFN_CPU_THREADS overrides (no built-in cap)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    if os.environ.get("GANFFN_CPU_THREADS"):
        n = int(os.environ["GANFFN_CPU_THREADS"])
    return max(1, n)


# ---------------------------------------------------------------------------------------------------------------------
# Self-launcher (replaces /root/reference/train_IEMOCAP.py:587-593, the nn.DataParallel wrap): `python bench.py --gpus N`
# with no WORLD_SIZE in the environment starts its own N ranks.  The parent makes NO GPU call (it never imports
# gan_ffn_amd, never asks torch.cuda anything): it spawns N children of this same file with RANK / LOCAL_RANK /
# WORLD_SIZE / MASTER_* set, relays their stderr, takes rank 0's JSON line and exits with the children's return code.
# It is also the watchdog: the in-line 3-communicator data-parallel default has never run on more than one rank (no
# multi-GPU node was available to this build), so if the children go silent for too long or fail, the parent ends
# THOSE processes (the exact process groups it started) and starts FRESH children on the next rung of LADDER.
# ---------------------------------------------------------------------------------------------------------------------
LADDER = [
    # (name, environment of the rung, --streams or None = as asked)
    ("inline-3streams", {"GANFFN_DP_MODE": "inline", "GANFFN_COMM_PER_STREAM": "1"}, None),
    ("inline-1stream", {"GANFFN_DP_MODE": "inline", "GANFFN_COMM_PER_STREAM": "0"}, 1),
    ("buckets", {"GANFFN_DP_MODE": "buckets", "GANFFN_COMM_PER_STREAM": "0"}, None),
]


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def rank_environments(n, port, rung_env=None, base=None):
    """the N child environments of one rung: what `python -m torch.distributed.run --nproc-per-node N` would export"""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update(rung_env or {})
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), GANFFN_BENCH_CHILD="1")
        envs.append(e)
    return envs


def _child_argv(argv, streams):
    """the parent's own arguments, with --streams replaced when the rung asks for it (and never --launcher: a child runs)"""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in ("--launcher", "--streams") and (a == "--launcher" or streams is not None):
            skip = True
            continue
        if a.startswith("--launcher=") or (streams is not None and a.startswith("--streams=")):
            continue
        out.append(a)
    if streams is not None:
        out += ["--streams", str(streams)]
    return out


def _end_children(procs, grace=10.0):
    """end exactly the process groups this launcher started (each child is its own session leader)"""
    import signal
    for sig in (signal.SIGTERM, signal.SIGKILL):
        alive = [p_ for p_ in procs if p_.poll() is None]
        if not alive:
            return
        for p_ in alive:
            try:
                os.killpg(p_.pid, sig)
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.time() + grace
        while time.time() < t_end and any(p_.poll() is None for p_ in alive):
            time.sleep(0.2)


def launch(args, argv, cmd=None):
    """parent of `--gpus N`: returns the process exit code.  No GPU call is made here.
    cmd: the child command line (default: this file with the parent's arguments) — tests/test_bench_launcher_cpu.py passes a
    stand-in child to exercise the watchdog and the ladder without a GPU."""
    import subprocess
    import threading
    n = args.gpus
    silence = float(os.environ.get("GANFFN_LAUNCH_SILENCE_S", "300"))     # a fresh box pages torch in for 1-2 min per process
    total_cap = float(os.environ.get("GANFFN_LAUNCH_RUNG_S", "1500"))
    ladder = LADDER if (n > 1 or os.environ.get("GANFFN_FORCE_DIST", "0") == "1") else [("single", {}, None)]
    if os.environ.get("GANFFN_LAUNCH_RUNGS"):                              # e.g. "1,2": start further down (tests, a known-bad box)
        ladder = [LADDER[int(i)] for i in os.environ["GANFFN_LAUNCH_RUNGS"].split(",")]
    if os.environ.get("GANFFN_DP_MODE"):                                   # the caller pinned the mode: one rung, as asked
        ladder = [("as-asked", {}, None)]
    tried = []
    for name, rung_env, streams in ladder:
        port = _free_port()
        envs = rank_environments(n, port, dict(rung_env, GANFFN_BENCH_RUNG=name, GANFFN_BENCH_FALLBACK_FROM=",".join(tried)))
        rung_cmd = (cmd if cmd is not None else [sys.executable, os.path.abspath(__file__)]) + _child_argv(argv, streams)
        print("[bench launcher] rung %r: starting %d ranks (port %d)%s" % (name, n, port, ", after " + ",".join(tried) if tried else ""),
              file=sys.stderr, flush=True)
        procs, last, lines = [], [time.time()], []
        for r in range(n):
            procs.append(subprocess.Popen(rung_cmd, env=envs[r], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                          start_new_session=True, cwd=ROOT))

        def pump(stream, sink, rank, keep):
            for line in stream:
                last[0] = time.time()
                if keep is not None and line.startswith("{"):
                    keep.append(line.strip())
                else:
                    sink.write(line if rank == 0 else "[rank %d] %s" % (rank, line))
                    sink.flush()
        threads = []
        for r, p_ in enumerate(procs):
            threads.append(threading.Thread(target=pump, args=(p_.stdout, sys.stderr, r, lines if r == 0 else None), daemon=True))
            threads.append(threading.Thread(target=pump, args=(p_.stderr, sys.stderr, r, None), daemon=True))
        for t_ in threads:
            t_.start()
        t0, why = time.time(), None
        while True:
            rcs = [p_.poll() for p_ in procs]
            if all(rc is not None for rc in rcs):
                break
            if any(rc not in (None, 0) for rc in rcs):
                why = "rank(s) %s exited with %s" % ([i for i, rc in enumerate(rcs) if rc not in (None, 0)], [rc for rc in rcs if rc not in (None, 0)])
                break
            if time.time() - last[0] > silence:
                why = "no output from any rank for %.0f s" % silence
                break
            if time.time() - t0 > total_cap:
                why = "rung exceeded %.0f s" % total_cap
                break
            time.sleep(0.25)
        if why is not None:
            _end_children(procs)
        for t_ in threads:
            t_.join(timeout=5)
        rcs = [p_.poll() for p_ in procs]
        if why is None and all(rc == 0 for rc in rcs) and lines:
            try:
                d = json.loads(lines[-1])
                d.setdefault("config", {})
                d["config"]["launcher"] = {"spawned_ranks": n, "rung": name, "fallback_from": tried or None,
                                           "how": "bench.py started its own ranks (no external torch.distributed.run)"}
                print(json.dumps(d), flush=True)
            except Exception:
                print(lines[-1], flush=True)
            return 0
        why = why or ("children returned %s%s" % (rcs, "" if lines else " and rank 0 printed no JSON line"))
        print("[bench launcher] rung %r failed: %s" % (name, why), file=sys.stderr, flush=True)
        tried.append(name)
    print("[bench launcher] every rung failed: %s" % tried, file=sys.stderr, flush=True)
    return 1


def supervise(args, argv, cmd=None, coord_dir=None):
    """One rank of an EXTERNAL launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: how the
    driver starts N > 1) — the same watchdog and fallback ladder as launch(), per rank.  The process torchrun started makes NO
    GPU call: it is a supervisor that runs the real rank as a child process, so that a hung collective can be answered by
    ending that child and starting a FRESH one on the next rung.  The N supervisors of a launch (same node: nnodes = 1) agree
    through files in one directory (named after torchrun's agent pid, which all of them share as their parent, and the master
    port): rank 0 publishes a fresh rendezvous port per rung (the agent's own store on MASTER_PORT cannot host a second
    rendezvous), any supervisor whose child dies or goes silent marks the rung failed, every supervisor that sees the mark ends
    its child and moves on; rank 0's JSON line ends the launch."""
    import subprocess
    import threading
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    silence = float(os.environ.get("GANFFN_LAUNCH_SILENCE_S", "300"))
    total_cap = float(os.environ.get("GANFFN_LAUNCH_RUNG_S", "1500"))
    D = coord_dir or os.path.join("/tmp", "ganffn_bench_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))
    os.makedirs(D, exist_ok=True)
    ladder = LADDER
    if os.environ.get("GANFFN_LAUNCH_RUNGS"):
        ladder = [LADDER[int(i)] for i in os.environ["GANFFN_LAUNCH_RUNGS"].split(",")]
    if os.environ.get("GANFFN_DP_MODE"):
        ladder = [("as-asked", {}, None)]
    tried = []

    def wait_file(path, timeout):
        t_end = time.time() + timeout
        while time.time() < t_end:
            if os.path.exists(path):
                return True
            time.sleep(0.1)
        return False

    for k, (name, rung_env, streams) in enumerate(ladder):
        pf = os.path.join(D, "rung%d.port" % k)
        if rank == 0:
            with open(pf + ".tmp", "w") as f:
                f.write(str(_free_port()))
            os.replace(pf + ".tmp", pf)
        if not wait_file(pf, 120):
            print("[bench supervisor %d] rung %r: no port from rank 0" % (rank, name), file=sys.stderr, flush=True)
            return 1
        port = int(open(pf).read())
        e = dict(os.environ)
        e.update(rung_env)
        e.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GANFFN_BENCH_CHILD="1", GANFFN_BENCH_RUNG=name,
                 GANFFN_BENCH_FALLBACK_FROM=",".join(tried), GANFFN_BENCH_SUPERVISED="1",
                 HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        for v in ("TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS"):
            e.pop(v, None)          # the child rendezvouses through its own store on the fresh port, not through the agent's
        rung_cmd = (cmd if cmd is not None else [sys.executable, os.path.abspath(__file__)]) + _child_argv(argv, streams)
        if rank == 0:
            print("[bench supervisor] rung %r: %d ranks, port %d%s" % (name, world, port, ", after " + ",".join(tried) if tried else ""),
                  file=sys.stderr, flush=True)
        proc = subprocess.Popen(rung_cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, cwd=ROOT)
        last, lines = [time.time()], []

        def pump(stream, keep):
            for line in stream:
                last[0] = time.time()
                if keep is not None and line.startswith("{"):
                    keep.append(line.strip())
                else:
                    sys.stderr.write(line if rank == 0 else "[rank %d] %s" % (rank, line))
                    sys.stderr.flush()
        ths = [threading.Thread(target=pump, args=(proc.stdout, lines if rank == 0 else None), daemon=True),
               threading.Thread(target=pump, args=(proc.stderr, None), daemon=True)]
        for t_ in ths:
            t_.start()
        fail, done = os.path.join(D, "rung%d.fail" % k), os.path.join(D, "rung%d.done" % k)
        t0, why, outcome, t_exit = time.time(), None, None, None
        while outcome is None:
            rc = proc.poll()
            if rc == 0 and t_exit is None:
                t_exit = time.time()
            if os.path.exists(done):
                outcome = "done"
            elif os.path.exists(fail):
                outcome = "failed"
            elif rc is not None and rc != 0:
                why, outcome = "rank %d exited with %d" % (rank, rc), "failed"
            elif rc == 0 and rank == 0:
                for t_ in ths:
                    t_.join(timeout=5)
                if lines:
                    outcome = "done"
                else:
                    why, outcome = "rank 0 printed no JSON line", "failed"
            elif rc == 0 and rank != 0 and time.time() - t_exit > 120.0:
                outcome = "done"              # this rank's worker finished cleanly; rank 0 will have had its line
            elif rc is None and time.time() - last[0] > silence:
                why, outcome = "rank %d silent for %.0f s" % (rank, silence), "failed"
            elif time.time() - t0 > total_cap:
                why, outcome = "rung exceeded %.0f s" % total_cap, "failed"
            else:
                time.sleep(0.25)
        if outcome == "failed":
            if why is not None and not os.path.exists(fail):
                try:
                    with open(fail + ".%d" % rank, "w") as f:
                        f.write(why)
                    os.replace(fail + ".%d" % rank, fail)
                except OSError:
                    pass
            _end_children([proc])
            if rank == 0:
                print("[bench supervisor] rung %r failed: %s" % (name, why or open(fail).read()), file=sys.stderr, flush=True)
            tried.append(name)
            continue
        # done
        if rank == 0:
            open(done, "w").write("ok")
            d = None
            try:
                d = json.loads(lines[-1])
                d.setdefault("config", {})
                d["config"]["launcher"] = {"spawned_ranks": world, "rung": name, "fallback_from": tried or None,
                                           "how": "each rank of the external launcher supervises its own worker process (watchdog + ladder)"}
                print(json.dumps(d), flush=True)
            except Exception:
                print(lines[-1], flush=True)
        else:
            t_end = time.time() + 30.0            # rank 0 has its line: a straggler in teardown is given 30 s, then ended
            while proc.poll() is None and time.time() < t_end:
                time.sleep(0.2)
        _end_children([proc], grace=3.0)
        return 0
    if rank == 0:
        print("[bench supervisor] every rung failed: %s" % tried, file=sys.stderr, flush=True)
    return 1


def heartbeat(msg, rank=0, every_rank=False):
    """progress line for the launcher's watchdog (stderr; any rank's output counts as a sign of life)"""
    if rank == 0 or every_rank:
        print("[bench] %s" % msg, file=sys.stderr, flush=True)


def preroll(step, sync, pg, dev, window=5, tol=0.003, cap_s=1.5):
    """internal pre-roll: run the step until two consecutive `window`-iteration blocks agree within `tol` (the clocks of an
    idle MI355X need a few hundred ms of load to settle), at most cap_s seconds; the caller's --warmup comes after it.  With a
    process group rank 0 decides and the others follow (one tiny broadcast per window).  -> iterations run"""
    t_begin, prev, n = time.perf_counter(), None, 0
    while True:
        sync()
        t0 = time.perf_counter()
        for _ in range(window):
            step()
        sync()
        dt = time.perf_counter() - t0
        n += window
        stop = (prev is not None and abs(dt - prev) <= tol * prev) or (time.perf_counter() - t_begin > cap_s)
        if pg is not None:
            import torch.distributed as dist
            flag = torch.tensor([1 if stop else 0], device=dev, dtype=torch.int32)
            dist.broadcast(flag, src=0)
            stop = bool(int(flag))
        if stop:
            return n
        prev = dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 30 untimed + 60 timed iterations (~3 s of GPU time): the clocks of an idle MI355X need a few hundred ms of load
    # to settle — measured on one box: 3 + 20 iterations 34.39 / 34.55 ms per step, 30 + 60: 34.33 / 34.33, 100 + 100: 34.33 / 34.25
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32, help="dialogues per GPU (reference hard-codes 32, train_IEMOCAP.py:603)")
    ap.add_argument("--seq", type=int, default=None, help="padded dialogue length S (default 94, model.py:1437; "
                    "--config meld: 33, MELD's longest dialogue [public, not stated by the reference])")
    ap.add_argument("--config", choices=["iemocap", "meld", "drnn"], default="iemocap",
                    help="iemocap = BASELINE.json configs[1] (the headline metric's workload); meld = configs[2], the "
                         "generic G/D stack at MELD's feature widths on the bi-modal schedule (extension: the reference "
                         "has no GAN path for MELD); drnn = configs[4], the phase-2 GAN-FFN + DialogueRNN "
                         "classifier step at batch 30")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-only", action="store_true",
                    help="time the step and print the line without the isolated kernel replays (roofline / roofline_worst) and "
                         "without the CPU baseline: what tools/gpu_round.sh profiles, so that the rocprofv3 summaries hold the "
                         "step's own launches only")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams running independent sub-steps concurrently")
    ap.add_argument("--cpu-sample-batch", type=int, default=None,
                    help="dialogues of the CPU-baseline sample (default: the whole batch, BASELINE.md §3)")
    ap.add_argument("--launcher", choices=["auto", "spawn", "none"], default="auto",
                    help="auto: when --gpus N > 1 and no WORLD_SIZE is set, this process starts the N ranks itself (and a fallback "
                         "ladder of data-parallel modes if they hang); spawn: always go through that path (also at N = 1); none: never")
    ap.add_argument("--replay-family", choices=["wgrad", "gemm_generic", "attention", "ffn_k100", "ffn_n100"], default=None,
                    help="only replay one kernel family's launch mix (warm-up pass + one timed pass): the command "
                         "tools/traffic_pmc.sh / tools/attention_pmc.sh profile with rocprofv3 --pmc, one counter set per pass")
    args = ap.parse_args()
    if args.seq is None:
        args.seq = 33 if args.config == "meld" else 94
    if args.config == "drnn" and args.batch == 32:
        args.batch = 30                                              # train_IEMOCAP_DialogueRNN.py:580
    if args.cpu_sample_batch is None:
        args.cpu_sample_batch = args.batch
    cfgname = args.config

    # ---- N ranks from one command: the parent launches, the children (WORLD_SIZE set) run.  Nothing above touches the GPU.
    is_child = "WORLD_SIZE" in os.environ or os.environ.get("GANFFN_BENCH_CHILD") == "1"
    if args.launcher == "spawn" or (args.launcher == "auto" and args.gpus > 1 and not is_child):
        if not is_child:
            sys.exit(launch(args, sys.argv[1:]))
    # started by an EXTERNAL launcher with several ranks (torch.distributed.run: how the driver runs N > 1): this process
    # becomes its rank's supervisor — same watchdog, same ladder, agreed between the ranks through files (supervise())
    if args.launcher != "none" and os.environ.get("GANFFN_BENCH_CHILD") != "1" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        sys.exit(supervise(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    heartbeat("rank %d of %d up (pid %d)" % (rank, world, os.getpid()), rank, every_rank=True)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the hot path has no CPU fallback"
    # backend: nccl = RCCL over xGMI, one GPU per rank (the product).  GANFFN_DIST_BACKEND=gloo is a REHEARSAL mode for a one-GPU
    # box: the ranks share the visible GPU(s) and reduce over gloo (RCCL refuses two ranks on one device) — it exercises the
    # launcher and every world > 1 line of this file end to end, and says so in the line (`dist_backend`); never a measurement
    backend = os.environ.get("GANFFN_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    force_dist = os.environ.get("GANFFN_FORCE_DIST", "0") == "1"   # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD
        heartbeat("rank %d: process group up (%s, %d ranks)" % (rank, "RCCL" if backend == "nccl" else backend, dist.get_world_size()), rank, every_rank=True)

    from gan_ffn_amd import _lib, engine, ops
    if args.config == "drnn":
        _lib.load()
        if os.environ.get("GANFFN_FFN_MODE"):
            _lib.load().ganffn_debug_set_ffn_mode(int(os.environ["GANFFN_FFN_MODE"]))
        run_drnn(args, dev, pg, rank, world)
        if pg is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.replay_family:
        _lib.load()
        fn = {"wgrad": lambda S_, B_, reps: time_dominant_kernel(S_, B_, reps=reps, config=cfgname), "gemm_generic": time_generic_gemm,
              "attention": time_attention, "ffn_k100": time_k100_family, "ffn_n100": time_n100_kernel}[args.replay_family]
        kt, kwork, klaunch = fn(args.seq, args.batch, reps=1)
        print(json.dumps({"family": args.replay_family, "avg_kernel_us": kt * 1e6, "launches": klaunch}), flush=True)
        return
    from gan_ffn_amd import data as D
    _lib.load()
    if os.environ.get("GANFFN_FFN_MODE"):
        _lib.load().ganffn_debug_set_ffn_mode(int(os.environ["GANFFN_FFN_MODE"]))

    gens, discs = build_workload(cfgname, dev)
    if pg is not None:
        import torch.distributed as dist
        for d in (gens, discs):
            for m in d.values():
                dist.broadcast(m.slab, src=0)                        # replicated parameters
    ops.manual_seed(3407 + 1000 * rank, dev)                          # rank-offset dropout streams
    batch = make_batch(cfgname, args.batch, args.seq, 3407 + rank, dev)
    S, B = batch["text"].shape[:2]
    use_graph = (not args.no_graph) and pg is None and args.streams == 1
    eng = engine.GanEngine(gens, discs, process_group=pg, use_graph=use_graph, n_streams=args.streams)

    def sync():
        eng.synchronize()
        torch.cuda.synchronize()
        if pg is not None:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    heartbeat("rank %d: engine built (dp_mode %s, %d streams)" % (rank, engine.dp_mode() if pg is not None else "none", eng.n_streams), rank, every_rank=True)
    # internal pre-roll (not counted in --warmup): until two consecutive 5-iteration windows agree within 0.3 %, <= 1.5 s —
    # so that the line does not depend on how many warm-up steps the caller asked for
    n_preroll = 0 if os.environ.get("GANFFN_BENCH_PREROLL", "1") == "0" else preroll(lambda: eng.iteration(batch), sync, pg, dev)
    heartbeat("rank %d: pre-roll %d iterations" % (rank, n_preroll), rank, every_rank=True)
    for _ in range(args.warmup):
        eng.iteration(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.iteration(batch)
    sync()
    dt = time.perf_counter() - t0
    losses = eng.loss_dict()
    heartbeat("rank %d: %d timed steps, %.3f ms per step" % (rank, args.steps, dt / args.steps * 1e3), rank, every_rank=True)
    # second, separately timed block (median over >= 20 device-synchronised iterations): here every iteration
    # is bracketed by a device synchronisation (+ barrier), so consecutive iterations do NOT overlap as they do in the block
    # above — it reads a little slower and is reported beside the headline, never instead of it
    per_iter = []
    for _ in range(0 if args.step_only else max(20, min(args.steps, 60))):
        sync()
        t1 = time.perf_counter()
        eng.iteration(batch)
        sync()
        per_iter.append((time.perf_counter() - t1) * 1e3)
    per_iter.sort()

    utts = torch.tensor([float(batch["umask"].sum()), dt], device=dev, dtype=torch.float64)
    rank_ms = [dt / args.steps * 1e3]
    if pg is not None:
        import torch.distributed as dist
        tmax = utts[1:2].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        usum = utts[0:1].clone()
        dist.all_reduce(usum, op=dist.ReduceOp.SUM)
        allt = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, utts[1:2].clone())
        rank_ms = [float(t_) / args.steps * 1e3 for t_ in allt]
        dt, total_utts = float(tmax), float(usum)
    else:
        total_utts = float(utts[0])
    ms_per_step = dt / args.steps * 1e3
    value = total_utts * args.steps / dt
    if pg is not None:
        import torch.distributed as dist
    dist_info = {"rccl_ranks": dist.get_world_size() if pg is not None else 0,          # ranks the process group really connected (0: none)
                 "dist_backend": (backend if pg is not None else None),                 # "nccl" = RCCL; anything else is a rehearsal, not a measurement
                 "dp_mode": engine.dp_mode() if pg is not None else None,
                 "communicators": len(getattr(eng, "pgs", [None])) if pg is not None else 0,
                 "launcher_rung": os.environ.get("GANFFN_BENCH_RUNG"),
                 "fallback_from": [x for x in os.environ.get("GANFFN_BENCH_FALLBACK_FROM", "").split(",") if x] or None,
                 "ms_per_step_min_max_over_ranks": [round(min(rank_ms), 3), round(max(rank_ms), 3)]}
    if rank == 0:
        print("[bench] gpu: %.3f ms/step, %.1f utterances/s (%s)" % (ms_per_step, value, "hipGraph" if use_graph else "eager"),
              file=sys.stderr, flush=True)

    if rank == 0 and args.step_only:
        print(json.dumps({"metric": "utterances/sec per GAN train step (step only)", "value": round(value, 2), "unit": "utterances/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
                          "config": dict({"workload": cfgname, "streams": eng.n_streams, "preroll_iterations": n_preroll}, **dist_info)}), flush=True)
    elif rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import roofline_model as RM
        fpt = flops_per_token(S, cfgname)
        step_tflops = fpt * S * B * world * args.steps / dt / 1e12
        out = {
            "metric": "utterances/sec per GAN train step, IEMOCAP tri-modal" if cfgname == "iemocap" else
                      "utterances/sec per GAN train step, MELD-dimension bi-modal (extension: no reference GAN path for MELD)",
            "value": round(value, 2), "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("IEMOCAP tri-modal batch=%d per GPU fp32 on %dxMI355X, full G+D step "
                                    "(12 sub-steps: 6 train_disc + 6 train_gen, train-mode dropout, BCE, Adam)" % (B, world))
                       if cfgname == "iemocap" else
                       ("MELD feature dims (text 600, audio 300; train_MELD.py:143, dataloader.py:93-95), 7-class labels, "
                        "batch=%d per GPU fp32 on %dxMI355X: generic G/D stacks (E = 600 / 300, 10 heads) on the bi-modal "
                        "schedule (4 sub-steps: 2 train_disc + 2 train_gen, train-mode dropout, BCE, Adam); EXTENSION — the "
                        "reference has no GAN path for MELD (SURVEY.md §8d)" % (B, world)),
                       "dialogues_per_gpu": B, "seq_len": S, "real_utterances_per_gpu_batch": float(batch["umask"].sum()),
                       "padded_tokens_per_s": round(S * B * world * args.steps / dt, 1),
                       "parallelism": "dp%d" % world, "launch": "hipGraph" if use_graph else "eager", "streams": eng.n_streams,
                       "step_tflops_reference_equivalent": round(step_tflops, 2),
                       "step_frac_of_fp32_mfma_peak": round(step_tflops * 1e12 / (FP32_MFMA_PEAK * world), 4),
                       # what the engine EXECUTES: the reference computes the frozen discriminator's weight gradients in
                       # train_gen and never uses them (train_IEMOCAP.py:216 zeroes them first); the engine skips those 6 passes
                       "step_tflops_executed": round(step_tflops * (1.0 - frozen_wgrad_flops_per_token(cfgname) / fpt), 2),
                       "step_frac_executed": round(step_tflops * (1.0 - frozen_wgrad_flops_per_token(cfgname) / fpt) * 1e12 / (FP32_MFMA_PEAK * world), 4),
                       "preroll_iterations": n_preroll,
                       "median_ms_per_step_synchronised": round(per_iter[len(per_iter) // 2], 3) if per_iter else None,
                       "min_max_ms_per_step_synchronised": [round(per_iter[0], 3), round(per_iter[-1], 3)] if per_iter else None,
                       "synchronised_iterations": len(per_iter),
                       "timing": "ms_per_step / value: EXACTLY --steps iterations between two barrier + device-synchronise points "
                                 "(consecutive iterations overlap on the three streams, as in training); median_ms_per_step_synchronised: "
                                 "a second block, device-synchronised at every iteration boundary (SURVEY.md 8d)",
                       **dist_info,
                       "last_losses": {k: round(v, 4) for k, v in losses.items()}},
        }
        if cfgname != "iemocap":
            # extension workload: the weight-gradient launch (its heaviest family; no committed single-stream profile of it)
            kt, kflop, klaunch = time_dominant_kernel(S, B, config=cfgname)
            out["roofline"] = {"bound": "mfma", "kernel": "the grouped weight-gradient launch (all 32 weight + bias gradients of one encoder "
                                                          "backward pass in one launch; %d launches per iteration)" % klaunch,
                               "achieved": round(kflop / kt / 1e12, 2), "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                               "frac": round(kflop / kt / FP32_MFMA_PEAK, 4), "traffic": None,
                               "algorithmic_bytes_per_launch": round(wgrad_algorithmic_bytes(S, B, cfgname)),
                               "avg_kernel_us": round(kt * 1e6, 2), "avg_gflop_per_launch": round(kflop / 1e9, 4),
                               "how": "HIP events around one iteration's launch mix of this launch replayed in isolation on the launch stream"}
        else:
            # ---- the bench line FOLLOWS the committed single-stream profile: every kernel family with >= 5 % of the step's
            # kernel time is listed with its in-step roofline fraction (tools/roofline_model.py: family table + algorithmic
            # work per iteration), each is also replayed live (HIP events on the launch stream); `roofline` is the family with
            # the LARGEST share, `roofline_worst` the one with the LOWEST fraction.
            full = (S, B) == (94, 32)
            prof_path = os.path.join(ROOT, IN_STEP_FILE)
            current = profile_is_current(IN_STEP_FILE)
            instep = RM.in_step(prof_path, S, B) if (full and os.path.exists(prof_path)) else None
            live = {}
            for key, fn in (("gemm_generic", time_generic_gemm), ("ffn_k100", time_k100_family), ("ffn_n100", time_n100_kernel),
                            ("wgrad", lambda S_, B_: time_dominant_kernel(S_, B_, config=cfgname)), ("attention", time_attention)):
                t_, w_, n_ = fn(S, B)
                live[key] = dict(avg_kernel_us=round(t_ * 1e6, 2), avg_gflop_per_launch=round(w_ / 1e9, 4), launches_replayed=n_,
                                 achieved=round(w_ / t_ / 1e12, 2), frac=round(w_ / t_ / FP32_MFMA_PEAK, 4))
            table = {f["key"]: f for f in RM.family_table(S, B)}
            fams = []
            order = [d["family"] for d in instep["families"]] if instep else list(live)
            for key in order:
                d = dict(next(x for x in instep["families"] if x["family"] == key)) if instep else \
                    dict(family=key, kernel=table[key]["title"], bound=table[key]["bound"])
                if instep:
                    d["in_step_frac"] = d.pop("frac")
                    d["in_step_achieved"] = d.pop("achieved")
                    if not current:
                        d["in_step_stale"] = True        # the committed profile was taken on other kernel sources
                if key in live:
                    d.update(live[key])                  # achieved / frac = the LIVE isolated replay
                    d.update(peak=FP32_MFMA_PEAK / 1e12, unit="TFLOP/s")
                else:
                    d.update(achieved=d.get("in_step_achieved"), frac=d.get("in_step_frac"), avg_kernel_us=None,
                             live="not replayed in isolation (its launches only exist inside an encoder pass): in-step figures only")
                fams.append(d)
            # a stale profile (taken on other kernel sources) ranks nothing: dominant / worst are then chosen among the families
            # replayed live, on their live fractions
            rankable = fams if current or not instep else [d for d in fams if d["family"] in live]
            dom = rankable[0] if current or not instep else max(rankable, key=lambda d: d["avg_kernel_us"] * d["launches_replayed"])
            traffic = committed_traffic(S, B, GEMM_TRAFFIC_FILE if dom["family"] == "gemm_generic" else TRAFFIC_FILE)
            alg_bytes = round(generic_gemm_algorithmic_bytes(S, B)) if dom["family"] == "gemm_generic" else \
                (round(wgrad_algorithmic_bytes(S, B, cfgname)) if dom["family"] == "wgrad" else None)
            traffic_note = None
            if traffic is not None and alg_bytes and dom.get("avg_kernel_us"):
                rate = traffic / (dom["avg_kernel_us"] * 1e-6) / 1e12
                traffic_note = ("%.2fx the algorithmic bytes; %s: %.1f MB per %.1f us launch = %.2f TB/s of the 8 TB/s HBM peak"
                                % (traffic / alg_bytes, "NOT the bound" if rate < 0.5 * HBM_PEAK_GBS / 1e3 else "close to the HBM roof",
                                   traffic / 1e6, dom["avg_kernel_us"], rate)) + \
                    ("; the excess over 1x is the weight operand, which each of the 8 XCDs reads through its own L2"
                     if dom["family"] == "gemm_generic" else "")
            out["roofline"] = {
                "bound": dom["bound"], "kernel": dom["kernel"], "family": dom["family"],
                "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                "traffic": traffic, "traffic_note": traffic_note,
                "traffic_unit": ("bytes per launch, averaged over the replayed launch mix (FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE, "
                                 "separate rocprofv3 --pmc passes: " + (GEMM_TRAFFIC_FILE if dom["family"] == "gemm_generic" else TRAFFIC_FILE) + ")")
                if traffic is not None else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_kernel_us": dom.get("avg_kernel_us"), "avg_gflop_per_launch": dom.get("avg_gflop_per_launch"),
                "share_of_step_kernel_time_pct": dom.get("share_pct"), "in_step_frac": dom.get("in_step_frac") if current else None,
                "how": "the kernel family with the largest share of GPU time in the committed single-stream rocprofv3 summary " + IN_STEP_FILE +
                       " (chosen by tools/roofline_model.py, not by hand); achieved / frac = algorithmic FLOPs per launch / average launch "
                       "duration of one iteration's launch mix of that family replayed live in isolation, HIP events on the launch stream; "
                       "in_step_frac = the family's algorithmic FLOPs per iteration / its kernel time per iteration inside the "
                       "single-stream step, from that summary" +
                       ("" if current else " (the committed summary is STALE — taken on other kernel sources: the family was chosen "
                                           "among the live replays by replayed time, in-step figures are not used)")}
            out["roofline_families"] = fams

            def rank_frac(d):
                return d["in_step_frac"] if (current and d.get("in_step_frac") is not None) else d["frac"]
            # the worst family PER ROOF: fractions of the MFMA peak and of the HBM peak are not comparable
            for bound, key in (("mfma", "roofline_worst"), ("hbm", "roofline_worst_hbm")):
                cands = [d for d in rankable if d["bound"] == bound]
                if not cands:
                    continue
                worst = min(cands, key=rank_frac)
                out[key] = dict(worst)
                out[key]["how"] = ("the lowest %s roofline fraction among the %s-bound kernel families with >= 5 %% of the step's kernel "
                                   "time (every such family is listed in roofline_families; the other roof's worst family is in %s)"
                                   % ("in-step" if current and worst.get("in_step_frac") is not None else "live", bound,
                                      "roofline_worst_hbm" if bound == "mfma" else "roofline_worst"))
            out["profile"] = {"file": IN_STEP_FILE, "matches_kernel_sources": current,
                              "kernel_time_ms_per_iteration_single_stream": instep["kernel_time_ms_per_iteration"] if instep else None,
                              "families_cover_pct": round(sum(d.get("share_pct", 0) for d in fams), 1) if instep else None}
            wtraffic = committed_traffic(S, B, TRAFFIC_FILE)
            for d in fams:
                if d["family"] == "wgrad":
                    d["traffic"] = wtraffic
                    d["algorithmic_bytes_per_launch"] = round(wgrad_algorithmic_bytes(S, B, cfgname))
        if world == 1 and not args.no_cpu_baseline:
            threads = host_threads()
            print("[bench] cpu_baseline on %d host threads (affinity / cgroup share)" % threads, file=sys.stderr, flush=True)
            cv, cdt, cutts = cpu_baseline(S, args.cpu_sample_batch, threads, config=cfgname)
            out["cpu_baseline"] = {"value": round(cv, 2), "unit": "utterances/s", "cores": threads, "kind": "port",
                                   "sample": "1 full %d-sub-step iteration, stock PyTorch CPU nn.TransformerEncoder "
                                             "stacks (oracle/stock_modules.py), train-mode dropout, %s%d dialogues "
                                             "padded to S=%d (%d real utterances), %.1f s" %
                                             (len(eng.schedule), "the identical batch: " if args.cpu_sample_batch == B else "",
                                              args.cpu_sample_batch, S, int(cutts), cdt)}
            out["config"]["gpu_over_cpu"] = round(value / cv, 1)
            cv0, cdt0, _ = cpu_baseline(S, args.cpu_sample_batch, threads, dropout=False, config=cfgname)
            out["cpu_baseline"]["dropout_free_value"] = round(cv0, 2)     # same sample, every dropout p = 0
            out["cpu_baseline"]["dropout_free_seconds"] = round(cdt0, 1)
            out["config"]["gpu_over_cpu_dropout_free"] = round(value / cv0, 1)
            if threads > 8:       # an 8-thread figure, comparable with the survey's 8-core measurement
                cv8, cdt8, _ = cpu_baseline(S, args.cpu_sample_batch, 8, config=cfgname)      # the same sample as `value`
                out["cpu_baseline"]["value_8_threads"] = round(cv8, 2)
                out["cpu_baseline"]["seconds_8_threads"] = round(cdt8, 1)
        print(json.dumps(out), flush=True)
    if pg is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
