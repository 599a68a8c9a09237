#!/usr/bin/env python3
"""bench.py — utterances/sec per GAN train step (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
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


TRAFFIC_FILE = "profiles/r03_wgrad_traffic.json"


def wgrad_algorithmic_bytes(S, B, config="iemocap"):
    """average compulsory bytes of one launch: every dY and X operand read once, every dW (and db) written once"""
    tot, n = 0.0, 0
    for cnt, probs in wgrad_groups(S, B, config):
        b = sum(4.0 * (k * m + k * n_ + m * n_ + m) for (m, n_, k) in probs)
        tot += cnt * b
        n += cnt
    return tot / n


def committed_traffic(S, B):
    """HBM-side bytes per launch of the roofline kernel from the committed PMC passes (tools/traffic_pmc.sh), or None
    when the file is absent or was taken at another problem size"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), TRAFFIC_FILE)
    try:
        d = json.load(open(path))
        if d.get("seq_len") == S and d.get("dialogues_per_gpu") == B:
            return d["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


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
    one_iteration()
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


def time_linear1_kernel(S, B, reps=3):
    """Live HIP-event timing (launch stream = torch's current stream) of the heavy kernel that sits lowest on its roofline:
    the d_model-100 feed-forward linear1 GEMM with its fused bias + ReLU + dropout epilogue, `gemm_wres_kernel<0,1,100>`
    ([T x 100] x [2048 x 100]^T, K = 100).  One iteration launches it 112 times at
    T = S*B (4 generator + 6 frozen-discriminator + 4 no-save generator passes x 8 layers) and 48 times at T = 2*S*B (the
    six batched [real | fake] discriminator passes).  Returns (avg seconds per launch, avg algorithmic flops per launch, launches)."""
    from gan_ffn_amd import _lib, ops
    st = ops._stream()
    E, F = 100, 2048
    rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
    calls = []
    for T, cnt in ((S * B, 112), (2 * S * B, 48)):
        x = torch.rand(T, E, device="cuda") - 0.5
        w1, b1 = (torch.rand(F, E, device="cuda") - 0.5) * 0.2, torch.zeros(F, device="cuda")
        h = torch.empty(T, F, device="cuda")
        calls.append((cnt, T, (x, w1, b1, h)))

    def one_iteration():
        for cnt, T, (x, w1, b1, h) in calls:
            for _ in range(cnt):
                _lib.call("ganffn_ffn_linear1_fwd", ops._ptr(x), ops._ptr(w1), ops._ptr(b1), ops._ptr(h), T, E, F, C.c_float(0.1),
                          C.c_uint32(18), ops._ptr(rng), C.c_uint64(0), 1, st)
    one_iteration()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        one_iteration()
    e1.record()
    torch.cuda.synchronize()
    n = sum(c[0] for c in calls)
    flops = sum(c[0] * 2.0 * c[1] * E * F for c in calls) / n
    return e0.elapsed_time(e1) * 1e-3 / (reps * n), flops, n


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
    one_iteration()
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


IN_STEP_FILE = "profiles/r03_bench_streams1_by_launch_shape.txt"


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


DRNN_IN_STEP_FILE = "profiles/r03_drnn_by_launch_shape.txt"


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
    """average duration of a kernel inside the configuration-5 step, from the committed rocprofv3 summary (or None)"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), DRNN_IN_STEP_FILE)
    try:
        for line in open(path):
            f = line.split()
            if len(f) >= 5 and f[0].endswith("%") and " ".join(f[4:]).startswith(symbol_prefix):
                return float(f[2])
    except Exception:
        pass
    return None


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
                         "how": "HIP events around 20 back-to-back launches on the launch stream (in_step_avg_us: the same "
                                "kernel inside the step, from the committed rocprofv3 summary); the 24 MB of weights stay "
                                "L2 / MALL-resident between launches, so the bound that applies is the L2 -> CU path, priced "
                                "here against the HBM peak as the contract asks"}}), flush=True)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
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
    ap.add_argument("--replay-dominant-only", action="store_true",
                    help="only replay the roofline kernel's launch mix once (the command tools/traffic_pmc.sh profiles "
                         "with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass)")
    args = ap.parse_args()
    if args.seq is None:
        args.seq = 33 if args.config == "meld" else 94
    if args.config == "drnn" and args.batch == 32:
        args.batch = 30                                              # train_IEMOCAP_DialogueRNN.py:580
    if args.cpu_sample_batch is None:
        args.cpu_sample_batch = args.batch
    cfgname = args.config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the hot path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    force_dist = os.environ.get("GANFFN_FORCE_DIST", "0") == "1"   # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        pg = dist.group.WORLD

    from gan_ffn_amd import _lib, engine, ops
    if args.config == "drnn":
        _lib.load()
        run_drnn(args, dev, pg, rank, world)
        if pg is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.replay_dominant_only:
        kt, kflop, klaunch = time_dominant_kernel(args.seq, args.batch, reps=1, config=cfgname)
        print(json.dumps({"avg_kernel_us": kt * 1e6, "launches": klaunch}), flush=True)
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

    for _ in range(args.warmup):
        eng.iteration(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.iteration(batch)
    sync()
    dt = time.perf_counter() - t0
    losses = eng.loss_dict()

    utts = torch.tensor([float(batch["umask"].sum()), dt], device=dev, dtype=torch.float64)
    if pg is not None:
        import torch.distributed as dist
        tmax = utts[1:2].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        usum = utts[0:1].clone()
        dist.all_reduce(usum, op=dist.ReduceOp.SUM)
        dt, total_utts = float(tmax), float(usum)
    else:
        total_utts = float(utts[0])
    ms_per_step = dt / args.steps * 1e3
    value = total_utts * args.steps / dt
    if rank == 0:
        print("[bench] gpu: %.3f ms/step, %.1f utterances/s (%s)" % (ms_per_step, value, "hipGraph" if use_graph else "eager"),
              file=sys.stderr, flush=True)

    if rank == 0 and args.step_only:
        print(json.dumps({"metric": "utterances/sec per GAN train step (step only)", "value": round(value, 2), "unit": "utterances/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
                          "config": {"workload": cfgname, "streams": eng.n_streams}}), flush=True)
    elif rank == 0:
        kt, kflop, klaunch = time_dominant_kernel(S, B, config=cfgname)
        traffic = committed_traffic(S, B) if cfgname == "iemocap" else None
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
                       "last_losses": {k: round(v, 4) for k, v in losses.items()}},
            "roofline": {"bound": "mfma", "kernel": "the grouped weight-gradient launch = all 32 weight-gradient GEMMs (+ bias gradients) of one "
                                                     "encoder backward pass: tn100_kernel + its ordered slab reduce for the d_model-100 "
                                                     "networks (112-wide 16x16x4 tiles, csrc/gemm_tn100.hip), gemm_tn_grouped_kernel<false,128> "
                                                     "for the d_model-512 generator; no atomics; %d launches per iteration; the kernel "
                                                     "family with the largest share of GPU time in the single-stream profile" % klaunch,
                         "achieved": round(kflop / kt / 1e12, 2), "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": round(kflop / kt / FP32_MFMA_PEAK, 4), "traffic": traffic,
                         "traffic_unit": "bytes per launch (FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE, separate "
                                         "rocprofv3 --pmc passes: " + TRAFFIC_FILE + ")" if traffic is not None else None,
                         "algorithmic_bytes_per_launch": round(wgrad_algorithmic_bytes(S, B, cfgname)),
                         "avg_kernel_us": round(kt * 1e6, 2), "avg_gflop_per_launch": round(kflop / 1e9, 4),
                         "how": "HIP events around one iteration's launch mix of this launch replayed in isolation on the launch "
                                "stream; in_step_avg_us = the same launches inside the single-stream step, from the committed "
                                "rocprofv3 summary " + IN_STEP_FILE,
                         "in_step_avg_us": in_step_wgrad_us() if (cfgname == "iemocap" and (S, B) == (94, 32)) else None},
        }
        if cfgname == "iemocap":
            # heavy kernel FAMILIES (>= 5 % of GPU time in profiles/r03_bench_streams1_*: the K = 100 -> 2048 products of the
            # d_model-100 feed-forward block, 16.6 %, and the 2048 -> 100 ones, 13.4 %), each timed live; the one that sits
            # lowest on its roofline is reported as roofline_worst, the other beside it
            lt, lflop, ln = time_linear1_kernel(S, B)
            nt_, nflop, nn = time_n100_kernel(S, B)
            full = (S, B) == (94, 32)
            cands = [
                {"kernel": "gemm_wres_kernel<0,1,100> = linear1 of the d_model-100 feed-forward block with fused bias + ReLU + "
                           "dropout ([T x 100] x [2048 x 100]^T, K = 100; persistent, weight fragments register-resident); "
                           "%d launches per iteration (T = S*B and 2*S*B)" % ln,
                 "achieved": round(lflop / lt / 1e12, 2), "frac": round(lflop / lt / FP32_MFMA_PEAK, 4), "avg_kernel_us": round(lt * 1e6, 2),
                 "avg_gflop_per_launch": round(lflop / 1e9, 4), "in_step_avg_us": in_step_kernel_us("gemm_wres_kernel<0, 1, 100>") if full else None},
                {"kernel": "gemm_n100_kernel = [T x 2048] x [2048 x 100] on 16x16x4 MFMAs, 112-wide feature tile, K-chunk slabs "
                           "(linear2 forward and the linear1 dgrad; csrc/gemm_n100.hip); %d launches per iteration" % nn,
                 "achieved": round(nflop / nt_ / 1e12, 2), "frac": round(nflop / nt_ / FP32_MFMA_PEAK, 4), "avg_kernel_us": round(nt_ * 1e6, 2),
                 "avg_gflop_per_launch": round(nflop / 1e9, 4), "in_step_avg_us": in_step_kernel_us("gemm_n100_kernel") if full else None},
            ]
            for c_ in cands:
                c_.update({"bound": "mfma", "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "traffic": None})
                if c_["in_step_avg_us"]:
                    c_["in_step_frac"] = round(c_["avg_gflop_per_launch"] * 1e9 / (c_["in_step_avg_us"] * 1e-6) / FP32_MFMA_PEAK, 4)
            cands.sort(key=lambda c_: c_["frac"])
            out["roofline_worst"] = dict(cands[0])
            out["roofline_worst"]["how"] = ("HIP events around one iteration's launch mix of this kernel replayed in isolation on the "
                                            "launch stream; chosen as the lowest roofline fraction among the kernel families with >= 5 % "
                                            "of GPU time; in_step_* = the same symbol's launch-weighted average inside the single-stream "
                                            "step, from the committed rocprofv3 summary " + IN_STEP_FILE)
            out["roofline_worst"]["other_heavy_families"] = cands[1:]
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
