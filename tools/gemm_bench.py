"""Micro-benchmark of the GEMM kernels over the shapes of the GAN-FFN step (tuning aid, GPU only)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_ffn_amd import _lib, ops  # noqa: E402

lib = _lib.load()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


def bench(kind, M, N, K, cfgs, tn_targets=(0,)):
    st = ops._stream()
    out = []
    # operand distribution matters (MFMA power / clocks): zero-mean data like the real activations and weights by
    # default, DIST=rand for uniform [0,1)
    rnd = torch.rand if os.environ.get("DIST", "randn") == "rand" else torch.randn
    if kind == "nt":
        a, w, b, c = rnd(M, K, device="cuda"), rnd(N, K, device="cuda"), torch.rand(N, device="cuda"), torch.empty(M, N, device="cuda")
        fn = lambda: _lib.call("ganffn_gemm_nt", ops._ptr(a), ops._ptr(w), ops._ptr(b), ops._ptr(c), M, N, K, st)
    elif kind == "nn":
        a, w, c = rnd(M, K, device="cuda"), rnd(K, N, device="cuda"), torch.empty(M, N, device="cuda")
        fn = lambda: _lib.call("ganffn_gemm_nn", ops._ptr(a), ops._ptr(w), ops._ptr(c), M, N, K, st)
    else:
        a, w, c, s = rnd(K, M, device="cuda"), rnd(K, N, device="cuda"), torch.zeros(M, N, device="cuda"), torch.zeros(M, device="cuda")
        fn = lambda: _lib.call("ganffn_gemm_tn_acc", ops._ptr(a), ops._ptr(w), ops._ptr(c), ops._ptr(s), M, N, K, st)
    us = timeit(fn)
    out.append("%6.1fus %5.1fTF" % (us, 2.0 * M * N * K / us / 1e6))
    print("%s M=%5d N=%5d K=%5d | %s" % (kind, M, N, K, " | ".join(out)), flush=True)


def bench_torch(M, N, K, iters=50):
    rnd = torch.rand if os.environ.get("DIST", "randn") == "rand" else torch.randn
    a = rnd(M, K, device="cuda")
    w = rnd(N, K, device="cuda")
    for _ in range(5):
        torch.mm(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.mm(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"torch.mm nt M={M:5d} N={N:5d} K={K:5d}: {us:7.1f}us {2.0 * M * N * K / us / 1e6:6.1f}TF", flush=True)


if __name__ == "__main__":
    cfgs = None    # tile-config sweeps used a debug hook that the release library no longer carries
    torch.backends.cuda.matmul.allow_tf32 = False
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        for (N, K) in ((2048, 100), (2048, 512), (512, 2048), (1536, 512)):
            bench("nt", 6016, N, K, cfgs)
        bench("nn", 6016, 2048, 100, cfgs)
        bench("nn", 6016, 100, 2048, cfgs)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        for M in (3008, 6016):
            for (N, K) in ((100, 100), (300, 100)):
                bench("nt", M, N, K, cfgs)
            for (N, K) in ((100, 100), (100, 300)):
                bench("nn", M, N, K, cfgs)
        sys.exit(0)
    for (N, K) in ((2048, 512), (1536, 512), (512, 512), (512, 2048), (2048, 100)):
        bench("nt", 3008, N, K, cfgs)
        bench_torch(3008, N, K)
    bench("nt", 6016, 2048, 100, cfgs)
    for (N, K) in ((2048, 512), (2048, 100), (512, 2048)):
        bench("nn", 3008, N, K, cfgs)
