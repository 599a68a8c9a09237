"""Ablation timings of the fused feed-forward kernel (d_model 100): with / without the hidden-activation store, with /
without dropout, against the two-GEMM path.  GPU only; GANFFN_LIB may point at a lab build."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops  # noqa: E402

lib = _lib.load()
E, F = 100, 2048


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for T in (3008, 6016):
    x = torch.randn(T, E, device="cuda")
    w1, b1 = torch.randn(F, E, device="cuda") / 10, torch.randn(F, device="cuda") / 10
    w2, b2 = torch.randn(E, F, device="cuda") / 45, torch.randn(E, device="cuda") / 10
    h = torch.empty(T, F, device="cuda")
    dh = torch.empty(T, F, device="cuda")
    slabs = torch.empty(16, T, E, device="cuda")
    pack = torch.empty(int(lib.ganffn_ffn_pack_floats(F)), device="cuda")
    y = torch.empty(T, E, device="cuda")
    rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
    st = ops._stream()
    P = ops._ptr
    gf = 4.0 * T * E * F / 1e3     # MFLOP -> us * TF
    for p in (0.1, 0.0):
        for keep_h in (True, False):
            us = timeit(lambda: lib.ganffn_ffn_fused_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(h) if keep_h else None, P(slabs), P(pack),
                                                         T, E, F, C.c_float(p), C.c_uint32(18), P(rng), C.c_uint64(0), 1, st))
            print("T=%d fused fwd  p=%.1f h=%d : %6.1f us  %5.1f TF (incl. pack)" % (T, p, keep_h, us, gf / us / 1e3), flush=True)
    us = timeit(lambda: lib.ganffn_ffn_fused_bwd(P(x), P(w1), P(w2), P(h), P(dh), P(slabs), P(pack), T, E, F, C.c_float(1.1), st))
    print("T=%d fused bwd             : %6.1f us  %5.1f TF (incl. pack)" % (T, us, gf / us / 1e3), flush=True)
    us1 = timeit(lambda: _lib.call("ganffn_ffn_linear1_fwd", P(x), P(w1), P(b1), P(h), T, E, F, C.c_float(0.1), C.c_uint32(18), P(rng),
                                   C.c_uint64(0), 1, st))
    us2 = timeit(lambda: _lib.call("ganffn_gemm_nt", P(h), P(w2), P(b2), P(y), T, E, F, st))
    print("T=%d two GEMMs fwd         : %6.1f + %6.1f us (linear2 unsplit here)" % (T, us1, us2), flush=True)
