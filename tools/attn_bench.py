"""Micro-benchmark of the attention forward / backward kernels (tuning aid, GPU only)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_ffn_amd import _lib, ops  # noqa: E402

lib = _lib.load()


def timeit(fn, reps=50):
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


for (S, B, E, H) in ((94, 32, 100, 10), (94, 64, 100, 10), (94, 32, 512, 8), (94, 32, 600, 10), (94, 32, 300, 10)):
    qkv = torch.randn(S, B, 3 * E, device="cuda")
    o = torch.empty(S, B, E, device="cuda")
    lse = torch.empty(B * H, S, device="cuda")
    do = torch.randn(S, B, E, device="cuda")
    dq = torch.empty(S, B, 3 * E, device="cuda")
    rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
    st = ops._stream()
    for p in (0.0, 0.1):
        us = timeit(lambda: _lib.call("ganffn_attention_fwd", ops._ptr(qkv), ops._ptr(o), ops._ptr(lse), S, B, E, H, C.c_float(p),
                                      C.c_uint32(16), ops._ptr(rng), C.c_uint64(0), st))
        usb = timeit(lambda: _lib.call("ganffn_attention_bwd", ops._ptr(qkv), ops._ptr(o), ops._ptr(lse), ops._ptr(do), ops._ptr(dq),
                                       S, B, E, H, C.c_float(p), C.c_uint32(16), ops._ptr(rng), C.c_uint64(0), st))
        print("S=%d B=%d E=%d H=%d p=%.1f | fwd %6.1f us | bwd %6.1f us" % (S, B, E, H, p, us, usb), flush=True)
