"""Launch the kernels under study a few times each (for rocprofv3 --pmc / --kernel-trace collection; GPU only):
linear1 + ReLU + dropout GEMM (T = 3008 / 6016), linear2, the small-head attention forward / backward."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops  # noqa: E402

lib = _lib.load()
P = ops._ptr
st = ops._stream()
E, F = 100, 2048
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
for T in (3008, 6016):
    x = torch.randn(T, E, device="cuda")
    w1, b1 = torch.randn(F, E, device="cuda") / 10, torch.randn(F, device="cuda") / 10
    w2, b2 = torch.randn(E, F, device="cuda") / 45, torch.randn(E, device="cuda") / 10
    h = torch.empty(T, F, device="cuda")
    y = torch.empty(T, E, device="cuda")
    for _ in range(10):
        _lib.call("ganffn_ffn_linear1_fwd", P(x), P(w1), P(b1), P(h), T, E, F, C.c_float(0.1), C.c_uint32(18), P(rng), C.c_uint64(0), 1, st)
    for _ in range(10):
        _lib.call("ganffn_gemm_nt", P(h), P(w2), P(b2), P(y), T, E, F, st)
S, H = 94, 10
for B in (32, 64):
    qkv = torch.randn(S, B, 3 * E, device="cuda")
    o = torch.empty(S, B, E, device="cuda")
    lse = torch.empty(B * H, S, device="cuda")
    do = torch.randn(S, B, E, device="cuda")
    dq = torch.empty(S, B, 3 * E, device="cuda")
    for p in (0.0, 0.1):
        for _ in range(10):
            _lib.call("ganffn_attention_fwd", P(qkv), P(o), P(lse), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
        for _ in range(10):
            _lib.call("ganffn_attention_bwd", P(qkv), P(o), P(lse), P(do), P(dq), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
torch.cuda.synchronize()
