"""a few launches of gemm_n100 (both layouts, T = 3008 / 6016) for rocprofv3 --pmc passes"""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, ops
lib = _lib.load()
P, st = ops._ptr, ops._stream()
K = 2048
for T in (3008, 6016):
    A = torch.randn(T, K, device="cuda")
    Wn = torch.randn(100, K, device="cuda") / 45
    Wk = torch.randn(K, 100, device="cuda") / 45
    b = torch.randn(100, device="cuda")
    slabs = torch.empty(16, T, 100, device="cuda")
    n = C.c_int(0)
    for km, W in ((0, Wn), (1, Wk)):
        for _ in range(10):
            _lib.call("ganffn_gemm_n100", P(A), P(W), km, P(b), P(slabs), C.c_int64(T * 100), T, K, 16, C.byref(n), st)
torch.cuda.synchronize()
