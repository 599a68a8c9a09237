"""Run a few GEMM shapes a few times each (for rocprofv3 --pmc collection)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
st = ops._stream()
for (M, N, K) in ((3008, 2048, 100), (3008, 2048, 512), (3008, 512, 2048), (3008, 300, 100)):
    a, w, b, c = torch.rand(M, K, device="cuda"), torch.rand(N, K, device="cuda"), torch.rand(N, device="cuda"), torch.empty(M, N, device="cuda")
    for _ in range(5):
        _lib.call("ganffn_gemm_nt", ops._ptr(a), ops._ptr(w), ops._ptr(b), ops._ptr(c), M, N, K, st)
    torch.cuda.synchronize()
