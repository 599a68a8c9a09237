"""Run one GEMM shape a few times under a forced config (for rocprofv3 --pmc collection)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
cfg = int(os.environ.get("CFG", "1"))
lib.ganffn_debug_set_gemm_cfg(cfg, 0)
st = ops._stream()
M, N, K = 3008, 2048, 512
a, w, b, c = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.rand(N, device="cuda"), torch.empty(M, N, device="cuda")
for _ in range(5):
    _lib.call("ganffn_gemm_nt", ops._ptr(a), ops._ptr(w), ops._ptr(b), ops._ptr(c), M, N, K, st)
torch.cuda.synchronize()
