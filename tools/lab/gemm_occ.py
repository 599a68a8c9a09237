import os, sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from gan_ffn_amd import _lib, ops
lib = _lib.load()
M, N, K = 3008, 2048, 512
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.rand(N, device="cuda"); c = torch.empty(M, N, device="cuda")
st = ops._stream()
for cfg in [int(c) for c in os.environ.get("CFGS", "1,7").split(",")]:
    lib.ganffn_debug_set_gemm_cfg(cfg, 0)
    for _ in range(5):
        _lib.call("ganffn_gemm_nt", ops._ptr(a), ops._ptr(w), ops._ptr(b), ops._ptr(c), M, N, K, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        _lib.call("ganffn_gemm_nt", ops._ptr(a), ops._ptr(w), ops._ptr(b), ops._ptr(c), M, N, K, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    print("pad", os.environ.get("GANFFN_LAB_LDSPAD"), "cfg", cfg, "%.1f us %.1f TF" % (us, 2.0 * M * N * K / us / 1e6), flush=True)
