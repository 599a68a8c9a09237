"""in-kernel time stamps of gemm_wres_kernel (linear1 forward, K = 100 -> 2048): prologue, tile loop, tiles per workgroup"""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import os
os.environ.setdefault("GANFFN_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gan_ffn_amd", "lib", "libganffn_lab.so"))  # make -C gan_ffn_amd/csrc LAB=1
from gan_ffn_amd import _lib, ops
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
P, st = ops._ptr, ops._stream()
rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
for T in (3008, 6016):
    for train in (0, 1):
        x, w1, b1 = torch.randn(T, 100, device="cuda"), torch.randn(2048, 100, device="cuda") * 0.1, torch.randn(2048, device="cuda") * 0.1
        h = torch.empty(T, 2048, device="cuda")
        call = lambda: _lib.call("ganffn_ffn_linear1_fwd", P(x), P(w1), P(b1), P(h), T, 100, 2048, C.c_float(0.1), 18, P(rng), C.c_uint64(0), train, st)
        for _ in range(5):
            call()
        stamps = torch.zeros(2048 * 4, dtype=torch.int64, device="cuda")
        raw.ganffn_lab_set_wres_stamps(C.c_void_p(stamps.data_ptr()))
        call()
        torch.cuda.synchronize()
        raw.ganffn_lab_set_wres_stamps(None)
        v = stamps.view(-1, 4).cpu().double()
        v = v[v[:, 3] > 0]
        pro, loop, nt = v[:, 1] - v[:, 0], v[:, 2] - v[:, 1], v[:, 3]
        print("T=%d train=%d: %d workgroups, tiles per workgroup %.1f | prologue %.0f cyc | loop %.0f cyc = %.0f per tile (MFMA 52 x 64 = 3328 per wave, 2 workgroups per CU)"
              % (T, train, len(v), nt.mean(), pro.median(), loop.median(), (loop / nt).median()), flush=True)
