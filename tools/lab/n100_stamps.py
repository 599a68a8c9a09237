"""in-kernel time stamps of gemm_n100 (lab hook ganffn_lab_set_n100_stamps): per workgroup start skew, prologue, K loop, epilogue"""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import os
os.environ.setdefault("GANFFN_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gan_ffn_amd", "lib", "libganffn_lab.so"))  # make -C gan_ffn_amd/csrc LAB=1
from gan_ffn_amd import _lib, ops
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
P, st = ops._ptr, ops._stream()
K = 2048
for T, s in ((3008, 5), (3008, 10), (6016, 8)):
    A = torch.randn(T, K, device="cuda"); W = torch.randn(100, K, device="cuda") / 45; b = torch.randn(100, device="cuda")
    slabs = torch.empty(16, T, 100, device="cuda"); n = C.c_int(0)
    lib.ganffn_debug_set_ffn_mode(s << 8)
    nwg = ((T + 63) // 64) * s
    stamps = torch.zeros(nwg * 5, dtype=torch.int64, device="cuda")
    call = lambda: _lib.call("ganffn_gemm_n100", P(A), P(W), 0, P(b), P(slabs), C.c_int64(T * 100), T, K, 16, C.byref(n), st)
    for _ in range(5):
        call()
    raw.ganffn_lab_set_n100_stamps(C.c_void_p(stamps.data_ptr()))
    call()
    torch.cuda.synchronize()
    raw.ganffn_lab_set_n100_stamps(None)
    v = stamps.view(nwg, 5).cpu().double()
    rt = (v[:, 0] - v[:, 0].min()) / 100.0          # us (100 MHz)
    pro, loop, epi = v[:, 2] - v[:, 1], v[:, 3] - v[:, 2], v[:, 4] - v[:, 3]
    steps = (K // 32 + s - 1) // s
    print("T=%d chunks=%d (%d workgroups, %d steps): start skew median %.2f us max %.2f us | prologue %.0f cyc | K loop %.0f cyc (%.0f per step; min %.0f max %.0f) | epilogue %.0f cyc"
          % (T, s, nwg, steps, rt.median(), rt.max(), pro.median(), loop.median(), loop.median() / steps, loop.min(), loop.max(), epi.median()), flush=True)
lib.ganffn_debug_set_ffn_mode(0)
