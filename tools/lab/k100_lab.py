"""linear1 forward (K = 100 -> 2048, bias + ReLU + dropout) on gemm_k100.hip (mode 0) against gemm_wres_kernel (mode 16),
T = 3008 / 6016, eval and train; and the whole step / passes (tools/lab/mode_ab.py does the latter)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
import ctypes as C
lib = _lib.load()
st = ops._stream()


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
for T in (3008, 6016):
    x, w1, b1 = torch.randn(T, 100, device="cuda"), torch.randn(2048, 100, device="cuda") * 0.1, torch.randn(2048, device="cuda") * 0.1
    h = torch.empty(T, 2048, device="cuda")
    for mode in (16, 0):
        lib.ganffn_debug_set_ffn_mode(mode)
        for train in (0, 1):
            us = timeit(lambda: _lib.call("ganffn_ffn_linear1_fwd", ops._ptr(x), ops._ptr(w1), ops._ptr(b1), ops._ptr(h), T, 100, 2048,
                                          C.c_float(0.1), 18, ops._ptr(rng), C.c_uint64(0), train, st))
            print("mode %2d linear1 fwd + bias/ReLU%-8s T=%d: %6.1f us  %5.1f TFLOP/s" % (mode, "/dropout" if train else "", T, us, 2.0 * T * 2048 * 100 / us / 1e6), flush=True)
lib.ganffn_debug_set_ffn_mode(0)
