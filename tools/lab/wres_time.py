"""gemm_wres_kernel (linear1 forward with its epilogue) in isolation: sustained back-to-back launches, T = 3008 / 6016.
GANFFN_LIB selects a lab build (e.g. non-temporal output stores)."""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, ops
lib = _lib.load()
P, st = ops._ptr, ops._stream()
rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
for T in (3008, 6016):
    x, w1, b1 = torch.randn(T, 100, device="cuda"), torch.randn(2048, 100, device="cuda") * 0.1, torch.randn(2048, device="cuda") * 0.1
    h = torch.empty(T, 2048, device="cuda")
    call = lambda: _lib.call("ganffn_ffn_linear1_fwd", P(x), P(w1), P(b1), P(h), T, 100, 2048, C.c_float(0.1), 18, P(rng), C.c_uint64(0), 1, st)
    for _ in range(300):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(500):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 500
    print("%s T=%d: %.2f us per launch = %.1f TFLOP/s" % (os.environ.get("GANFFN_LIB", "default lib"), T, us, 2.0 * T * 100 * 2048 / us / 1e6), flush=True)
