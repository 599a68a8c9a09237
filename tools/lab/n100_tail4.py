"""lab: gemm_n100 with features 96..99 on v_mfma_f32_4x4x1 (default) against the padded seventh 16-wide tile (bit 23).
HIP-event timing of 100 back-to-back launches, both weight layouts, the chunk counts the rule picks."""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, ops
lib = _lib.load()
P, st = ops._ptr, ops._stream()
K = 2048


def timeit(fn, reps=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for T in (3008, 6016):
    A = torch.randn(T, K, device="cuda")
    Wn = torch.randn(100, K, device="cuda") / 45
    Wk = torch.randn(K, 100, device="cuda") / 45
    b = torch.randn(100, device="cuda")
    slabs = torch.empty(16, T, 100, device="cuda")
    n = C.c_int(0)
    for rep in range(2):
        for bits, name in ((0, "4x4x1 tail"), (1 << 23, "padded tile")):
            lib.ganffn_debug_set_ffn_mode(bits)
            r = [timeit(lambda: _lib.call("ganffn_gemm_n100", P(A), P(W), km, P(b), P(slabs), C.c_int64(T * 100), T, K, 16, C.byref(n), st))
                 for km, W in ((0, Wn), (1, Wk))]
            print("T=%d %-12s %d slabs: rows-of-K %.1f us, K-major %.1f us" % (T, name, n.value, r[0], r[1]), flush=True)
lib.ganffn_debug_set_ffn_mode(0)
