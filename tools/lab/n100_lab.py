"""[T x 2048] x [2048 x 100] on gemm_n100.hip against the generic 64 x 64 kernel, both weight layouts, forced K-chunk counts
(ganffn_debug_set_ffn_mode bits 8..15).  HIP-event timing of 50 back-to-back launches."""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, ops
lib = _lib.load()
P, st = ops._ptr, ops._stream()
K = 2048


def timeit(fn, reps=50):
    for _ in range(5):
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
    y = torch.empty(T, 100, device="cuda")
    n = C.c_int(0)
    t_nt = timeit(lambda: _lib.call("ganffn_gemm_nt", P(A), P(Wn), P(b), P(y), T, 100, K, st))
    t_nn = timeit(lambda: _lib.call("ganffn_gemm_nn", P(A), P(Wk), P(y), T, 100, K, st))
    print("T=%d generic (unsplit) NT %.1f us, NN %.1f us" % (T, t_nt, t_nn), flush=True)
    for s in (0, 3, 4, 5, 6, 8, 10, 16):
        for kw in (1, 2):                     # waves per 16-token group along K (bits 20..21): 4- or 8-wave workgroups
            lib.ganffn_debug_set_ffn_mode((s << 8) | (kw << 20))
            r = []
            for km, W in ((0, Wn), (1, Wk)):
                r.append(timeit(lambda: _lib.call("ganffn_gemm_n100", P(A), P(W), km, P(b), P(slabs), C.c_int64(T * 100), T, K, 16, C.byref(n), st)))
            print("T=%d forced chunks %2d -> %2d slabs, %d waves: rows-of-K %.1f us, K-major %.1f us  (%.1f / %.1f TFLOP/s useful)" % (
                T, s, n.value, 4 * kw, r[0], r[1], 2e-6 * T * 100 * K / r[0], 2e-6 * T * 100 * K / r[1]), flush=True)
lib.ganffn_debug_set_ffn_mode(0)
