"""what does reading the saved activation cost the linear2 dgrad of the d_model-100 feed-forward block?  [T x 100] x [100 x 2048]
through ganffn_gemm_hook with the mask epilogue (reads h [T x 2048]) against the plain epilogue (same product, no aux read):
the upper bound of what a 1-bit mask instead of the fp32 activation could save."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
_lib.load()
P, st = ops._ptr, ops._stream()
rng = torch.tensor([1, 0], dtype=torch.int64, device="cuda")
def timeit(fn, reps=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
n = C.c_int(0)
for T in (3008, 6016):
    x, w2 = torch.randn(T, 100, device="cuda"), torch.randn(100, 2048, device="cuda") * 0.1
    h, out = torch.relu(torch.randn(T, 2048, device="cuda")), torch.empty(T, 2048, device="cuda")
    f = lambda epi, aux: (lambda: _lib.call("ganffn_gemm_hook", 1, epi, P(x), P(w2), None, P(aux) if aux is not None else None, P(out), C.c_int64(T * 2048),
                                            T, 2048, 100, C.c_float(0.1), C.c_uint32(18), P(rng), C.c_uint64(0), 1, 1, C.byref(n), st))
    r = [(timeit(f(3, h)), timeit(f(0, None))) for _ in range(3)]
    print("T=%d: mask epilogue %s us | plain %s us" % (T, " ".join("%.2f" % a for a, _ in r), " ".join("%.2f" % b for _, b in r)), flush=True)
