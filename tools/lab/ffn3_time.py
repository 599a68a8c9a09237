"""lab: ffn3.hip's single forward kernel against the two kernels it replaces (gemm_wres linear1 + gemm_n100 linear2), in
isolation: T = 3008 / 6016, train-mode dropout and eval, with and without the hidden-tensor store.
    python tools/lab/ffn3_time.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib as lib  # noqa: E402

ptr = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    E, F = 100, 2048
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    for T in (3008, 6016):
        x = torch.randn(T, E, device="cuda")
        w1, b1 = torch.randn(F, E, device="cuda") / 10, torch.randn(F, device="cuda") / 10
        w2, b2 = torch.randn(E, F, device="cuda") / 45, torch.randn(E, device="cuda") / 10
        h = torch.empty(T, F, device="cuda")
        slabs = torch.empty(16, T, E, device="cuda")
        n = C.c_int(0)
        for p, train in ((0.1, 1), (0.1, 0)):
            def two():
                lib.call("ganffn_ffn_linear1_fwd", ptr(x), ptr(w1), ptr(b1), ptr(h), T, E, F, C.c_float(p), C.c_uint32(7), ptr(rng),
                         C.c_uint64(0), train, st)
                lib.call("ganffn_gemm_n100", ptr(h), ptr(w2), 0, ptr(b2), ptr(slabs), C.c_int64(T * E), T, F, 16, C.byref(n), st)

            def one(save):
                lib.call("ganffn_ffn3_fwd", ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(h) if save else None, ptr(slabs),
                         C.c_int64(T * E), T, C.c_float(p), C.c_uint32(7), ptr(rng), C.c_uint64(0), train, 16, C.byref(n), st)
            t2 = timeit(two)
            n2 = n.value
            t1s = timeit(lambda: one(True))
            t1n = timeit(lambda: one(False))
            print("T=%d train=%d: linear1 + n100 (%d slabs) %.1f us | ffn3 save %.1f us  nosave %.1f us (%d slabs)" %
                  (T, train, n2, t2, t1s, t1n, n.value), flush=True)


if __name__ == "__main__":
    main()
