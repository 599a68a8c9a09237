"""The two K = 100 -> N = 2048 products of the d_model-100 feed-forward block in isolation: linear1 forward with its fused
bias + ReLU + dropout epilogue (ganffn_ffn_linear1_fwd) and the plain NN product of linear2's dgrad, T = 3008 / 6016."""
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
for _once in (0,):
  for T in (3008, 6016):
      x, w1, b1 = torch.randn(T, 100, device="cuda"), torch.randn(2048, 100, device="cuda") * 0.1, torch.randn(2048, device="cuda") * 0.1
      h = torch.empty(T, 2048, device="cuda")
      for train in (0, 1):
          us = timeit(lambda: _lib.call("ganffn_ffn_linear1_fwd", ops._ptr(x), ops._ptr(w1), ops._ptr(b1), ops._ptr(h), T, 100, 2048,
                                        C.c_float(0.1), 18, ops._ptr(rng), C.c_uint64(0), train, st))
          print("linear1 fwd + bias/ReLU%s  T=%d: %6.1f us  %5.1f TF" % ("/dropout" if train else "", T, us, 2.0 * T * 2048 * 100 / us / 1e6))
      w2 = torch.randn(100, 2048, device="cuda") * 0.1
      us = timeit(lambda: _lib.call("ganffn_gemm_nn", ops._ptr(x), ops._ptr(w2), ops._ptr(h), T, 2048, 100, st))
      print("NN [T x 100] x [100 x 2048]      T=%d: %6.1f us  %5.1f TF" % (T, us, 2.0 * T * 2048 * 100 / us / 1e6))
      us = timeit(lambda: _lib.call("ganffn_gemm_nt", ops._ptr(x), ops._ptr(w1), ops._ptr(b1), ops._ptr(h), T, 2048, 100, st))
      print("NT plain + bias                  T=%d: %6.1f us  %5.1f TF" % (T, us, 2.0 * T * 2048 * 100 / us / 1e6))
