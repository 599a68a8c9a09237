"""lab: the d_model-100 grouped weight-gradient launch with rows 96..99 on v_mfma_f32_4x4x1 (default) against the padded
seventh 16-row tile (bit 23); HIP-event timing of 20 back-to-back launches, alternating, T = 6016 and 3008"""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, ops
lib = _lib.load()
P, st = ops._ptr, ops._stream()
probs = [(100, 2048), (2048, 100), (100, 100), (300, 100)] * 8
n = len(probs)
nws = int(lib.ganffn_gemm_tn_grouped_workspace_floats())
ws = torch.empty(nws, device="cuda")
for T in (6016, 3008):
    bufs = {}
    for (m, nn) in set(probs):
        bufs[(m, nn)] = (torch.randn(T, m, device="cuda"), torch.randn(T, nn, device="cuda"))
    Cd = [torch.zeros(m, nn, device="cuda") for (m, nn) in probs]
    Sd = [torch.zeros(m, device="cuda") for (m, nn) in probs]
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    A_, B_, C_, S_ = arr([bufs[p][0] for p in probs]), arr([bufs[p][1] for p in probs]), arr(Cd), arr(Sd)
    Ms, Ns, Ks = (C.c_int * n)(*[p[0] for p in probs]), (C.c_int * n)(*[p[1] for p in probs]), (C.c_int * n)(*[T] * n)
    flop = sum(2.0 * m * nn * T for (m, nn) in probs)
    for rep in range(2):
        for bits, name in ((0, "4x4x1 tail"), (1 << 23, "padded tile")):
            lib.ganffn_debug_set_ffn_mode(bits)
            call = lambda: _lib.call("ganffn_gemm_tn_grouped", n, A_, B_, C_, S_, Ms, Ns, Ks, P(ws), nws, st)
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20):
                call()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print("T=%d %-12s %7.1f us  %.1f TFLOP/s useful (%.0f %% of 157.3)" % (T, name, us, flop / us / 1e6, flop / us / 1e6 / 1.573), flush=True)
lib.ganffn_debug_set_ffn_mode(0)
