"""Grouped weight-gradient launch of the d_model-512 generator (32 problems, T = 3008): 64 x 64 against 64 x 128 tiles.
Needs a lab build exporting `ganffn_lab_tn_wide(int)` (0: 64 x 64, 1: 64 x 128 where N % 128 == 0, 2: also N <= 128);
the release library applies the rule recorded in launch_gemm_tn_grouped."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
T, E, F = 3008, 512, 2048
probs = [(3 * E, E, T), (E, E, T), (F, E, T), (E, F, T)] * 8
n = len(probs)
g = torch.Generator().manual_seed(1)
bufs = {}
for (M, N, K) in set(probs):
    bufs[(M, N, K)] = ((torch.rand(K, M, generator=g) - 0.5).cuda(), (torch.rand(K, N, generator=g) - 0.5).cuda())
def run(wide):
    lib.ganffn_lab_tn_wide(wide)
    outs = [(torch.zeros(M, N, device="cuda"), torch.zeros(M, device="cuda")) for (M, N, K) in probs]
    PA = (C.c_void_p * n)(*[bufs[p][0].data_ptr() for p in probs]); PB = (C.c_void_p * n)(*[bufs[p][1].data_ptr() for p in probs])
    PC = (C.c_void_p * n)(*[o[0].data_ptr() for o in outs]); PS = (C.c_void_p * n)(*[o[1].data_ptr() for o in outs])
    Ms = (C.c_int * n)(*[p[0] for p in probs]); Ns = (C.c_int * n)(*[p[1] for p in probs]); Ks = (C.c_int * n)(*[p[2] for p in probs])
    call = lambda: _lib.call("ganffn_gemm_tn_grouped", n, PA, PB, PC, PS, Ms, Ns, Ks, None, 0, ops._stream())
    call()
    first = [o[0].clone() for o in outs[:4]] + [outs[0][1].clone()]
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = sum(2.0 * M * N * K for M, N, K in probs)
    print("wide=%d: %.1f us  %.1f TFLOP/s" % (wide, us, fl / us / 1e6))
    return first
a = run(0); b = run(1); a2 = run(0)
print('d_model 100, T = 6016')
T, E = 6016, 100
probs = [(3 * E, E, T), (E, E, T), (F, E, T), (E, F, T)] * 8
for (M, N, K) in set(probs):
    bufs[(M, N, K)] = ((torch.rand(K, M, generator=g) - 0.5).cuda(), (torch.rand(K, N, generator=g) - 0.5).cuda())
c = run(0); d2 = run(2); c2 = run(0)
print('d100 bits equal:', all(torch.equal(x, y) for x, y in zip(c, d2)))
print("max rel diff wide vs narrow:", max(float((x - y).abs().max() / x.abs().max()) for x, y in zip(a, b)), " bits equal:", all(torch.equal(x, y) for x, y in zip(a, b)))
