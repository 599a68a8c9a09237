"""What the linear1 / linear2-dgrad launches of the d_model-100 feed-forward block cost by MODE (ganffn_ffn_k100_hook, T = 3008 /
6016): train (Philox + 1-bit pattern), eval with the pattern, eval without — the difference train - eval is what the dropout
decisions cost inside gemm_wres_kernel<0,1,100>; and the dgrad <1,3,100> from pattern bits / from the saved activation."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
def timeit(fn, reps=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
P, st = ops._ptr, ops._stream()
rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
E, F = 100, 2048
for T in (3008, 6016):
    x = torch.rand(T, E, device="cuda") - 0.5
    w1, b1 = (torch.rand(F, E, device="cuda") - 0.5) * 0.2, torch.zeros(F, device="cuda")
    w2 = (torch.rand(E, F, device="cuda") - 0.5) * 0.2
    h = torch.empty(T, F, device="cuda"); hm = torch.zeros(((T + 31) // 32) * F, device="cuda")
    def k(which, train, mask, hs=None):
        return lambda: _lib.call("ganffn_ffn_k100_hook", which, P(x), P(w2 if which else w1), None if which else P(b1), P(h), P(hm) if mask else None,
                                 P(hs) if hs is not None else None, T, C.c_float(0.1), C.c_uint32(18), P(rng), C.c_uint64(0), train, st)
    hs = torch.relu(torch.rand(T, F, device="cuda") - 0.5)
    for _ in range(3):
        print("T=%d linear1: train+bits %.2f | eval+bits %.2f | eval %.2f | train no bits %.2f || dgrad bits %.2f | dgrad saved-h %.2f us" %
              (T, timeit(k(0, 1, True)), timeit(k(0, 0, True)), timeit(k(0, 0, False)), timeit(k(0, 1, False)), timeit(k(1, 1, True)), timeit(k(1, 1, False, hs))), flush=True)
