"""head_dim 60 / 64 attention through the 16x16x4 kernels (attention16.hip) against attention.hip.  Needs a lab build that
exports `ganffn_lab_attn16_big(int)` and lifts the S <= 48 rule; the numbers are recorded at attn16_supported()."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
for (S, B, E, H) in ((94, 32, 512, 8), (33, 32, 600, 10), (94, 32, 600, 10)):
    qkv = torch.randn(S, B, 3 * E, device="cuda"); do = torch.randn(S, B, E, device="cuda")
    res = []
    for big in (0, 1):
        lib.ganffn_lab_attn16_big(big)
        o = torch.empty(S, B, E, device="cuda"); lse = torch.zeros(B * H, S, device="cuda"); dq = torch.empty(S, B, 3 * E, device="cuda")
        f = lambda: _lib.call("ganffn_attention_fwd", ops._ptr(qkv), ops._ptr(o), ops._ptr(lse), S, B, E, H, C.c_float(0.1), C.c_uint32(16), ops._ptr(rng), C.c_uint64(0), ops._stream())
        b = lambda: _lib.call("ganffn_attention_bwd", ops._ptr(qkv), ops._ptr(o), ops._ptr(lse), ops._ptr(do), ops._ptr(dq), S, B, E, H, C.c_float(0.1), C.c_uint32(16), ops._ptr(rng), C.c_uint64(0), ops._stream())
        uf = timeit(f); ub = timeit(b)
        res.append((o.clone(), dq.clone()))
        print("S=%d B=%d E=%d H=%d attn16=%d: fwd %.1f us, bwd %.1f us" % (S, B, E, H, big, uf, ub), flush=True)
    print("   max |o diff| %.2e  max |dqkv diff| %.2e (scale %.2e)" % (float((res[0][0]-res[1][0]).abs().max()), float((res[0][1]-res[1][1]).abs().max()), float(res[0][1].abs().max())))
