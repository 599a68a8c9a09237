"""isolated timing of the d_model-100 attention core (head_dim 10, S = 94) at B = 32 and 64 dialogues, forward / backward."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
def timeit(fn, reps=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
for (S, B, E, H) in ((94, 32, 100, 10), (94, 64, 100, 10), (94, 96, 100, 10), (33, 64, 100, 10)):
    qkv = torch.randn(S, B, 3 * E, device="cuda"); do = torch.randn(S, B, E, device="cuda")
    o = torch.empty(S, B, E, device="cuda"); lse = torch.zeros(B * H, S, device="cuda"); dq = torch.empty(S, B, 3 * E, device="cuda")
    f = lambda: _lib.call("ganffn_attention_fwd", ops._ptr(qkv), ops._ptr(o), ops._ptr(lse), S, B, E, H, C.c_float(0.1), C.c_uint32(16), ops._ptr(rng), C.c_uint64(0), ops._stream())
    b = lambda: _lib.call("ganffn_attention_bwd", ops._ptr(qkv), ops._ptr(o), ops._ptr(lse), ops._ptr(do), ops._ptr(dq), S, B, E, H, C.c_float(0.1), C.c_uint32(16), ops._ptr(rng), C.c_uint64(0), ops._stream())
    print("S=%d B=%d E=%d H=%d: fwd %.1f us, bwd %.1f us" % (S, B, E, H, timeit(f), timeit(b)), flush=True)
