"""A/B of the d_model-100 attention backward (train mode, S = 94): the whole-problem kernel (320 / 640 six-wave workgroups; Philox
re-evaluated or the forward's keep words) against the key-split kernel of round 5 (960 / 1920 two-wave workgroups, partial dQ slabs;
keep words or, with mode bit 29, Philox).  Interleaved repeats, HIP events around 300 back-to-back launches each."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
def timeit(fn, reps=300):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
P, st = ops._ptr, ops._stream()
for (S, B, E, H) in ((94, 32, 100, 10), (94, 64, 100, 10), (110, 32, 100, 10), (60, 32, 100, 10)):
    qkv = torch.randn(S, B, 3 * E, device="cuda"); do = torch.randn(S, B, E, device="cuda")
    o = torch.empty(S, B, E, device="cuda"); lse = torch.zeros(B * H, S, device="cuda"); dq = torch.empty(S, B, 3 * E, device="cuda")
    slabs = torch.empty(3, S, B, E, device="cuda"); n = C.c_int(0)
    keep = torch.zeros(int(lib.ganffn_attention_keep_words(B, H)), dtype=torch.int32, device="cuda")
    p = 0.1
    fk = lambda: _lib.call("ganffn_attention_fwd_keep", P(qkv), P(o), P(lse), P(keep), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
    b = lambda: _lib.call("ganffn_attention_bwd", P(qkv), P(o), P(lse), P(do), P(dq), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
    bk = lambda: _lib.call("ganffn_attention_bwd_keep", P(qkv), P(o), P(lse), P(do), P(keep), P(dq), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
    bs = lambda: _lib.call("ganffn_attention_bwd_split", P(qkv), P(o), P(lse), P(do), P(keep), P(dq), P(slabs), C.c_int64(S * B * E), C.byref(n), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
    bsp = lambda: _lib.call("ganffn_attention_bwd_split", P(qkv), P(o), P(lse), P(do), None, P(dq), P(slabs), C.c_int64(S * B * E), C.byref(n), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
    fk()
    for _ in range(4):
        print("S=%d B=%d: whole philox %.2f keep %.2f | split keep %.2f philox %.2f us (%d parts) | fwd+keep %.2f" %
              (S, B, timeit(b), timeit(bk), timeit(bs), timeit(bsp), n.value, timeit(fk)), flush=True)
