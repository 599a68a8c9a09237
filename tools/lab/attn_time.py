"""isolated timing of the d_model-100 attention core (head_dim 10, S = 94) at B = 32 and 64 dialogues: forward in eval and
train mode, backward in eval mode, train mode with recomputed Philox masks, train mode with the forward's saved keep words"""
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
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
P, st = ops._ptr, ops._stream()
for (S, B, E, H) in ((94, 32, 100, 10), (94, 64, 100, 10), (94, 96, 100, 10), (33, 64, 100, 10)):
    qkv = torch.randn(S, B, 3 * E, device="cuda"); do = torch.randn(S, B, E, device="cuda")
    o = torch.empty(S, B, E, device="cuda"); lse = torch.zeros(B * H, S, device="cuda"); dq = torch.empty(S, B, 3 * E, device="cuda")
    has_keep = hasattr(lib, "ganffn_attention_keep_words")       # (an older build loaded through GANFFN_LIB has no keep pair)
    keep = torch.zeros(int(lib.ganffn_attention_keep_words(B, H)) if has_keep else 4, dtype=torch.int32, device="cuda")
    res = []
    for p in (0.0, 0.1):
        f = lambda: _lib.call("ganffn_attention_fwd", P(qkv), P(o), P(lse), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
        fk = lambda: _lib.call("ganffn_attention_fwd_keep", P(qkv), P(o), P(lse), P(keep), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
        b = lambda: _lib.call("ganffn_attention_bwd", P(qkv), P(o), P(lse), P(do), P(dq), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
        bk = lambda: _lib.call("ganffn_attention_bwd_keep", P(qkv), P(o), P(lse), P(do), P(keep), P(dq), S, B, E, H, C.c_float(p), C.c_uint32(16), P(rng), C.c_uint64(0), st)
        res.append("p=%.1f: fwd %.1f (keep %.1f) us, bwd %.1f (keep %.1f) us" % (p, timeit(f), timeit(fk) if has_keep else -1, timeit(b), timeit(bk) if has_keep else -1))
    print("S=%d B=%d E=%d H=%d: %s" % (S, B, E, H, " | ".join(res)), flush=True)
