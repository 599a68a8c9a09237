"""Attention forward (hd = 10): query tiles per workgroup 1 / 2 / 3 / 6 at 320 and 640 problems.  Needs a lab build of
attention16.hip that exports `ganffn_lab_attn_wpb(int)` to force the cut (the release library picks 2 below 512 problems,
the whole problem otherwise; the measurement is recorded in the launcher's comment)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
def timeit(fn, reps=100):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
S, E, H = 94, 100, 10
rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda")
for B in (32, 64, 30):
    qkv = torch.randn(S, B, 3 * E, device="cuda"); o = torch.empty(S, B, E, device="cuda"); lse = torch.empty(B * H, S, device="cuda")
    ref = None
    for wpb in (6, 3, 2, 1):
        lib.ganffn_lab_attn_wpb(wpb)
        us = timeit(lambda: _lib.call("ganffn_attention_fwd", ops._ptr(qkv), ops._ptr(o), ops._ptr(lse), S, B, E, H, C.c_float(0.1),
                                      C.c_uint32(16), ops._ptr(rng), C.c_uint64(0), ops._stream()))
        if ref is None: ref = (o.clone(), lse.clone())
        same = torch.equal(o, ref[0]) and torch.equal(lse, ref[1])
        print("B=%d  %d query tiles per workgroup: %5.1f us  bits equal to whole-problem workgroups: %s" % (B, wpb, us, same), flush=True)
