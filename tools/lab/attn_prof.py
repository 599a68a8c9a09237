import ctypes as C, os, sys, torch
sys.path.insert(0, "/root/repo")
from gan_ffn_amd import _lib, ops
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
for (S, B, E, H) in ((94, 32, 100, 10), (94, 32, 512, 8)):
    qkv = torch.randn(S, B, 3 * E, device="cuda"); do = torch.randn(S, B, E, device="cuda"); dq = torch.empty(S, B, 3 * E, device="cuda")
    rng = torch.tensor([1, 2], dtype=torch.int64, device="cuda"); st = ops._stream()
    for it in range(3):
        _lib.call("ganffn_attention_bwd", ops._ptr(qkv), None, None, ops._ptr(do), ops._ptr(dq), S, B, E, H, C.c_float(0.1), C.c_uint32(16), ops._ptr(rng), C.c_uint64(0), st)
        torch.cuda.synchronize()
        buf = (C.c_longlong * 32)()
        raw.ganffn_lab_attn_prof(buf)
        t = list(buf)[:13]
        print("E=%d it%d total %d :" % (E, it, t[12] - t[0]), [t[i + 1] - t[i] for i in range(12)], flush=True)
