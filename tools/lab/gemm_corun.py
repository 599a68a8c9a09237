"""Two streams each running the same GEMM shape back to back: how much of the per-launch prologue / epilogue idle time
does plain multi-stream co-running already recover?"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
for (M, N, K) in ((3008, 2048, 100), (3008, 2048, 512), (3008, 100, 100)):
    bufs = []
    for _ in range(2):
        bufs.append((torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.rand(N, device="cuda"), torch.empty(M, N, device="cuda")))
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    def run(n_streams, reps=50):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in s[:n_streams]:
            st.wait_event(e0)
        for i in range(reps):
            for j in range(2):
                st = s[j % n_streams]
                a, w, b, c = bufs[j]
                _lib.call("ganffn_gemm_nt", ops._ptr(a), ops._ptr(w), ops._ptr(b), ops._ptr(c), M, N, K, C.c_void_p(st.cuda_stream))
        for st in s[:n_streams]:
            torch.cuda.current_stream().wait_stream(st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (2 * reps)
    run(1, 5); run(2, 5)
    t1, t2 = run(1), run(2)
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d: one stream %.1f us/GEMM (%.1f TF), two streams %.1f us/GEMM (%.1f TF)" % (M, N, K, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6), flush=True)
