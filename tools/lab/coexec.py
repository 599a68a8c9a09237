"""Do kernels of different HIP streams execute at the same time, or do several streams only hide the dead time between
dependent launches?  N launches of one kernel on one stream against N/2 + N/2 on two (tuned) streams, for a kernel that
leaves most of the chip idle (attention core, 8 dialogues: 240 workgroups of 128 threads), one that fills it once
(32 dialogues: 960 workgroups) and a weight-resident GEMM that fills it for its whole life (linear1, T = 6016)."""
import ctypes as C, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, ops, engine as E
lib = _lib.load()
P = ops._ptr
rng = torch.tensor([3407, 0], dtype=torch.int64, device="cuda")
gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
eng = E.GanEngine(gens, discs, n_streams=3)
from gan_ffn_amd import data as D
eng.iteration(D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda"))      # builds (and tunes) the side streams
eng.synchronize(); torch.cuda.synchronize()
s0, s1 = eng.streams[0], eng.streams[1]


def attn(B):
    S, Em, H = 94, 100, 10
    bufs = []
    for _ in range(2):
        qkv = torch.randn(S, B, 3 * Em, device="cuda"); o = torch.empty(S, B, Em, device="cuda"); lse = torch.zeros(B * H, S, device="cuda")
        bufs.append((qkv, o, lse))
    def run(i, st):
        qkv, o, lse = bufs[i]
        _lib.call("ganffn_attention_fwd", P(qkv), P(o), P(lse), S, B, Em, H, C.c_float(0.1), C.c_uint32(16), P(rng), C.c_uint64(0), C.c_void_p(st.cuda_stream))
    return run


def lin1(T):
    bufs = []
    for _ in range(2):
        bufs.append((torch.randn(T, 100, device="cuda"), torch.randn(2048, 100, device="cuda") * 0.1, torch.randn(2048, device="cuda") * 0.1, torch.empty(T, 2048, device="cuda")))
    def run(i, st):
        x, w1, b1, h = bufs[i]
        _lib.call("ganffn_ffn_linear1_fwd", P(x), P(w1), P(b1), P(h), T, 100, 2048, C.c_float(0.1), 18, P(rng), C.c_uint64(0), 1, C.c_void_p(st.cuda_stream))
    return run


def timed(run, two, n=400):
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(cur); s0.wait_event(e0); s1.wait_event(e0)
    if two:
        for _ in range(n // 2):
            run(0, s0); run(1, s1)
    else:
        for _ in range(n):
            run(0, s0)
    cur.wait_stream(s0); cur.wait_stream(s1); e1.record(cur); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, run in (("attention forward, 8 dialogues (240 workgroups)", attn(8)), ("attention forward, 32 dialogues (960 workgroups)", attn(32)),
                  ("linear1 + epilogue, T = 3008", lin1(3008)), ("linear1 + epilogue, T = 6016", lin1(6016))):
    for _ in range(2):
        timed(run, False, 100); timed(run, True, 100)
    a = min(timed(run, False) for _ in range(3)); b = min(timed(run, True) for _ in range(3))
    print("%-52s one stream %.2f us per launch, two streams %.2f us per launch (x %.2f)" % (name, a, b, a / b), flush=True)
