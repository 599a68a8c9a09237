"""A/B of library debug modes (ganffn_debug_set_ffn_mode bits) on isolated d_model-100 passes and on the whole step:
forward + backward (with weight gradients) of a discriminator at 2B = 64 dialogues and of a generator at B = 32, S = 94,
single stream, then the 3-stream iteration.  Usage: mode_ab.py [modes...]   (default: 0 2)"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, data as D, engine as E

modes = [int(m) for m in sys.argv[1:]] or [0, 2]
S = 94
lib = _lib.load()
gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
eng = E.GanEngine(gens, discs, n_streams=3)
dev = eng.dev
batch = D.synthetic_batch(B=32, S_max=S, seed=3407, device="cuda")


def mk(net, B):
    p = E._Pass(net, S, B, dev, True)
    ws = torch.empty(p.n_ws, device=dev)
    x = torch.rand(S, B, 100, device=dev)
    dout = torch.rand(S, B, net.D2 if net.kind == 0 else 1, device=dev) * 1e-3
    return p, ws, x, dout


def run(net, P):
    p, ws, x, dout = P
    eng.ws = ws
    adds = eng._net_fwd(net, p, x, True, True)
    eng._net_bwd(net, p, dout, True, adds, True)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


PD, PG = mk(eng.D["acoustic"], 64), mk(eng.G["acoustic"], 32)
for rep in range(2):
    for m in modes:
        lib.ganffn_debug_set_ffn_mode(m)
        td = timeit(lambda: run(eng.D["acoustic"], PD))
        tg = timeit(lambda: run(eng.G["acoustic"], PG))

        def it():
            eng.iteration(batch)
        for _ in range(3):
            it()
        eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            it()
        eng.synchronize(); torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / 20 * 1e3
        print("mode %d: D pass (64 dialogues) fwd+bwd %.3f ms | G100 pass (32) %.3f ms | 3-stream step %.3f ms" % (m, td, tg, ts), flush=True)
lib.ganffn_debug_set_ffn_mode(0)
