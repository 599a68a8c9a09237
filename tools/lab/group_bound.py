"""Upper bound of what batching two same-shape sibling sub-steps into ONE grouped launch chain could gain:
a d_model-100 discriminator forward + backward (with weight gradients)
  (a) twice in a row on one stream at 2B = 64 dialogues            (what a single stream does today),
  (b) two networks on two streams, 64 dialogues each               (what the multi-stream engine does today),
  (c) once at 128 dialogues on one stream                          (launch shapes of a grouped pair; same FLOPs as a/b).
The same for the generator-shaped pass at 32 / 64 dialogues."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import engine as E, ops

S = 94
gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
eng = E.GanEngine(gens, discs, n_streams=2)
dev = eng.dev
rng = eng.rng


def mk(net, B):
    p = E._Pass(net, S, B, dev, True)
    ws = torch.empty(p.n_ws, device=dev)
    x = torch.rand(S, B, net.E if not net.has_obj else 100, device=dev)
    dout = torch.rand(S, B, net.D2 if net.kind == 0 else 1, device=dev) * 1e-3
    return p, ws, x, dout


def run(net, P):
    p, ws, x, dout = P
    eng.ws = ws
    adds = eng._net_fwd(net, p, x, True, True)
    eng._net_bwd(net, p, dout, True, adds, True)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, n1, n2, B in (("D (2B=64 dialogues)", eng.D["acoustic"], eng.D["text"], 64), ("G100 (B=32)", eng.G["acoustic"], eng.G["text"], 32)):
    Pa, Pb, Pc = mk(n1, B), mk(n2, B), mk(n1, 2 * B)

    def seq():
        run(n1, Pa); run(n2, Pb)

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            run(n1, Pa)
        with torch.cuda.stream(s2):
            run(n2, Pb)
        cur.wait_stream(s1); cur.wait_stream(s2)

    def wide():
        run(n1, Pc)

    ta, tb, tc = timeit(seq), timeit(two), timeit(wide)
    print("%-22s sequential x2 %.3f ms | two streams %.3f ms | one pass at 2x dialogues %.3f ms" % (name, ta, tb, tc), flush=True)
