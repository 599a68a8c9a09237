"""Duration of each of the 12 sub-steps run alone (single stream, device-synchronised around each), and the length of
the chains the 3-stream map creates."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import data as D, engine as E
gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
eng = E.GanEngine(gens, discs, n_streams=1)
b = D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda")
for _ in range(3):
    eng.iteration(b)
torch.cuda.synchronize()
tot = [0.0] * 12
REP = 5
for _ in range(REP):
    eng._adds = 0
    for i, (kind, who, partner) in enumerate(E.SCHEDULE):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        (eng.train_disc if kind == "D" else eng.train_gen)(who, partner, b, i)
        torch.cuda.synchronize(); tot[i] += time.perf_counter() - t0
    eng._base_add += eng._adds
ms = [t / REP * 1e3 for t in tot]
for i, (k, w, p) in enumerate(E.SCHEDULE):
    print("sub-step %2d  %s %-8s vs %-8s  %6.2f ms" % (i, k, w, p, ms[i]))
print("sum %.2f ms" % sum(ms))
smap = E.STREAM_MAP[3]
for s in range(3):
    print("stream %d: sub-steps %s = %.2f ms" % (s, [i for i in range(12) if smap[i] == s], sum(ms[i] for i in range(12) if smap[i] == s)))
