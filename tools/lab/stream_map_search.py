"""Search over sub-step -> stream maps of the 3-stream GAN step (pairs (train_disc, train_gen) stay together: the second
depends on the first).  Prints ms per iteration for every distinct assignment of the 6 pairs to 3 streams."""
import itertools, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import engine, ops, data as D
gens, discs = engine.build_networks(device="cuda", seed=3407)
batch = D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda")
seen, res = set(), []
maps = []
for a in itertools.product(range(3), repeat=6):
    # canonical relabelling (streams are interchangeable)
    relabel, canon = {}, []
    for x in a:
        relabel.setdefault(x, len(relabel)); canon.append(relabel[x])
    if tuple(canon) in seen or len(set(canon)) < 3:
        continue
    seen.add(tuple(canon)); maps.append(canon)
only = int(sys.argv[1]) if len(sys.argv) > 1 else len(maps)
import random
random.Random(1).shuffle(maps)
maps = [[0, 1, 0, 1, 2, 2]] + maps[:only]          # the current default first
for m in maps:
    full = [s for p in m for s in (p, p)]
    os.environ["GANFFN_STREAM_MAP"] = ",".join(map(str, full))
    ops.manual_seed(1)
    eng = engine.GanEngine(gens, discs, n_streams=3)
    for _ in range(2):
        eng.iteration(batch)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        eng.iteration(batch)
    eng.synchronize(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    res.append((ms, full))
    print("%.2f ms  %s" % (ms, full), flush=True)
    del eng
res.sort()
print("best:", res[:5])
