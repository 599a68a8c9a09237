"""Is the step host-bound?  Time the host's enqueue loop separately from the device completion."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import data as D, engine as E
gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
for ns in (1, 3):
    eng = E.GanEngine(gens, discs, n_streams=ns)
    b = D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda")
    for _ in range(3):
        eng.iteration(b)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.iteration(b)
    t1 = time.perf_counter()
    eng.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("streams %d: host enqueue %.2f ms/iter, total %.2f ms/iter" % (ns, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3), flush=True)
for ns in (1, 3):
    eng = E.GanEngine(gens, discs, n_streams=ns)
    for _ in range(3):
        eng.iteration(b)
    eng.synchronize(); torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        eng.iteration(b)
        t1 = time.perf_counter()
        eng.synchronize(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append(t1 - t0); ts.append(t2 - t0)
    print("streams %d, one iteration from an idle GPU: host enqueue %.2f ms (min %.2f), until done %.2f ms" % (ns, sum(hs) / 10 * 1e3, min(hs) * 1e3, sum(ts) / 10 * 1e3), flush=True)
