"""Step time with every dropout probability set to 0 (no Philox call anywhere): the upper bound of what a cheaper
counter-based generator could gain.  NOT a valid benchmark of the workload (the reference trains with dropout)."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import data as D, engine as E
for mode in ("dropout on", "dropout p=0"):
    gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
    eng = E.GanEngine(gens, discs, n_streams=int(os.environ.get("STREAMS", "3")))
    if mode != "dropout on":
        for n in list(eng.G.values()) + list(eng.D.values()):
            n.p_enc = n.p_pe = n.p_head = 0.0
    b = D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda")
    for _ in range(5):
        eng.iteration(b)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        eng.iteration(b)
    eng.synchronize(); torch.cuda.synchronize()
    print("%s: %.3f ms/step (%d streams)" % (mode, (time.perf_counter() - t0) / 30 * 1e3, eng.n_streams), flush=True)
