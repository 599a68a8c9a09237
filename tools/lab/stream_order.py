"""What makes a second engine of a process slow?  argv[1] = number of dummy streams created (and used once) BEFORE the engine;
then engine 1 is timed, deleted, and engine 2 is built and timed — with the per-process stream cache (GANFFN_STREAM_CACHE=1,
default) and without (=0)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import data as D, engine as E
ndummy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dummies = [torch.cuda.Stream() for _ in range(ndummy)]
x = torch.zeros(1024, device="cuda")
for s_ in dummies:
    with torch.cuda.stream(s_):
        x.add_(1.0)
torch.cuda.synchronize()
b = D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda")
for k in range(3):
    gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
    eng = E.GanEngine(gens, discs, n_streams=3)
    for _ in range(5):
        eng.iteration(b)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        eng.iteration(b)
    eng.synchronize(); torch.cuda.synchronize()
    print("dummy streams %d, cache %s, engine %d: %.2f ms/step" % (ndummy, os.environ.get("GANFFN_STREAM_CACHE", "1"), k + 1, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    del eng, gens, discs
    torch.cuda.empty_cache()
