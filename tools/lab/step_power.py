"""Socket power and shader clock while the full 12-sub-step iteration runs in a loop (3 streams)."""
import os, sys, subprocess, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import data as D, engine as E
samples, stop = [], False
def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        sclk = [l.split("(")[-1].split("M")[0] for l in out.split("\n") if "sclk" in l and "GPU[0]" in l]
        pw = [l.split(":")[-1].strip() for l in out.split("\n") if "Power" in l and "GPU[0]" in l]
        samples.append((time.time(), sclk[0] if sclk else "?", pw[0] if pw else "?"))
        time.sleep(0.25)
gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
b = D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda")
for ns in (3, 1):
    eng = E.GanEngine(gens, discs, n_streams=ns)
    for _ in range(3):
        eng.iteration(b)
    eng.synchronize(); torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        for _ in range(10):
            eng.iteration(b)
        eng.synchronize(); torch.cuda.synchronize(); n += 10
    dt = time.time() - t0
    stop = True; th.join()
    mid = [s for s in samples if 1.0 < s[0] - t0 < 3.8]
    print("streams %d: %.2f ms/step; sclk MHz %s; power W %s" % (ns, dt / n * 1e3, [m[1] for m in mid][:8], [m[2] for m in mid][:8]), flush=True)
