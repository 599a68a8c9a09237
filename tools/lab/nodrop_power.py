"""Step time, shader clock and socket power with dropout on and with every dropout probability 0, each in its own engine,
in both orders (is the p = 0 slowdown of nodrop_bound.py a property of the data — more non-zeros, more power, lower clock —
or of being the second engine of the process?)."""
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
b = D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda")
for mode in ("p=0", "dropout on", "p=0", "dropout on"):
    gens, discs = E.build_networks(100, 0.2, "cuda", seed=3407)
    eng = E.GanEngine(gens, discs, n_streams=3)
    if mode == "p=0":
        for n in list(eng.G.values()) + list(eng.D.values()):
            n.p_enc = n.p_pe = n.p_head = 0.0
    for _ in range(5):
        eng.iteration(b)
    eng.synchronize(); torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 3.0:
        for _ in range(10):
            eng.iteration(b)
        eng.synchronize(); torch.cuda.synchronize(); n += 10
    dt = time.time() - t0
    stop = True; th.join()
    mid = [s for s in samples if 0.8 < s[0] - t0 < 2.8]
    print("%s: %.2f ms/step; sclk MHz %s; power W %s" % (mode, dt / n * 1e3, [m[1] for m in mid][:6], [m[2] for m in mid][:6]), flush=True)
    del eng, gens, discs
    torch.cuda.empty_cache()
