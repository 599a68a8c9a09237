"""Shader clock and socket power while the weight-gradient launch mix (bench.time_dominant_kernel) runs in a loop; rocm-smi polled from a side thread."""
import os, sys, subprocess, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
samples, stop = [], False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [l.strip() for l in out.split("\n") if ("sclk" in l or "Power" in l) and "GPU[0]" in l]
            samples.append((time.time(), s))
        except Exception as e:
            samples.append((time.time(), [repr(e)]))
        time.sleep(0.25)
th = threading.Thread(target=poll); th.start()
time.sleep(1.0)
t0 = time.time()
for rep in range(4):
    kt, fl, n = bench.time_dominant_kernel(94, 32, reps=150)
    print("t=%.1f  wgrad family %.1f us per launch (%d launches), %.1f TF" % (time.time() - t0, kt * 1e6, n, fl / kt / 1e12), flush=True)
time.sleep(0.5)
stop = True; th.join()
for t, s in samples:
    print("%.1f" % (t - t0), " | ".join(s))
