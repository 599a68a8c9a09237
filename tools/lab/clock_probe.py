"""Shader clock under load: run one GEMM shape (or the full step) in a loop and poll rocm-smi from a side thread."""
import os, sys, subprocess, threading, time, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import _lib, ops
lib = _lib.load()
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [l.strip() for l in out.split("\n") if ("sclk" in l or "Power" in l or "mclk" in l) and "GPU[0]" in l]
            samples.append((time.time(), s))
        except Exception as e:
            samples.append((time.time(), [repr(e)]))
        time.sleep(0.3)
th = threading.Thread(target=poll); th.start()
time.sleep(1.0)
M, N, K = 3008, 2048, 512
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.rand(N, device="cuda"); c = torch.empty(M, N, device="cuda")
st = ops._stream()
t0 = time.time()
n = 0
while time.time() - t0 < 4.0:
    for _ in range(200):
        _lib.call("ganffn_gemm_nt", ops._ptr(a), ops._ptr(w), ops._ptr(b), ops._ptr(c), M, N, K, st)
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
print("GEMM loop: %.1f us/GEMM, %.1f TF" % (dt / n * 1e6, 2.0 * M * N * K * n / dt / 1e12))
time.sleep(0.5)
stop = True; th.join()
for t, s in samples:
    print("%.1f" % (t - t0), " | ".join(s))
