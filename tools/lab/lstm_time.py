"""N4: the MELD classifier's 4-layer bidirectional LSTM (D_m = 600, D_e = 300; train_MELD.py:143-151) forward + backward at
(S, B) = (33, 32): the build's kernels (csrc/lstm.hip via ops.lstm_forward) against the device library's nn.LSTM (MIOpen), HIP
events around 20 iterations each, interleaved."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import ops
torch.manual_seed(0)
for (S, B) in ((33, 32), (94, 32)):
    lstm = torch.nn.LSTM(600, 300, num_layers=4, bidirectional=True, dropout=0.5).cuda().train()
    x = torch.randn(S, B, 600, device="cuda", requires_grad=True)
    gy = torch.randn(S, B, 600, device="cuda")
    def ours():
        y = ops.lstm_forward(x, lstm, True); (y * gy).sum().backward()
    def lib():
        y, _ = lstm(x); (y * gy).sum().backward()
    def timeit(fn, reps=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    for _ in range(3):
        print("S=%d B=%d  lstm.hip %.3f ms | nn.LSTM (MIOpen) %.3f ms   (forward + backward, 4 layers x 2 directions, train mode)" % (S, B, timeit(ours), timeit(lib)), flush=True)
