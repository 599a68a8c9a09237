"""Launch mix for the rocprofv3 line of the masked general2 attention at configuration 5's shape:
S = 94 steps, B = 30 dialogues, D = 200 (2 x D_e).  Run under `rocprofv3 --kernel-trace --stats`; tools/general2_line.py
turns the kernel stats into achieved GB/s against the algorithmic bytes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import ops
S, B, D = 94, 30, 200
g = torch.Generator().manual_seed(1)
M = (torch.rand(S, B, D, generator=g) - 0.5).cuda().requires_grad_(True)
X = ((torch.rand(S, B, D, generator=g) - 0.5) * 0.7).cuda().requires_grad_(True)
mask = torch.ones(B, S, device="cuda")
gy = (torch.rand(S, B, D, generator=g) - 0.5).cuda()
for _ in range(50):
    att, alpha = ops.General2AttnFn.apply(X, M, mask)
    (att * gy).sum().backward()
torch.cuda.synchronize()
print("done")
