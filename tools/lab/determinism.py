import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import test_hip_modules as M
from gan_ffn_amd import data as D
b = D.synthetic_batch(B=32, S_max=94, seed=3407, device="cuda")
for cls, key in (("AcousticGenerator", "text"), ("TextDiscriminator", "text"), ("VisualGenerator", "visual")):
    net = M.build(cls).eval()
    x = b[key]
    with torch.no_grad():
        y1 = net(x); y2 = net(x)
        perm = torch.randperm(32, generator=torch.Generator().manual_seed(1)).cuda()
        yp = net(x[:, perm].contiguous())
    d12 = (y1 != y2)
    dp = (yp != y1[:, perm])
    print(cls, "same-input reruns differ:", int(d12.sum()), "| permuted differ:", int(dp.sum()), "of", y1.numel(),
          "max abs", float((yp - y1[:, perm]).abs().max()))
    if int(dp.sum()):
        idx = dp.nonzero()
        print("  differing (s, b_in_perm_order) sample:", idx[:8].tolist(), "distinct s:", len(set(idx[:, 0].tolist())), "distinct b:", len(set(idx[:, 1].tolist())))
