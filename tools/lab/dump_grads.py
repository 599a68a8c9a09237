"""Dump dx of one deterministic train-mode pass (diagnostic)."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import formula as F_
from gan_ffn_amd import ops
import test_hip_modules as M
net = M.build("AcousticDiscriminator").train()
ops.manual_seed(424242)
S, B, E = 94, 4, 100
x = torch.from_numpy(F_.formula_input("train.AcousticDiscriminator", S, B, E, pad_from=S - 4)).cuda().requires_grad_(True)
y = net(x)
gy = (torch.from_numpy(F_.formula_input("grad.train.AcousticDiscriminator", S, B, y.shape[-1])) - 0.5)
(y * gy.cuda()).sum().backward()
np.save(sys.argv[1], x.grad.cpu().numpy())
