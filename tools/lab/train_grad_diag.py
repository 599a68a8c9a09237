"""Print per-parameter gradient error statistics of one train-mode pass vs the oracle (diagnostic, GPU only)."""
import os, sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import formula as F_
from util import NETS, formula_sd
from oracle import ganffn_oracle as O
from gan_ffn_amd import ops
import test_hip_modules as M

cls_name, din, S, B = "AcousticDiscriminator", 100, 94, 4
kind, _, E, H, fcs, has_obj = NETS[cls_name]
net = M.build(cls_name).train()
seed = 424242
ops.manual_seed(seed)
tag = "train.%s" % cls_name
x_np = F_.formula_input(tag, S, B, din, pad_from=max(1, S - 4))
x = torch.from_numpy(x_np).cuda().requires_grad_(True)
y = net(x)
gy = (torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1])) - 0.5)
(y * gy.cuda()).sum().backward()
onet = O.OracleNet(kind, formula_sd(cls_name), H, 0.2, torch.float64)
xo = torch.from_numpy(x_np).double().requires_grad_(True)
h = O.encoder_stack(xo, onet.P, H, O.Rng(seed, 0, True))
r1 = O.Rng(seed, 1, True)
t = O.gelu(h)
t = O.gelu(O._drop(t @ onet.P["fc1.weight"].T + onet.P["fc1.bias"], 0.2, O.SITE_HEAD1, r1))
t = O.gelu(O._drop(t @ onet.P["fc2.weight"].T + onet.P["fc2.bias"], 0.2, O.SITE_HEAD2, r1))
yo = torch.sigmoid(O._drop(t @ onet.P["fc3.weight"].T + onet.P["fc3.bias"], 0.2, O.SITE_HEAD3, r1))
(yo * gy.double()).sum().backward()

def stat(label, a, b, rtol):
    scale = max(np.abs(b).max(), 1e-30); err = np.abs(a - b)
    print("%-60s max rel %.2e  frac>%g: %.3f%%" % (label, err.max() / scale, rtol, 100 * (err > rtol * scale).mean()), flush=True)
stat("out", y.detach().cpu().double().numpy(), yo.detach().numpy(), 1e-4)
stat("dx", x.grad.cpu().double().numpy(), xo.grad.numpy(), 2e-4)
for k, p in net.named_parameters():
    if p.grad is None or k not in onet.P or onet.P[k].grad is None: continue
    if "layers.0." in k or "layers.7." in k or "layers.4." in k or k.startswith("fc"):
        stat(k, p.grad.cpu().double().numpy(), onet.P[k].grad.numpy(), 1e-3)
