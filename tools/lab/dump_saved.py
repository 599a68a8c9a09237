"""Dump encoder forward saved buffers for one deterministic train-mode pass (diagnostic)."""
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
x = torch.from_numpy(F_.formula_input("train.AcousticDiscriminator", S, B, E, pad_from=S - 4)).cuda()
cfg = ops.enc_cfg(S, B, E, net.nhead, net.num_layers, train=True, p_pe=0.1, p_enc=0.1)
n_saved, n_ws = ops.enc_sizes(cfg)
saved = torch.zeros(n_saved, device="cuda"); ws = torch.zeros(n_ws, device="cuda"); out = torch.empty(S, B, E, device="cuda")
rng = ops.DeviceRng.get(x.device)
ops.encoder_fwd_raw(cfg, x, net.position_encoding.pe, net._slab, out, saved, ws, rng.state, 0)
torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([out.flatten().cpu().numpy(), saved.cpu().numpy()]))
print("saved", n_saved, "ws", n_ws)
