"""Configuration 5 step time: GAN_FFN_DialogueRNN forward + MaskedNLLLoss + backward + Adam, B = 30, S = 94."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import model as M, data as D, ops
torch.manual_seed(3)
net = M.GAN_FFN_DialogueRNN(M.AcousticGenerator(100), M.VisualGenerator(100), M.TextGenerator(100), 100, 500, 500, 100, 100, 100,
                            n_classes=6, listener_state=False, context_attention="general", dropout_rec=0.1, dropout=0.6).cuda().train()
b = D.synthetic_batch(B=30, S_max=94, seed=5, device="cuda")
w = torch.tensor([1.2, 0.60072, 0.38066, 0.94019, 0.67924, 0.34332], device="cuda")
loss_fn = M.MaskedNLLLoss(w)
opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=1e-5)
ops.manual_seed(1)
def step():
    opt.zero_grad()
    lp = net(b["acoustic"], b["visual"], b["text"], b["qmask"], b["umask"])[0]
    loss = loss_fn(lp.transpose(0, 1).contiguous().view(-1, 6), b["label"].view(-1), b["umask"])
    loss.backward(); opt.step()
    return loss
for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("config 5 step: %.1f ms (%.0f utterances/s), loss %.4f" % (dt * 1e3, float(b["umask"].sum()) / dt, float(l)))
# generators alone (forward + backward) for scale
x = [b[k].clone().requires_grad_(True) for k in ("acoustic", "visual", "text")]
def gens():
    y = net.acoustic_generator(x[0]) + net.visual_generator(x[1]) + net.text_generator(x[2])
    y.sum().backward()
gens(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    gens()
torch.cuda.synchronize()
print("three generators fwd+bwd alone: %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
