"""Can the DialogueRNN head run as a captured graph (torch.cuda.make_graphed_callables)?  Times eager vs graphed."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_ffn_amd import dialogue_rnn as DR, data as D
torch.manual_seed(3)
class Head(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.bi = DR.BiModel(100, 500, 500, 100, 100, n_classes=6, context_attention="general", dropout_rec=0.1, dropout=0.6)
    def forward(self, fusion, qmask, umask):
        return self.bi(fusion, qmask, umask)[0]
head = Head().cuda().train()
b = D.synthetic_batch(B=30, S_max=94, seed=5, device="cuda")
fusion = torch.randn(94, 30, 100, device="cuda", requires_grad=True)
qmask, umask = b["qmask"], b["umask"]
def run(mod, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        lp = mod(fusion, qmask, umask)
        lp.sum().backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(head, 1)
print("eager head fwd+bwd: %.1f ms" % run(head), flush=True)
t0 = time.perf_counter()
g = torch.cuda.make_graphed_callables(head, (fusion, qmask, umask))
torch.cuda.synchronize()
print("capture took %.1f s" % (time.perf_counter() - t0), flush=True)
run(g, 1)
print("graphed head fwd+bwd: %.1f ms" % run(g), flush=True)
