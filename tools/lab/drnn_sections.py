"""configuration 5: where a DrnnEngine step spends its time — generator forwards | recurrence + head (forward, loss,
backward) | generator backwards + Adam — with 1 and with 3 generator streams (HIP events recorded on the step's main stream
at the two C-ABI calls that separate the sections)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import engine, model as M, ops, data as D, _lib
dev = "cuda"
batch = D.synthetic_batch(B=30, S_max=94, seed=3407, device=dev)
marks = []
orig_call = _lib.call


def call(name, *a):
    if name == "ganffn_add3":
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    r = orig_call(name, *a)
    if name == "ganffn_seq_reverse" and a[-2] == 1:
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    return r


_lib.call = call
engine._lib.call = call
for ns in (1, 3, 1, 3):
    torch.manual_seed(3407)
    net = M.GAN_FFN_DialogueRNN(M.AcousticGenerator(100), M.VisualGenerator(100), M.TextGenerator(100), 100, 500, 500, 100, 100, 100,
                                n_classes=6, listener_state=False, context_attention="general", dropout_rec=0.1, dropout=0.6).to(dev).train()
    ops.manual_seed(1, dev)
    eng = engine.DrnnEngine(net, n_streams=ns)
    for _ in range(40):
        eng.step(batch)
    torch.cuda.synchronize()
    tot = [0.0, 0.0, 0.0]
    n = 20
    for _ in range(n):
        del marks[:]
        main = eng.streams[1] if eng.streams is not None else torch.cuda.current_stream()
        e0, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        eng.step(batch)
        e3.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        tot[0] += e0.elapsed_time(marks[0]); tot[1] += marks[0].elapsed_time(marks[1]); tot[2] += marks[1].elapsed_time(e3)
    print("%d stream(s): generators forward %.3f ms | recurrence + head %.3f ms | generators backward + Adam %.3f ms | sum %.3f ms" % (
        ns, tot[0] / n, tot[1] / n, tot[2] / n, sum(tot) / n), flush=True)
