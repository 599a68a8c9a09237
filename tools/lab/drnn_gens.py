"""configuration 5: the three generators' forward (and backward + Adam) passes of one DrnnEngine step, on one stream against
the engine's three tuned streams, in isolation (HIP events, 30 repetitions): what could concurrent generator passes buy?"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import engine, model as M, ops, data as D
torch.manual_seed(3407)
dev = "cuda"
net = M.GAN_FFN_DialogueRNN(M.AcousticGenerator(100), M.VisualGenerator(100), M.TextGenerator(100), 100, 500, 500, 100, 100, 100,
                            n_classes=6, listener_state=False, context_attention="general", dropout_rec=0.1, dropout=0.6).to(dev).train()
ops.manual_seed(1, dev)
batch = D.synthetic_batch(B=30, S_max=94, seed=3407, device=dev)
eng = engine.DrnnEngine(net, n_streams=3)
eng.step(batch)                       # buffers + stream tuning
torch.cuda.synchronize()
keys = ("acoustic", "visual", "text")
S, B = batch["text"].shape[:2]
d_fusion = torch.randn(S, B, 100, device=dev) * 1e-3


def run(multi, bwd, host_only=False):
    cur = torch.cuda.current_stream()
    adds = {}
    fork = torch.cuda.Event(); fork.record(cur)
    for i, k in enumerate(keys):
        eng.ws = eng.ws3[k]
        st = eng.streams[i] if multi else cur
        if multi:
            st.wait_event(fork)
        with torch.cuda.stream(st):
            adds[k] = eng._net_fwd(eng.G[k], eng.pass_G[k], batch[k], train=True, save=True, adds=(2 * i, 2 * i + 1))
            if bwd:
                eng.G[k].grad.zero_()
                eng._net_bwd(eng.G[k], eng.pass_G[k], d_fusion, True, (eng._base_add + 2 * i, eng._base_add + 2 * i + 1), True, None)
                eng._adam(eng.G[k])
    if multi:
        for st in eng.streams:
            cur.wait_stream(st)


for bwd in (False, True):
    for multi in (False, True, False, True):
        for _ in range(5):
            run(multi, bwd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(30):
            run(multi, bwd)
        e1.record()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("%s, %s: %.3f ms per round on the device, host enqueue %.3f ms" % ("fwd+bwd+Adam" if bwd else "fwd only", "3 streams" if multi else "1 stream ",
                                                                            e0.elapsed_time(e1) / 30, th / 30 * 1e3), flush=True)
