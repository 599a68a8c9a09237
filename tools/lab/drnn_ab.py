"""configuration-5 step (engine.DrnnEngine) with 1 / 3 generator streams and library debug modes; ms per step after 40 warm-up steps"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gan_ffn_amd import _lib, data as D, engine as E, model as M, ops
lib = _lib.load()
b = D.synthetic_batch(B=30, S_max=94, seed=3407, device="cuda")
cfgs = ((3, 0), (1, 0), (3, 14), (1, 14))
for streams, mode in cfgs[:1] + cfgs * 2:          # (the first run only warms the clocks up)
    lib.ganffn_debug_set_ffn_mode(mode)
    torch.manual_seed(3407)
    net = M.GAN_FFN_DialogueRNN(M.AcousticGenerator(100), M.VisualGenerator(100), M.TextGenerator(100), 100, 500, 500, 100, 100, 100,
                                n_classes=6, listener_state=False, context_attention="general", dropout_rec=0.1, dropout=0.6).cuda().train()
    eng = E.DrnnEngine(net, n_streams=streams)
    for _ in range(80):
        eng.step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.step(b)
    torch.cuda.synchronize()
    print("streams %d mode %2d: %.2f ms/step" % (streams, mode, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
lib.ganffn_debug_set_ffn_mode(0)
