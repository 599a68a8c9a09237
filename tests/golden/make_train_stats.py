"""Training-level (statistical) parity: the GAN phase and the phase-2 classifier trained with dropout ON for several seeds
on a LEARNABLE synthetic IEMOCAP-schema set, once on the host with stock PyTorch modules (oracle/stock_modules.py — the
checker) and once on the GPU with the HIP engines; the per-metric mean and spread over seeds are compared
(tests/test_hip_train_stats.py).  This is the only end-metric evidence available without the IEMOCAP pickle
(/root/reference/README.md:9-31 needs it; SURVEY.md §7: training-level parity is statistical only).

    python tests/golden/make_train_stats.py cpu   # here, no GPU: writes tests/golden/train_stats.npz (the committed fixture)
    python tests/golden/make_train_stats.py hip   # on the GPU box: returns the same table for the HIP path (the test calls run())

Protocol (the reference's recipe, /root/reference/train_IEMOCAP.py:255-393, 595-607, 629-691, shrunk to minutes):
 * data: 16 training / 8 test dialogues of 8-20 utterances; every utterance has a label in 0..5 and its three modality
   vectors are 0.55 x a fixed class prototype + 0.45 x uniform noise (so the label is learnable from any modality); fixed,
   seed-independent;
 * per seed: the six networks get their default initialisation under torch.manual_seed(seed) on stock modules and the HIP
   modules LOAD that state_dict (identical start); dropout streams differ by construction (torch CPU generator / Philox);
 * GAN phase: N_GAN iterations of the 12-sub-step schedule over the two training batches of 8 (lr 1e-4 / 1.1e-4 / 0.5e-4,
   betas (0.5, 0.6)); the six losses are recorded at fixed iterations;
 * phase 2: log_softmax(fc(G_a + G_v + G_t)), MaskedNLLLoss with the reference's class weights, Adam(lr 1e-4,
   weight_decay 0.008) for N_P2 steps — long enough for the loss to fall from 1.8 to ~0.3 and the test accuracy to leave the
   one-class plateau, short enough that seeds still differ; train loss at fixed steps, then test loss / accuracy / weighted
   F1 in eval mode.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p_ in (ROOT, HERE):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

SEEDS = [11, 23, 37, 41, 59, 67, 73, 89]
N_GAN, GAN_AT = 16, (1, 4, 8, 12, 16)
N_P2, P2_AT = 100, (1, 20, 40, 60, 80, 100)     # (the set is learned between steps ~40 and ~120: the transition region)
B, N_TRAIN, N_TEST, LMIN, LMAX = 8, 16, 8, 8, 20
DIMS = {"text": 100, "visual": 512, "acoustic": 100}
CLASS_WEIGHTS = [1.2, 0.60072, 0.38066, 0.94019, 0.67924, 0.34332]          # train_IEMOCAP.py:653
LOSS_NAMES = ["acoustic_G_loss", "visual_G_loss", "text_G_loss", "visual_D_loss", "text_D_loss", "acoustic_D_loss"]
SCHEDULE = [("D", "visual", "acoustic"), ("G", "acoustic", "visual"), ("D", "visual", "text"), ("G", "text", "visual"),
            ("D", "text", "acoustic"), ("G", "acoustic", "text"), ("D", "acoustic", "text"), ("G", "text", "acoustic"),
            ("D", "text", "visual"), ("G", "visual", "text"), ("D", "acoustic", "visual"), ("G", "visual", "acoustic")]


def metric_names():
    n = ["gan/it%d/%s" % (it, k) for it in GAN_AT for k in LOSS_NAMES]
    n += ["p2/train_loss/step%d" % s for s in P2_AT]
    n += ["p2/test_loss", "p2/test_acc", "p2/test_f1"]
    return n


def make_data():
    """fixed learnable set -> (train batches [2 x dict], test batch dict), CPU tensors, schema of gan_ffn_amd.data.synthetic_batch"""
    rng = np.random.default_rng(20261004)
    protos = {k: rng.random((6, d)) for k, d in DIMS.items()}

    def dialogues(n):
        out = []
        for _ in range(n):
            L = int(rng.integers(LMIN, LMAX + 1))
            lab = rng.integers(0, 6, L)
            feats = {k: 0.55 * protos[k][lab] + 0.45 * rng.random((L, d)) for k, d in DIMS.items()}
            out.append((L, lab, feats, rng.integers(0, 2, L)))
        return out

    def batch(ds):
        S, n = max(d[0] for d in ds), len(ds)
        b = {k: torch.zeros(S, n, d) for k, d in DIMS.items()}
        b["umask"], b["label"], b["qmask"] = torch.zeros(n, S), torch.zeros(n, S, dtype=torch.long), torch.zeros(S, n, 2)
        for j, (L, lab, feats, spk) in enumerate(ds):
            for k in DIMS:
                b[k][:L, j] = torch.from_numpy(feats[k]).float()
            b["umask"][j, :L] = 1
            b["label"][j, :L] = torch.from_numpy(lab)
            b["qmask"][torch.arange(L), j, torch.from_numpy(spk)] = 1
        return b
    tr, te = dialogues(N_TRAIN), dialogues(N_TEST)
    return [batch(tr[i:i + B]) for i in range(0, N_TRAIN, B)], batch(te)


def weighted_f1(labels, preds, masks):
    from sklearn.metrics import accuracy_score, f1_score
    return (accuracy_score(labels, preds, sample_weight=masks) * 100, f1_score(labels, preds, sample_weight=masks, average="weighted") * 100)


def masked_nll(log_prob, label, umask, w):
    """/root/reference/model.py:74-81 on (S, B, C) log-probabilities (batch-major flattening as train_IEMOCAP.py:154)"""
    lp = log_prob.transpose(0, 1).reshape(-1, log_prob.shape[2])
    y, m = label.reshape(-1), umask.reshape(-1)
    picked = lp.gather(1, y.unsqueeze(1)).squeeze(1)
    wy = w[y]
    return -(picked * wy * m).sum() / (wy * m).sum()


def fc_init(seed):
    g = torch.Generator().manual_seed(1000 + seed)
    return (torch.rand(6, 100, generator=g) - 0.5) * 0.2, (torch.rand(6, generator=g) - 0.5) * 0.2


def run_seed_cpu(seed, train, test):
    from oracle import stock_modules as SM
    torch.manual_seed(seed)
    gens, discs, opts = SM.build_stock()
    init = {("G", k): {a: b.clone() for a, b in m.state_dict().items()} for k, m in gens.items()}
    init.update({("D", k): {a: b.clone() for a, b in m.state_dict().items()} for k, m in discs.items()})
    vals = {}
    for it in range(1, N_GAN + 1):
        out = SM.stock_gan_iteration(gens, discs, opts, train[(it - 1) % len(train)], SCHEDULE)
        if it in GAN_AT:
            for k in LOSS_NAMES:
                vals["gan/it%d/%s" % (it, k)] = float(out[k])
    fc = torch.nn.Linear(100, 6)
    w0, b0 = fc_init(seed)
    with torch.no_grad():
        fc.weight.copy_(w0)
        fc.bias.copy_(b0)
    params = [p for m in gens.values() for p in m.parameters()] + list(fc.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=0.008)
    cw = torch.tensor(CLASS_WEIGHTS)

    def forward(b):
        fusion = gens["acoustic"](b["acoustic"]) + gens["visual"](b["visual"]) + gens["text"](b["text"])     # model.py:1441-1447
        return torch.log_softmax(fc(fusion), 2)
    for step in range(1, N_P2 + 1):
        for m in gens.values():
            m.train()
        b = train[(step - 1) % len(train)]
        opt.zero_grad()
        loss = masked_nll(forward(b), b["label"], b["umask"], cw)
        loss.backward()
        opt.step()
        if step in P2_AT:
            vals["p2/train_loss/step%d" % step] = float(loss)
    for m in gens.values():
        m.eval()
    with torch.no_grad():
        lp = forward(test)
        vals["p2/test_loss"] = float(masked_nll(lp, test["label"], test["umask"], cw))
    pred = lp.transpose(0, 1).reshape(-1, 6).argmax(1).numpy()
    acc, f1 = weighted_f1(test["label"].reshape(-1).numpy(), pred, test["umask"].reshape(-1).numpy())
    vals["p2/test_acc"], vals["p2/test_f1"] = acc, f1
    return vals, init


def run_seed_hip(seed, train, test):
    """the same protocol on the HIP engines (GanEngine, Phase2Engine); initial weights = the stock modules' under the seed"""
    from oracle import stock_modules as SM          # (the checker's modules only supply the initial state_dict)
    from gan_ffn_amd import engine as E, model as M, ops
    dev = "cuda"
    torch.manual_seed(seed)
    sg, sd, _ = SM.build_stock()
    gens = {"acoustic": M.AcousticGenerator(100), "visual": M.VisualGenerator(100), "text": M.TextGenerator(100)}
    discs = {"acoustic": M.AcousticDiscriminator(100), "visual": M.VisualDiscriminator(100), "text": M.TextDiscriminator(100)}
    for k in gens:
        gens[k].load_state_dict(sg[k].state_dict())
        discs[k].load_state_dict(sd[k].state_dict())
        gens[k], discs[k] = gens[k].to(dev), discs[k].to(dev)
    ops.manual_seed(seed, dev)
    tr = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in train]
    te = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in test.items()}
    eng = E.GanEngine(gens, discs, n_streams=3)
    vals = {}
    for it in range(1, N_GAN + 1):
        eng.iteration(tr[(it - 1) % len(tr)])
        if it in GAN_AT:
            d = eng.loss_dict()
            for k in LOSS_NAMES:
                vals["gan/it%d/%s" % (it, k)] = float(d[k])
    eng.synchronize()
    torch.cuda.synchronize()
    net = M.GAN_FFN(gens["acoustic"], gens["visual"], gens["text"], n_classes=6).to(dev)
    w0, b0 = fc_init(seed)
    with torch.no_grad():
        net.fc.weight.copy_(w0.to(dev))
        net.fc.bias.copy_(b0.to(dev))
    p2 = E.Phase2Engine(net, lr=1e-4, weight_decay=0.008)
    for step in range(1, N_P2 + 1):
        loss, _ = p2.step(tr[(step - 1) % len(tr)], train=True)
        if step in P2_AT:
            vals["p2/train_loss/step%d" % step] = float(loss)
    loss, lp = p2.step(te, train=False)
    vals["p2/test_loss"] = float(loss)
    pred = lp.transpose(0, 1).reshape(-1, 6).argmax(1).cpu().numpy()
    acc, f1 = weighted_f1(te["label"].reshape(-1).cpu().numpy(), pred, te["umask"].reshape(-1).cpu().numpy())
    vals["p2/test_acc"], vals["p2/test_f1"] = acc, f1
    return vals


def run(backend, seeds=SEEDS, log=print):
    train, test = make_data()
    names = metric_names()
    table = np.zeros((len(seeds), len(names)))
    for i, s in enumerate(seeds):
        vals = run_seed_cpu(s, train, test)[0] if backend == "cpu" else run_seed_hip(s, train, test)
        table[i] = [vals[n] for n in names]
        if log:
            log("%s seed %d: gan it%d G %.4f / %.4f / %.4f  p2 loss %.4f -> %.4f test loss %.4f acc %.1f f1 %.1f" % (
                backend, s, GAN_AT[-1], vals["gan/it%d/acoustic_G_loss" % GAN_AT[-1]], vals["gan/it%d/visual_G_loss" % GAN_AT[-1]],
                vals["gan/it%d/text_G_loss" % GAN_AT[-1]], vals["p2/train_loss/step1"], vals["p2/train_loss/step%d" % N_P2],
                vals["p2/test_loss"], vals["p2/test_acc"], vals["p2/test_f1"]))
    return names, table


if __name__ == "__main__":
    backend = sys.argv[1] if len(sys.argv) > 1 else "cpu"
    torch.set_num_threads(8)
    names, table = run(backend)
    if backend == "cpu":
        np.savez_compressed(os.path.join(HERE, "train_stats.npz"), names=np.array(names), seeds=np.array(SEEDS), cpu=table)
        print("written", os.path.join(HERE, "train_stats.npz"))
    else:
        out = os.path.join(ROOT, "gpurun_out", "train_stats_hip.npz")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        np.savez_compressed(out, names=np.array(names), seeds=np.array(SEEDS), hip=table)
        print("written", out)
