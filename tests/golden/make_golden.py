"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU, this container only).

    python tests/golden/make_golden.py            # needs /root/reference

Imports /root/reference/model.py and train_IEMOCAP.py unmodified, loads the
formula weights of tests/golden/formula.py into the reference's own modules and
records inputs-independent expected outputs.  The fixtures are data (inputs are
regenerated from formulas; expected outputs/gradients/losses are stored).  The
reference never travels: only these .npz files and this script are committed.

Dropout: eval mode, or train mode with every dropout probability set to 0 on the
reference module instances (attribute assignment on objects; no reference code is
changed), because torch's CPU dropout stream cannot be reproduced elsewhere.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn as nn

import formula as F_  # noqa: E402
import model as ref  # noqa: E402  (the reference)

torch.set_num_threads(8)

GEN = {"acoustic": "AcousticGenerator", "visual": "VisualGenerator", "text": "TextGenerator"}
DISC = {"acoustic": "AcousticDiscriminator", "visual": "VisualDiscriminator", "text": "TextDiscriminator"}
DIN = {"acoustic": 100, "visual": 512, "text": 100}

SELECTED = [
    "transformer_encoder.layers.0.self_attn.in_proj_weight",
    "transformer_encoder.layers.0.self_attn.in_proj_bias",
    "transformer_encoder.layers.3.self_attn.out_proj.weight",
    "transformer_encoder.layers.3.self_attn.out_proj.bias",
    "transformer_encoder.layers.7.linear1.weight",
    "transformer_encoder.layers.7.linear1.bias",
    "transformer_encoder.layers.4.linear2.weight",
    "transformer_encoder.layers.4.linear2.bias",
    "transformer_encoder.layers.2.norm1.weight",
    "transformer_encoder.layers.2.norm1.bias",
    "transformer_encoder.layers.5.norm2.weight",
    "transformer_encoder.layers.5.norm2.bias",
    "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias",
    "object.weight", "object.bias",
]


def build(cls_name):
    m = getattr(ref, cls_name)(100, dropout=0.2)
    sd = F_.formula_state_dict(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m


def zero_dropout(m):
    for sub in m.modules():
        if isinstance(sub, nn.Dropout):
            sub.p = 0.0
        if isinstance(sub, nn.MultiheadAttention):
            sub.dropout = 0.0


def put(d, prefix, t):
    for k, v in F_.summarize(t.detach().cpu().numpy() if torch.is_tensor(t) else t).items():
        d[prefix + "/" + k] = v


def module_cases():
    out = {}
    for cls_name, din in (("AcousticGenerator", 100), ("TextGenerator", 100), ("VisualGenerator", 512),
                          ("AcousticDiscriminator", 100), ("TextDiscriminator", 100),
                          ("VisualDiscriminator", 512), ("VisualDiscriminator", 100)):
        m = build(cls_name).eval()
        for (S, B) in ((7, 2), (110, 3)):
            tag = "%s.%d.%dx%d" % (cls_name, din, S, B)
            x = torch.from_numpy(F_.formula_input(tag, S, B, din, pad_from=max(1, S - 3))).requires_grad_(True)
            for p in m.parameters():
                p.grad = None
            y = m(x)
            g = torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1])) - 0.5
            (y * g).sum().backward()
            put(out, tag + "/out", y)
            put(out, tag + "/dx", x.grad)
            sd = dict(m.named_parameters())
            for k in SELECTED:
                if k in sd and sd[k].grad is not None:
                    put(out, tag + "/grad/" + k, sd[k].grad)
            # the template layer never receives a gradient (model.py:1210-1213)
            out[tag + "/template_grad_is_none"] = np.array(
                all(p.grad is None for k, p in sd.items() if k.startswith("encoder_layer.")))
            print("module", tag, tuple(y.shape), float(y.abs().mean()))
    return out


def gan_steps():
    """Two full 12-sub-step iterations via the reference's own train_disc/train_gen, dropout p=0."""
    import train_IEMOCAP as T
    out = {}
    S, B = 7, 2
    gens = {k: build(v) for k, v in GEN.items()}
    discs = {k: build(v) for k, v in DISC.items()}
    for m in list(gens.values()) + list(discs.values()):
        zero_dropout(m)
    lr, b1, b2 = 1e-4, 0.5, 0.6  # the actual call, train_IEMOCAP.py:603-606
    A = torch.optim.Adam
    opt = {("G", "acoustic"): A(gens["acoustic"].parameters(), lr=lr, betas=(b1, b2)),
           ("D", "acoustic"): A(discs["acoustic"].parameters(), lr=lr / 2, betas=(b1, b2)),
           ("G", "visual"): A(gens["visual"].parameters(), lr=lr, betas=(b1, b2)),
           ("D", "visual"): A(discs["visual"].parameters(), lr=lr / 2, betas=(b1, b2)),
           ("G", "text"): A(gens["text"].parameters(), lr=lr * 1.1, betas=(b1, b2)),
           ("D", "text"): A(discs["text"].parameters(), lr=lr / 2, betas=(b1, b2))}
    bce = nn.BCELoss()
    batch = {k: torch.from_numpy(F_.formula_input("gan." + k, S, B, DIN[k], pad_from=5)) for k in DIN}
    valid = torch.ones(S, B, 1)
    fake = torch.zeros(S, B, 1)
    sched = [("D", "visual", "acoustic"), ("G", "acoustic", "visual"), ("D", "visual", "text"),
             ("G", "text", "visual"), ("D", "text", "acoustic"), ("G", "acoustic", "text"),
             ("D", "acoustic", "text"), ("G", "text", "acoustic"), ("D", "text", "visual"),
             ("G", "visual", "text"), ("D", "acoustic", "visual"), ("G", "visual", "acoustic")]
    losses = []
    seen = set()
    for it in range(2):
        for kind, who, partner in sched:  # order of train_IEMOCAP.py:355-382
            if kind == "D":
                v = T.train_disc(discs[who], batch[who], gens[partner], batch[partner], opt[("D", who)], bce, valid, fake)
            else:
                v = T.train_gen(gens[who], batch[who], discs[partner], opt[("G", who)], bce, valid, fake)
            losses.append(float(v))
            print("gan it%d %s %s|%s loss %.7f" % (it, kind, who, partner, float(v)))
            if (kind, who) not in seen:
                # parameter delta right after this module's FIRST Adam step (t = 1).  Later states are
                # not fixtures: Adam's first steps are sign-like (delta = -lr*g/(|g|+eps)), so fp32
                # rounding noise on ~0 gradients flips +-lr updates and trajectories separate
                # chaotically (an fp64 restatement drifts from this fp32 run just as much).
                seen.add((kind, who))
                sd = dict((discs if kind == "D" else gens)[who].named_parameters())
                for k in SELECTED:
                    if k in sd:
                        w0 = F_.formula_tensor(k, tuple(sd[k].shape))
                        put(out, "gan/%s_%s/delta1/%s" % (kind, who, k), sd[k].detach().numpy() - w0)
    out["gan/losses"] = np.array(losses, dtype=np.float64)
    return out


def misc():
    out = {}
    out["pe/100"] = ref.PositionalEncoding(100).pe.numpy()[:, 0, :]
    out["pe/512"] = ref.PositionalEncoding(512).pe.numpy()[:, 0, :]
    # BCE edge cases (call site train_IEMOCAP.py:300)
    p = torch.tensor([0.0, 1.0, 1e-45, 1.0 - 1e-7, 0.5, 0.25, 1e-30, 0.9999999], dtype=torch.float32)
    for tgt in (0.0, 1.0):
        y = torch.full_like(p, tgt)
        out["bce/target%d" % int(tgt)] = np.float64(nn.BCELoss()(p, y).item())
        out["bce/target%d_elem" % int(tgt)] = nn.BCELoss(reduction="none")(p, y).numpy()
    out["bce/probs"] = p.numpy()
    # Adam (call sites train_IEMOCAP.py:292-297, :661)
    for tag, kw in (("gan", dict(lr=1e-4, betas=(0.5, 0.6))), ("phase2", dict(lr=1e-4, weight_decay=0.008))):
        w = torch.from_numpy(F_.formula_tensor("adam.w", (37, 11))).clone().requires_grad_(True)
        o = torch.optim.Adam([w], **kw)
        for step in range(3):
            w.grad = torch.from_numpy(F_.formula_tensor("adam.g%d" % step, (37, 11))).clone()
            o.step()
            out["adam/%s/step%d" % (tag, step)] = w.detach().numpy().copy()
    # phase 2: GAN_FFN forward + MaskedNLLLoss (model.py:1434-1462, :62-81; train_IEMOCAP.py:151-156,653)
    S, B = 7, 2
    gens = {k: build(v).eval() for k, v in GEN.items()}
    net = ref.GAN_FFN(gens["acoustic"], gens["visual"], gens["text"], n_classes=6, dropout=0.2).eval()
    with torch.no_grad():
        net.fc.weight.copy_(torch.from_numpy(F_.formula_tensor("phase2.fc.weight", (6, 100))))
        net.fc.bias.copy_(torch.from_numpy(F_.formula_tensor("phase2.fc.bias", (6,))))
    batch = {k: torch.from_numpy(F_.formula_input("gan." + k, S, B, DIN[k], pad_from=5)) for k in DIN}
    lp, _, _, _ = net(batch["acoustic"], batch["visual"], batch["text"])
    umask = torch.tensor([[1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 0, 0]], dtype=torch.float32)
    label = torch.tensor([[0, 1, 2, 3, 4, 5, 0], [5, 4, 3, 2, 1, 0, 0]], dtype=torch.long)
    wts = torch.FloatTensor([1.2, 0.60072, 0.38066, 0.94019, 0.67924, 0.34332])
    lp_ = lp.transpose(0, 1).contiguous().view(-1, lp.size()[2])
    out["phase2/log_prob"] = lp.detach().numpy()
    loss_w = ref.MaskedNLLLoss(wts)(lp_, label.view(-1), umask)
    out["phase2/loss_weighted"] = np.float64(loss_w.item())
    out["phase2/loss_unweighted"] = np.float64(ref.MaskedNLLLoss()(lp_, label.view(-1), umask).item())
    loss_w.backward()
    out["phase2/grad_fc_weight"] = net.fc.weight.grad.numpy()
    put(out, "phase2/grad_text_fc2_weight", gens["text"].fc2.weight.grad)
    put(out, "phase2/grad_visual_l0_inproj", gens["visual"].transformer_encoder.layers[0].self_attn.in_proj_weight.grad)
    out["phase2/umask"] = umask.numpy()
    out["phase2/label"] = label.numpy()
    return out


def artifacts():
    """N3 fixtures: the reference's dataset / collate on a seeded synthetic pickle, its GAN_loss.csv writer, and the
    report text its __main__ assembles (train_IEMOCAP.py:733-752, restated here with the same sklearn calls)."""
    import tempfile
    import pandas as pd
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from gan_ffn_amd.data import write_synthetic_iemocap_pickle     # writes the INPUT pickle only
    import dataloader as ref_dl                                     # the reference
    import train_IEMOCAP as ref_tr
    out = {}
    tmp = tempfile.mkdtemp()
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        pk = os.path.join(tmp, "syn_%s.pkl" % tag)
        write_synthetic_iemocap_pickle(pk, n_train=12, n_test=5, seed=3407, dtype=dt)
        for split, train in (("train", True), ("test", False)):
            ds = ref_dl.IEMOCAPDataset(pk, train=train)
            out["ds/%s/%s/len" % (tag, split)] = np.array(len(ds))
            batch = ds.collate_fn([ds[i] for i in range(3)])
            for name, t in zip(("text", "visual", "audio", "qmask", "umask", "label"), batch[:6]):
                out["ds/%s/%s/%s" % (tag, split, name)] = t.numpy()
            out["ds/%s/%s/vids" % (tag, split)] = np.array(batch[6])
    # sampler split (indices only; the order inside is random by design)
    ds = ref_dl.IEMOCAPDataset(os.path.join(tmp, "syn_f64.pkl"), train=True)
    tr, va = ref_tr.get_train_valid_sampler(ds, 0.2)
    out["sampler/train"], out["sampler/valid"] = np.array(sorted(tr.indices)), np.array(sorted(va.indices))
    # GAN_loss.csv: the frame built the way train_GAN builds it, written by the reference's save_GAN_loss
    cols = ["epoch", "acoustic_G_loss", "visual_G_loss", "text_G_loss", "visual_D_loss", "text_D_loss", "acoustic_D_loss"]
    df = pd.DataFrame(columns=cols)
    vals = np.array(F_.formula_tensor("artifacts.losses", (3, 6)), dtype=np.float32) + 0.7
    for e in range(3):
        loss = {"epoch": e}
        for j, c in enumerate(cols[1:]):
            loss[c] = np.asarray(vals[e, j])          # train_disc returns a 0-d float32 ndarray
        df = pd.concat([df, pd.DataFrame(loss, index=[0])], axis=0, ignore_index=True)
    path = os.path.join(tmp, "out", "GAN_loss.csv")
    ref_tr.save_GAN_loss(df, path)
    out["csv/values"] = vals
    out["csv/text"] = np.array(open(path).read())
    more = pd.concat([pd.DataFrame(pd.read_csv(path)), df.iloc[:1]], axis=0)      # continue-training concat, :558
    ref_tr.save_GAN_loss(more, path)
    out["csv/text_continued"] = np.array(open(path).read())
    # report text
    from sklearn.metrics import f1_score, classification_report, confusion_matrix, accuracy_score
    rng = np.random.default_rng(11)
    labels = rng.integers(0, 6, 200)
    preds = np.where(rng.random(200) < 0.6, labels, rng.integers(0, 6, 200))
    masks = (rng.random(200) < 0.85).astype(np.float32)
    best_loss = 1.2345
    f1 = round(f1_score(labels, preds, sample_weight=masks, average="weighted") * 100, 2)
    text = "Loss {} F1-score {}".format(best_loss, f1)
    text += str(classification_report(labels, preds, sample_weight=masks, digits=4))
    text += str(confusion_matrix(labels, preds, sample_weight=masks))
    out["report/labels"], out["report/preds"], out["report/masks"] = labels, preds, masks
    out["report/best_loss"], out["report/f1"], out["report/text"] = np.array(best_loss), np.array(f1), np.array(text)
    out["report/acc"] = np.array(round(accuracy_score(labels, preds, sample_weight=masks) * 100, 2))
    return out


DRNN_CASES = {"general": dict(context_attention="general", listener_state=False),
              "simple_listener": dict(context_attention="simple", listener_state=True),
              "simple": dict(context_attention="simple", listener_state=False)}       # (round 5: DialogueRNNCell's default attention)
DRNN_DIMS = dict(D_m=100, D_g=500, D_p=500, D_e=100, D_h=100, n_classes=6, D_a=100, dropout_rec=0.1, dropout=0.6)
DRNN_LENS = [7, 4, 6]


def drnn_inputs():
    S, B = max(DRNN_LENS), len(DRNN_LENS)
    U = F_.formula_input("drnn.U", S, B, 100)
    umask = np.zeros((B, S), np.float32)
    for b, L in enumerate(DRNN_LENS):
        umask[b, :L] = 1
        U[L:, b] = 0
    spk = (np.arange(S)[:, None] * 3 + np.arange(B)[None, :] * 2 + (np.arange(S)[:, None] // 3)) % 2
    qmask = np.stack([1 - spk, spk], -1).astype(np.float32) * umask.T[:, :, None]
    return U, qmask, umask


def dialogue_rnn():
    """N2 fixtures: the reference's BiModel (eval mode: torch's CPU dropout stream is not reproducible), formula
    weights, ragged dialogues: log-probabilities, the three attention maps, input gradient, parameter gradients."""
    out = {}
    U, qmask, umask = drnn_inputs()
    for tag, kw in DRNN_CASES.items():
        torch.manual_seed(0)
        m = ref.BiModel(**DRNN_DIMS, **kw).eval()
        sd = F_.formula_state_dict({k: v for k, v in m.state_dict().items()})
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        Ut = torch.from_numpy(U).requires_grad_(True)
        lp, alpha, alpha_f, alpha_b = m(Ut, torch.from_numpy(qmask), torch.from_numpy(umask))
        gy = torch.from_numpy(F_.formula_input("drnn.grad", lp.shape[0], lp.shape[1], lp.shape[2])) - 0.5
        (lp * gy).sum().backward()
        out["%s/log_prob" % tag] = lp.detach().numpy()
        out["%s/alpha" % tag] = torch.stack(alpha, 0).detach().numpy()                     # (S, B, S)
        for name, al in (("alpha_f", alpha_f), ("alpha_b", alpha_b)):
            for t, a in enumerate(al):
                out["%s/%s/%d" % (tag, name, t)] = a.detach().numpy()
            out["%s/%s/n" % (tag, name)] = np.array(len(al))
        out["%s/dU" % tag] = Ut.grad.numpy()
        for k, p_ in m.named_parameters():
            if p_.grad is not None:
                out["%s/grad/%s" % (tag, k)] = p_.grad.numpy() if p_.grad.numel() <= 4096 else \
                    p_.grad.reshape(-1)[F_.sample_indices(p_.grad.numel())].numpy()
    # N4: the reference's MELDLSTMModel (eval), real dims of train_MELD.py:143-151
    torch.manual_seed(0)
    mm = ref.MELDLSTMModel(600, 300, 600, n_classes=7, dropout=0.5).eval()
    sd = F_.formula_state_dict(mm.state_dict())
    mm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    Um = torch.from_numpy(F_.formula_input("meld.U", 7, 3, 600)).requires_grad_(True)
    lp, alpha, _, _ = mm(Um, None, torch.from_numpy(umask))
    gy = torch.from_numpy(F_.formula_input("meld.grad", 7, 3, 7)) - 0.5
    (lp * gy).sum().backward()
    out["meld/log_prob"], out["meld/alpha"], out["meld/dU"] = lp.detach().numpy(), torch.stack(alpha, 0).detach().numpy(), Um.grad.numpy()
    for k in ("lstm.weight_ih_l0", "lstm.weight_hh_l3_reverse", "lstm.bias_ih_l2", "matchatt.transform.weight", "smax_fc.weight"):
        gk = dict(mm.named_parameters())[k].grad
        out["meld/grad/" + k] = gk.numpy() if gk.numel() <= 4096 else gk.reshape(-1)[F_.sample_indices(gk.numel())].numpy()
    # MatchingAttention general2 alone, with a mask (the A12 parity target, model.py:169-182)
    att = ref.MatchingAttention(200, 200, att_type="general2").eval()
    sd = F_.formula_state_dict(att.state_dict())
    att.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    M = torch.from_numpy(F_.formula_input("drnn.M", 7, 3, 200)) - 0.5
    pool, al = att(M, M[2], mask=torch.from_numpy(umask))
    out["general2/pool"], out["general2/alpha"] = pool.detach().numpy(), al.detach().numpy()
    return out


BIG_S, BIG_B = 94, 30        # configuration 5's real size: train_IEMOCAP_DialogueRNN.py:580 (batch 30), model.py:1437 (S = 94)


def drnn_big_inputs():
    """ragged (94, 30) batch, closed form: dialogue 0 is full length, the others 12 .. 93 utterances"""
    S, B = BIG_S, BIG_B
    lens = [S] + [12 + (b * 37) % 82 for b in range(1, B)]
    U = F_.formula_input("drnn.bigU", S, B, 100)
    umask = np.zeros((B, S), np.float32)
    for b, L in enumerate(lens):
        umask[b, :L] = 1
        U[L:, b] = 0
    spk = (np.arange(S)[:, None] * 3 + np.arange(B)[None, :] * 2 + (np.arange(S)[:, None] // 3)) % 2
    qmask = np.stack([1 - spk, spk], -1).astype(np.float32) * umask.T[:, :, None]
    return U, qmask, umask


def dialogue_rnn_big():
    """the reference's BiModel (general attention, no listener: the trained configuration,
    train_IEMOCAP_DialogueRNN.py:586,595) at configuration 5's real size, eval mode, formula weights: summaries of the
    log-probabilities, the matching-attention map, the input gradient and sampled parameter gradients"""
    out = {}
    U, qmask, umask = drnn_big_inputs()
    torch.manual_seed(0)
    m = ref.BiModel(**DRNN_DIMS, **DRNN_CASES["general"]).eval()
    sd = F_.formula_state_dict({k: v for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    Ut = torch.from_numpy(U).requires_grad_(True)
    lp, alpha, alpha_f, alpha_b = m(Ut, torch.from_numpy(qmask), torch.from_numpy(umask))
    gy = torch.from_numpy(F_.formula_input("drnn.biggrad", lp.shape[0], lp.shape[1], lp.shape[2])) - 0.5
    (lp * gy).sum().backward()
    put(out, "big/log_prob", lp)
    put(out, "big/alpha", torch.stack(alpha, 0))
    put(out, "big/alpha_f_last", alpha_f[-1])
    put(out, "big/alpha_b_last", alpha_b[-1])
    put(out, "big/dU", Ut.grad)
    for k, p_ in m.named_parameters():
        if p_.grad is not None:
            put(out, "big/grad/" + k, p_.grad)
    return out


HEAD_S, HEAD_B = 94, 32      # the headline configuration: BASELINE.json configs[1] (batch 32, train_IEMOCAP.py:603; S = 94, model.py:1437)
BIG_CASES = (("AcousticGenerator", 100), ("TextGenerator", 100), ("VisualGenerator", 512), ("AcousticDiscriminator", 100),
             ("TextDiscriminator", 100), ("VisualDiscriminator", 512), ("VisualDiscriminator", 100))


def modules_big():
    """the six reference modules (seven input cases) at the HEADLINE size (94, 32), eval mode, formula weights: summaries
    (strided samples + sum + l2) of the output, the input gradient and the SELECTED parameter gradients — the same
    quantities as module_cases(), so the GPU tests compare the HIP path with the reference directly at this size"""
    out = {}
    S, B = HEAD_S, HEAD_B
    for cls_name, din in BIG_CASES:
        m = build(cls_name).eval()
        tag = "%s.%d.%dx%d" % (cls_name, din, S, B)
        x = torch.from_numpy(F_.formula_input(tag, S, B, din, pad_from=61)).requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(x)
        g = torch.from_numpy(F_.formula_input("grad." + tag, S, B, y.shape[-1])) - 0.5
        (y * g).sum().backward()
        put(out, tag + "/out", y)
        put(out, tag + "/dx", x.grad)
        sd = dict(m.named_parameters())
        for k in SELECTED:
            if k in sd and sd[k].grad is not None:
                put(out, tag + "/grad/" + k, sd[k].grad)
        print("module_big", tag, tuple(y.shape), float(y.abs().mean()))
    return out


def gan_steps_big():
    """ONE full 12-sub-step iteration at the headline size (94, 32) through the reference's own train_disc / train_gen
    (dropout p -> 0 on the instances): the 12 losses and every network's first-update parameter deltas"""
    import train_IEMOCAP as T
    out = {}
    S, B = HEAD_S, HEAD_B
    gens = {k: build(v) for k, v in GEN.items()}
    discs = {k: build(v) for k, v in DISC.items()}
    for m in list(gens.values()) + list(discs.values()):
        zero_dropout(m)
    lr, b1, b2 = 1e-4, 0.5, 0.6
    A = torch.optim.Adam
    opt = {("G", "acoustic"): A(gens["acoustic"].parameters(), lr=lr, betas=(b1, b2)),
           ("D", "acoustic"): A(discs["acoustic"].parameters(), lr=lr / 2, betas=(b1, b2)),
           ("G", "visual"): A(gens["visual"].parameters(), lr=lr, betas=(b1, b2)),
           ("D", "visual"): A(discs["visual"].parameters(), lr=lr / 2, betas=(b1, b2)),
           ("G", "text"): A(gens["text"].parameters(), lr=lr * 1.1, betas=(b1, b2)),
           ("D", "text"): A(discs["text"].parameters(), lr=lr / 2, betas=(b1, b2))}
    bce = nn.BCELoss()
    batch = {k: torch.from_numpy(F_.formula_input("ganbig." + k, S, B, DIN[k], pad_from=61)) for k in DIN}
    valid, fake = torch.ones(S, B, 1), torch.zeros(S, B, 1)
    sched = [("D", "visual", "acoustic"), ("G", "acoustic", "visual"), ("D", "visual", "text"),
             ("G", "text", "visual"), ("D", "text", "acoustic"), ("G", "acoustic", "text"),
             ("D", "acoustic", "text"), ("G", "text", "acoustic"), ("D", "text", "visual"),
             ("G", "visual", "text"), ("D", "acoustic", "visual"), ("G", "visual", "acoustic")]
    losses, seen = [], set()
    for kind, who, partner in sched:
        if kind == "D":
            v = T.train_disc(discs[who], batch[who], gens[partner], batch[partner], opt[("D", who)], bce, valid, fake)
        else:
            v = T.train_gen(gens[who], batch[who], discs[partner], opt[("G", who)], bce, valid, fake)
        losses.append(float(v))
        print("gan_big %s %s|%s loss %.7f" % (kind, who, partner, float(v)), flush=True)
        if (kind, who) not in seen:
            seen.add((kind, who))
            sd = dict((discs if kind == "D" else gens)[who].named_parameters())
            for k in SELECTED:
                if k in sd:
                    w0 = F_.formula_tensor(k, tuple(sd[k].shape))
                    put(out, "gan/%s_%s/delta1/%s" % (kind, who, k), sd[k].detach().numpy() - w0)
    out["gan/losses"] = np.array(losses, dtype=np.float64)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dialogue_rnn":
        np.savez_compressed(os.path.join(HERE, "dialogue_rnn.npz"), **dialogue_rnn())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dialogue_rnn_big":
        np.savez_compressed(os.path.join(HERE, "dialogue_rnn_big.npz"), **dialogue_rnn_big())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "modules_big":
        np.savez_compressed(os.path.join(HERE, "modules_big.npz"), **modules_big())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gan_steps_big":
        np.savez_compressed(os.path.join(HERE, "gan_steps_big.npz"), **gan_steps_big())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "artifacts":
        np.savez_compressed(os.path.join(HERE, "artifacts.npz"), **artifacts())
        sys.exit(0)
    torch.manual_seed(0)
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **misc())
    np.savez_compressed(os.path.join(HERE, "modules.npz"), **module_cases())
    np.savez_compressed(os.path.join(HERE, "gan_steps.npz"), **gan_steps())
    np.savez_compressed(os.path.join(HERE, "artifacts.npz"), **artifacts())
    np.savez_compressed(os.path.join(HERE, "dialogue_rnn.npz"), **dialogue_rnn())
    np.savez_compressed(os.path.join(HERE, "dialogue_rnn_big.npz"), **dialogue_rnn_big())
    np.savez_compressed(os.path.join(HERE, "modules_big.npz"), **modules_big())
    np.savez_compressed(os.path.join(HERE, "gan_steps_big.npz"), **gan_steps_big())
    print("golden fixtures written to", HERE)
