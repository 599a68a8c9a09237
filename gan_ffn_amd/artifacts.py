"""Artefact formats of the reference trainer, written/read by the build's harness:

* `GAN_loss.csv`           columns and row policy of train_GAN                 /root/reference/train_IEMOCAP.py:308-316, 384-393, 425-427
* `<dir>/<name>.pth`       whole-module checkpoints, six fixed names             train_IEMOCAP.py:430-438, 527-533
* `test_out_GAN-epochs=..` final report: loss/F1 line + sklearn report + matrix  train_IEMOCAP.py:733-754
* epoch metrics            rounding and weighting of train_or_eval_model         train_IEMOCAP.py:178-197

Host-side reporting only: metrics come from the same scikit-learn calls the reference makes.  The numbers that feed
them (losses, log-probabilities) come from the HIP path (engine.GanEngine / engine.Phase2Engine).
"""
import json
import os

import numpy as np
import torch

from .engine import LOSS_COLUMNS

GAN_LOSS_COLUMNS = ["epoch"] + LOSS_COLUMNS
MODEL_NAMES = ["acoustic_gen", "acoustic_disc", "visual_gen", "visual_disc", "text_gen", "text_disc"]   # train_IEMOCAP.py:431-438


def create_path(path):
    """make the directory of `path` (train_IEMOCAP.py:396-400)"""
    d = os.path.split(path)[0]
    if d and not os.path.exists(d):
        os.makedirs(d)


# ------------------------------------------------------------------------------------------------
# GAN_loss table
# ------------------------------------------------------------------------------------------------
def loss_table(rows=()):
    """DataFrame with the reference's columns; `rows` = the per-epoch dicts engine.train_GAN returns (last batch of
    each epoch, train_IEMOCAP.py:388-392).  Built the way train_GAN builds it (an empty frame, then one concat per
    epoch) so dtypes and CSV formatting come out the same."""
    import pandas as pd
    df = pd.DataFrame(columns=GAN_LOSS_COLUMNS)
    for r in rows:
        one = pd.DataFrame({c: r[c] for c in GAN_LOSS_COLUMNS}, index=[0])
        df = pd.concat([df, one], axis=0, ignore_index=True)
    return df


def save_GAN_loss(df, path="./output/GAN_loss.csv"):
    create_path(path)
    df.to_csv(path, index=False)


def load_GAN_loss(path="./output/GAN_loss.csv"):
    import pandas as pd
    return pd.DataFrame(pd.read_csv(path))


def extend_GAN_loss(df, more):
    """continue-training concatenation (train_IEMOCAP.py:558): NOT re-indexed and the new epochs restart at 0, as in
    the reference"""
    import pandas as pd
    return pd.concat([df, more], axis=0)


def draw_GAN_loss(df, path="./output/GAN_loss.png"):
    """six loss curves over epoch (train_IEMOCAP.py:403-422); needs matplotlib"""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.figure(figsize=(10, 8), dpi=300)
    for c in LOSS_COLUMNS:
        plt.plot(df["epoch"], df[c], label=c)
    plt.legend()
    plt.xlabel("epoch")
    plt.ylabel("loss")
    plt.title("GAN loss")
    create_path(path)
    plt.savefig(path)
    plt.close()


# ------------------------------------------------------------------------------------------------
# checkpoints: whole pickled modules, `save_path + name + ".pth"`
# ------------------------------------------------------------------------------------------------
RNG_STATE_FILE = "philox_state.json"


def save_GAN_models(models, save_path):
    """models in the order of MODEL_NAMES (a dict {"gens": .., "discs": ..} pair is accepted too)"""
    if isinstance(models, dict):
        g, d = models["gens"], models["discs"]
        models = [g["acoustic"], d["acoustic"], g["visual"], d["visual"], g["text"], d["text"]]
    for name, m in zip(MODEL_NAMES, models):
        torch.save(m, save_path + name + ".pth")
    dev = next(models[0].parameters()).device
    if dev.type == "cuda":                           # beside the reference's six files: where the Philox stream stands
        from . import ops
        with open(save_path + RNG_STATE_FILE, "w") as f:
            json.dump(ops.DeviceRng.get(dev).state_dict(), f)


def load_GAN_models(save_path, device="cuda"):
    """-> (gens, discs) dicts, modules in eval mode (train_IEMOCAP.py:527-533).  A reference checkpoint saved from an
    nn.DataParallel wrapper is unwrapped; its state_dict loads into the build's classes (same keys and shapes)."""
    from . import model as M
    cls = {"acoustic_gen": M.AcousticGenerator, "acoustic_disc": M.AcousticDiscriminator,
           "visual_gen": M.VisualGenerator, "visual_disc": M.VisualDiscriminator,
           "text_gen": M.TextGenerator, "text_disc": M.TextDiscriminator}
    out = {}
    for name in MODEL_NAMES:
        obj = torch.load(save_path + name + ".pth", map_location="cpu", weights_only=False)
        if isinstance(obj, torch.nn.DataParallel):
            obj = obj.module
        if not isinstance(obj, cls[name]):           # a foreign (stock nn.Module) pickle: take its weights
            sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
            obj = cls[name](100)
            obj.load_state_dict(sd)
        out[name] = obj.to(device).eval()
    gens = {"acoustic": out["acoustic_gen"], "visual": out["visual_gen"], "text": out["text_gen"]}
    discs = {"acoustic": out["acoustic_disc"], "visual": out["visual_disc"], "text": out["text_disc"]}
    if os.path.exists(save_path + RNG_STATE_FILE) and torch.device(device).type == "cuda":
        # a resumed run continues the dropout stream of the run it resumes (absent for reference-made checkpoints)
        from . import ops
        with open(save_path + RNG_STATE_FILE) as f:
            ops.DeviceRng.get(device).load_state_dict(json.load(f))
    return gens, discs


# ------------------------------------------------------------------------------------------------
# metrics and the final report
# ------------------------------------------------------------------------------------------------
def epoch_metrics(losses, labels, preds, masks):
    """(avg_loss, avg_accuracy, avg_fscore) with the reference's weighting and rounding (train_IEMOCAP.py:184-188):
    losses = per-batch loss * (number of real utterances of the batch)."""
    from sklearn.metrics import accuracy_score, f1_score
    avg_loss = round(np.sum(losses) / np.sum(masks), 4)
    avg_accuracy = round(accuracy_score(labels, preds, sample_weight=masks) * 100, 2)
    avg_fscore = round(f1_score(labels, preds, sample_weight=masks, average="weighted") * 100, 2)
    return avg_loss, avg_accuracy, avg_fscore


def report_text(best_loss, labels, preds, masks):
    """(text, final_f1): exactly what the reference writes to test_out_*.txt (train_IEMOCAP.py:733-752)"""
    from sklearn.metrics import classification_report, confusion_matrix, f1_score
    final_f1 = round(f1_score(labels, preds, sample_weight=masks, average="weighted") * 100, 2)
    text = "Loss {} F1-score {}".format(best_loss, final_f1)
    text += str(classification_report(labels, preds, sample_weight=masks, digits=4))
    text += str(confusion_matrix(labels, preds, sample_weight=masks))
    return text, final_f1


def write_test_report(best_loss, labels, preds, masks, g_epochs, out_dir="./output/"):
    text, f1 = report_text(best_loss, labels, preds, masks)
    file_name = os.path.join(out_dir, "test_out_GAN-epochs={}_F1-score={}.txt".format(g_epochs, f1))
    create_path(file_name)
    with open(file_name, "w") as f:
        f.write(text)
    return file_name, f1


# ------------------------------------------------------------------------------------------------
# phase-2 epoch loop (counterpart of train_or_eval_model, with the :679 argument slip fixed)
# ------------------------------------------------------------------------------------------------
def train_or_eval_model(engine, loader, train=False, device="cuda"):
    """One epoch of the classifier over `loader` (batches as data.get_IEMOCAP_loaders yields them) through
    engine.Phase2Engine.  Returns (avg_loss, avg_accuracy, labels, preds, masks, avg_fscore, [[], [], [], vids]) like
    train_IEMOCAP.py:189-197 (GAN_FFN has no attention weights: the alpha lists stay empty)."""
    from . import data as D
    losses, preds, labels, masks, vids = [], [], [], [], []
    for collated in loader:
        batch = D.to_batch(collated, device)
        loss, log_prob = engine.step(batch, train=train)
        pred = engine.predictions(log_prob)
        m = batch["umask"].reshape(-1).cpu().numpy()
        preds.append(pred.cpu().numpy())
        labels.append(batch["label"].reshape(-1).cpu().numpy())
        masks.append(m)
        losses.append(float(loss) * m.sum())
        if not train and batch.get("vids"):
            vids += batch["vids"]
    if not preds:
        return float("nan"), float("nan"), [], [], [], float("nan"), []
    preds, labels, masks = np.concatenate(preds), np.concatenate(labels), np.concatenate(masks)
    avg_loss, avg_acc, avg_f = epoch_metrics(losses, labels, preds, masks)
    return avg_loss, avg_acc, labels, preds, masks, avg_f, [[], [], [], vids]


def run_training(dataset_path, g_epochs=150, n_epochs=160, lr=1e-4, l2=0.008, batch_size=32, out_dir="./output/",
                 model_save_path="./GAN_save/", device="cuda", seed=None, log=print):
    """The reference's __main__ flow on the HIP path: GAN phase (lr 1e-4, betas (0.5, 0.6), batch 32 whatever
    `batch_size` says — train_IEMOCAP.py:595-607) -> GAN_loss.csv + six checkpoints -> GAN_FFN phase for `n_epochs`
    -> test_out_*.txt from the epoch with the best test loss.  Returns (report file, final F1, loss table)."""
    from . import data as D, engine as E, model as M
    gens, discs = E.build_networks(100, 0.2, device, seed)
    train_loader, _, _ = D.get_IEMOCAP_loaders(dataset_path, batch_size=32, valid=0.1)
    rows = E.train_GAN(gens, discs, _DeviceBatches(train_loader, device), epochs=g_epochs, lr=1e-4, b1=0.5, b2=0.6,
                       reserve_S=110)
    df = loss_table(rows)
    save_GAN_loss(df, os.path.join(out_dir, "GAN_loss.csv"))
    if not os.path.exists(model_save_path):
        os.makedirs(model_save_path)
    save_GAN_models({"gens": gens, "discs": discs}, model_save_path)
    for m in list(gens.values()) + list(discs.values()):
        m.eval()
    net = M.GAN_FFN(gens["acoustic"], gens["visual"], gens["text"], n_classes=6).to(device)
    eng = E.Phase2Engine(net, lr=lr, weight_decay=l2)
    eng.reserve(110, batch_size)                     # PositionalEncoding caps a dialogue at 110 utterances
    train_loader, valid_loader, test_loader = D.get_IEMOCAP_loaders(dataset_path, batch_size=batch_size, valid=0.1)
    best = None
    for e in range(n_epochs):
        tr = train_or_eval_model(eng, train_loader, True, device)
        va = train_or_eval_model(eng, valid_loader, False, device)
        te = train_or_eval_model(eng, test_loader, False, device)
        if best is None or best[0] > te[0]:
            best = te
        if log:
            log("epoch {} train_loss {} train_acc {} train_fscore {} valid_loss {} valid_acc {} val_fscore {} "
                "test_loss {} test_acc {} test_fscore {}".format(e + 1, tr[0], tr[1], tr[5], va[0], va[1], va[5],
                                                                 te[0], te[1], te[5]))
    file_name, f1 = write_test_report(best[0], best[2], best[3], best[4], g_epochs, out_dir)
    return file_name, f1, df


class _DeviceBatches:
    """re-iterable view of a DataLoader that yields engine batches already on the device"""

    def __init__(self, loader, device):
        self.loader, self.device = loader, device

    def __iter__(self):
        from . import data as D
        for collated in self.loader:
            yield D.to_batch(collated, self.device)
