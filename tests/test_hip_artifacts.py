"""N3 on the GPU: the reference's __main__ flow (GAN phase -> GAN_loss.csv + six checkpoints -> classifier phase ->
test_out_*.txt) driven through the HIP engines on a small synthetic IEMOCAP-schema pickle."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_run_training_writes_reference_artefacts(tmp_path):
    from gan_ffn_amd import artifacts as A, data as D
    pk = str(tmp_path / "syn.pkl")
    D.write_synthetic_iemocap_pickle(pk, n_train=40, n_test=9, seed=7, dtype=np.float32)
    out_dir, save = str(tmp_path / "output") + "/", str(tmp_path / "GAN_save") + "/"
    logs = []
    file_name, f1, df = A.run_training(pk, g_epochs=2, n_epochs=2, out_dir=out_dir, model_save_path=save, seed=3407,
                                       log=logs.append)
    # GAN_loss.csv: header and one row per GAN epoch (last batch of the epoch), losses finite and BCE-like
    lines = open(out_dir + "GAN_loss.csv").read().strip().split("\n")
    assert lines[0] == ",".join(A.GAN_LOSS_COLUMNS) and len(lines) == 3
    vals = np.array([[float(x) for x in ln.split(",")] for ln in lines[1:]])
    assert list(vals[:, 0]) == [0, 1] and np.isfinite(vals).all() and (vals[:, 1:] > 0.05).all() and (vals[:, 1:] < 5).all()
    # checkpoints: six whole-module pickles that load back and run
    assert sorted(os.path.basename(p) for p in glob.glob(save + "*.pth")) == sorted(n + ".pth" for n in A.MODEL_NAMES)
    gens, discs = A.load_GAN_models(save)
    x = torch.rand(9, 2, 100, device="cuda")
    p = discs["acoustic"](gens["text"](x))
    assert p.shape == (9, 2, 1) and bool(((p > 0) & (p < 1)).all())
    # report: name carries the GAN epochs and the F1; content starts with the loss/F1 line
    assert os.path.basename(file_name) == "test_out_GAN-epochs=2_F1-score=%s.txt" % f1
    txt = open(file_name).read()
    assert txt.startswith("Loss ") and "F1-score %s" % f1 in txt and "weighted avg" in txt
    assert len(logs) == 2 and logs[0].startswith("epoch 1 train_loss ")


def test_phase2_epoch_metrics_consistent_with_manual_count(tmp_path):
    """train_or_eval_model (eval): loss = sum(batch loss * real utterances) / real utterances; accuracy from argmax"""
    from gan_ffn_amd import artifacts as A, data as D, engine as E, model as M
    pk = str(tmp_path / "syn.pkl")
    D.write_synthetic_iemocap_pickle(pk, n_train=10, n_test=6, seed=9, dtype=np.float32)
    gens, _ = E.build_networks(100, 0.2, "cuda", seed=1)
    net = M.GAN_FFN(gens["acoustic"], gens["visual"], gens["text"], n_classes=6).cuda()
    eng = E.Phase2Engine(net)
    _, _, test_loader = D.get_IEMOCAP_loaders(pk, batch_size=4, valid=0.1)
    avg_loss, acc, labels, preds, masks, f, extra = A.train_or_eval_model(eng, test_loader, train=False)
    assert labels.shape == preds.shape == masks.shape and masks.sum() > 0
    assert abs(acc - round(100.0 * ((labels == preds) * masks).sum() / masks.sum(), 2)) < 1e-9
    assert np.isfinite(avg_loss) and extra[3] == D.IEMOCAPDataset(pk, train=False).testVid


def test_checkpoints_carry_the_philox_stream_and_engines_refuse_a_moved_slab(tmp_path):
    """ the dropout-offset allocator is one per device and travels with the checkpoints; a module that
    re-packs into a new slab after an engine captured it is refused instead of silently ignored; a no-op .to() keeps
    the slab in place"""
    from gan_ffn_amd import artifacts as A, engine as E, ops, data as D
    gens, discs = E.build_networks(100, 0.2, "cuda", 5)
    ops.manual_seed(11)
    eng = E.GanEngine(gens, discs)
    b = D.synthetic_batch(B=3, S_max=9, seed=2, device="cuda")
    eng.iteration(b)
    eng.synchronize()
    rng = ops.DeviceRng.get("cuda")
    assert rng.counter == 48
    y = gens["text"].train()(b["text"])                      # the module path draws from the same allocator
    assert rng.counter == 50 and torch.isfinite(y).all()      # encoder + head: two dropout-bearing calls
    save = str(tmp_path) + "/GAN_save_"
    A.save_GAN_models({"gens": gens, "discs": discs}, save)
    ops.manual_seed(3)                                        # a new process would start from its own default
    A.load_GAN_models(save, "cuda")
    assert rng.state_dict() == {"seed": 11, "offset": 0, "counter": 50}
    # no-op .to(): same slab, engine still attached
    ptr = gens["text"].slab.data_ptr()
    gens["text"].to("cuda")
    assert gens["text"].slab.data_ptr() == ptr
    eng.iteration(b)
    eng.synchronize()
    # a real re-pack (round trip through the host) moves the slab: the engine must notice
    gens["acoustic"].cpu()
    gens["acoustic"].cuda()
    assert gens["acoustic"].slab.data_ptr() != eng.G["acoustic"].slab.data_ptr()
    with pytest.raises(RuntimeError, match="re-packed"):
        eng.iteration(b)
