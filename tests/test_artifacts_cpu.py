"""N3: data and artefact formats against fixtures produced by the reference itself
(tests/golden/make_golden.py artifacts): dataset normalisation + collate, sampler split, GAN_loss.csv bytes,
the final report text, checkpoint naming / loading.  CPU only — no kernels involved."""
import os

import numpy as np
import pytest
import torch

from util import golden


@pytest.fixture(scope="module")
def g():
    return golden("artifacts")


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_iemocap_dataset_and_collate_match_reference(g, tmp_path, tag, dtype):
    from gan_ffn_amd import data as D
    pk = str(tmp_path / "syn.pkl")
    D.write_synthetic_iemocap_pickle(pk, n_train=12, n_test=5, seed=3407, dtype=dtype)
    for split, train in (("train", True), ("test", False)):
        ds = D.IEMOCAPDataset(pk, train=train)
        assert len(ds) == int(g["ds/%s/%s/len" % (tag, split)])
        batch = ds.collate_fn([ds[i] for i in range(3)])
        for name, t in zip(("text", "visual", "audio", "qmask", "umask", "label"), batch[:6]):
            ref = g["ds/%s/%s/%s" % (tag, split, name)]
            assert t.numpy().dtype == ref.dtype and t.shape == ref.shape, name
            assert np.array_equal(t.numpy(), ref), "%s/%s/%s differs from the reference's collate" % (tag, split, name)
        assert list(batch[6]) == list(g["ds/%s/%s/vids" % (tag, split)])
    # the dict the engines take
    b = D.to_batch(batch, "cpu")
    assert b["text"].shape[0] == b["umask"].shape[1] and b["label"].dtype == torch.int64 and b["umask"].dtype == torch.float32


def test_sampler_split_and_loaders(g, tmp_path):
    from gan_ffn_amd import data as D
    pk = str(tmp_path / "syn.pkl")
    D.write_synthetic_iemocap_pickle(pk, n_train=12, n_test=5, seed=3407)
    ds = D.IEMOCAPDataset(pk, train=True)
    tr, va = D.get_train_valid_sampler(ds, 0.2)
    assert sorted(tr.indices) == list(g["sampler/train"]) and sorted(va.indices) == list(g["sampler/valid"])
    train_loader, valid_loader, test_loader = D.get_IEMOCAP_loaders(pk, batch_size=4, valid=0.2)
    assert sum(len(b[6]) for b in train_loader) == len(tr.indices)
    assert sum(len(b[6]) for b in valid_loader) == len(va.indices)
    assert [v for b in test_loader for v in b[6]] == ds.testVid           # test order is sequential


def test_gan_loss_csv_bytes_match_reference(g, tmp_path):
    from gan_ffn_amd import artifacts as A
    vals = g["csv/values"]
    rows = [dict(epoch=e, **{c: np.asarray(vals[e, j]) for j, c in enumerate(A.GAN_LOSS_COLUMNS[1:])}) for e in range(3)]
    df = A.loss_table(rows)
    assert list(df.columns) == A.GAN_LOSS_COLUMNS
    path = str(tmp_path / "out" / "GAN_loss.csv")           # directory is created on demand, as the reference does
    A.save_GAN_loss(df, path)
    assert open(path).read() == str(g["csv/text"])
    # continue-training: read back, append, rewrite (train_IEMOCAP.py:540-560)
    more = A.extend_GAN_loss(A.load_GAN_loss(path), df.iloc[:1])
    A.save_GAN_loss(more, path)
    assert open(path).read() == str(g["csv/text_continued"])


def test_loss_table_accepts_engine_rows():
    """engine.train_GAN rows carry python floats; the table keeps the reference's column order"""
    from gan_ffn_amd import artifacts as A
    rows = [dict(epoch=0, acoustic_G_loss=0.7, visual_G_loss=0.6, text_G_loss=0.8, visual_D_loss=0.5, text_D_loss=0.69,
                 acoustic_D_loss=0.71)]
    df = A.loss_table(rows)
    assert list(df.columns) == ["epoch", "acoustic_G_loss", "visual_G_loss", "text_G_loss", "visual_D_loss",
                                "text_D_loss", "acoustic_D_loss"] and len(df) == 1


def test_report_text_and_metrics_match_reference(g, tmp_path):
    from gan_ffn_amd import artifacts as A
    labels, preds, masks = g["report/labels"], g["report/preds"], g["report/masks"]
    text, f1 = A.report_text(float(g["report/best_loss"]), labels, preds, masks)
    assert f1 == float(g["report/f1"]) and text == str(g["report/text"])
    name, f1b = A.write_test_report(float(g["report/best_loss"]), labels, preds, masks, 150, str(tmp_path) + "/")
    assert os.path.basename(name) == "test_out_GAN-epochs=150_F1-score=%s.txt" % f1 and open(name).read() == text
    # epoch metrics: loss weighted by real utterances, 4 / 2 digit rounding
    losses = [1.5 * masks[:100].sum(), 0.5 * masks[100:].sum()]
    avg_loss, acc, f = A.epoch_metrics(losses, labels, preds, masks)
    assert avg_loss == round((losses[0] + losses[1]) / masks.sum(), 4)
    assert acc == float(g["report/acc"]) and f == f1


def test_checkpoints_names_roundtrip_and_foreign_pickles(tmp_path):
    """whole-module pickles under the reference's six names; a stock-PyTorch module with the reference's state_dict
    keys (what a reference checkpoint contains) loads into the build's classes"""
    from gan_ffn_amd import artifacts as A, model as M
    from oracle import stock_modules as SM
    torch.manual_seed(5)
    mods = [M.AcousticGenerator(100), M.AcousticDiscriminator(100), M.VisualGenerator(100), M.VisualDiscriminator(100),
            M.TextGenerator(100), M.TextDiscriminator(100)]
    sp = str(tmp_path) + "/GAN_save_"
    A.save_GAN_models(mods, sp)
    assert sorted(os.listdir(tmp_path)) == sorted("GAN_save_" + n + ".pth" for n in A.MODEL_NAMES)
    gens, discs = A.load_GAN_models(sp, device="cpu")
    assert not gens["visual"].training
    for a, b in ((mods[2], gens["visual"]), (mods[5], discs["text"])):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    # foreign pickles: stock nn.TransformerEncoder modules, one of them wrapped in nn.DataParallel like the
    # reference's GPU checkpoints (train_IEMOCAP.py:587-593)
    names = ["AcousticGenerator", "AcousticDiscriminator", "VisualGenerator", "VisualDiscriminator", "TextGenerator",
             "TextDiscriminator"]
    stock = [SM.StockNet(n) for n in names]
    fp = str(tmp_path) + "/ref_"
    for i, (n, m) in enumerate(zip(A.MODEL_NAMES, stock)):
        torch.save(torch.nn.DataParallel(m) if i == 2 else m, fp + n + ".pth")
    gens, discs = A.load_GAN_models(fp, device="cpu")
    assert type(gens["visual"]) is M.VisualGenerator and type(discs["acoustic"]) is M.AcousticDiscriminator
    ssd = stock[2].state_dict()
    bsd = gens["visual"].state_dict()
    assert set(ssd) == set(bsd)
    for k in ssd:
        assert torch.equal(ssd[k].reshape(-1), bsd[k].reshape(-1).cpu()), k
