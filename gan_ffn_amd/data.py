"""Synthetic IEMOCAP-schema batches (the dataset pickle is absent: /root/reference/.MISSING_LARGE_BLOBS)
and a loader for the real pickle when it is supplied.

Schema follows /root/reference/dataloader.py:11-13,41-58: text (S,B,100), visual (S,B,512),
audio (S,B,100), qmask (S,B,2), umask (B,S), label (B,S); features are min-max normalised to [0,1]
per dialogue (dataloader.py:20-35) and zero-padded to the longest dialogue of the batch.
"""
import numpy as np
import torch

DIMS = {"text": 100, "visual": 512, "acoustic": 100}


def dialogue_lengths(B, S_max=94, seed=3407, lo=8, mean=48):
    """Seeded lengths mimicking IEMOCAP (min ~8, mean ~48, max 110); one dialogue is forced to S_max so the
    batch pads to S = S_max (94 is the reference's own example, model.py:1437)."""
    rng = np.random.default_rng(seed)
    L = np.clip(rng.gamma(shape=4.0, scale=(mean - lo) / 4.0, size=B) + lo, lo, S_max).astype(np.int64)
    L[int(rng.integers(0, B))] = S_max
    return L


def synthetic_batch(B=32, S_max=94, seed=3407, device="cpu", n_classes=6):
    """-> dict(text, visual, acoustic, qmask, umask, label, lengths); uniform[0,1) features, zero padding."""
    L = dialogue_lengths(B, S_max, seed)
    S = int(L.max())
    g = torch.Generator().manual_seed(seed)
    out = {}
    valid = (torch.arange(S).unsqueeze(1) < torch.from_numpy(L).unsqueeze(0)).float()       # (S, B)
    for k, d in DIMS.items():
        out[k] = (torch.rand(S, B, d, generator=g) * valid.unsqueeze(-1)).to(device).contiguous()
    spk = torch.randint(0, 2, (S, B), generator=g)
    out["qmask"] = (torch.nn.functional.one_hot(spk, 2).float() * valid.unsqueeze(-1)).to(device)
    out["umask"] = valid.t().contiguous().to(device)
    out["label"] = (torch.randint(0, n_classes, (B, S), generator=g) * valid.t().long()).to(device)
    out["lengths"] = L
    return out


def shard_batch(batch, rank, world):
    """data-parallel shard along the DIALOGUE axis (dim 1 of the (S,B,.) tensors, dim 0 of umask/label).
    Never the sequence axis — that is the reference's nn.DataParallel bug (train_IEMOCAP.py:587-593)."""
    B = batch["text"].shape[1]
    assert B % world == 0, "global batch %d not divisible by world %d" % (B, world)
    b0, b1 = rank * (B // world), (rank + 1) * (B // world)
    out = {}
    for k, v in batch.items():
        if k in ("text", "visual", "acoustic", "qmask"):
            out[k] = v[:, b0:b1].contiguous()
        elif k in ("umask", "label"):
            out[k] = v[b0:b1].contiguous()
        else:
            out[k] = v[b0:b1]
    return out


def load_iemocap_pickle(path, train=True):
    """Real-data path: same unpickling + per-dialogue global min-max normalisation as dataloader.py:10-39.
    Returns a list of (text, visual, audio, speakers, labels, vid) numpy tuples."""
    import pickle
    (ids, speakers, labels, text, audio, visual, sentence, train_vid, test_vid) = pickle.load(open(path, "rb"), encoding="latin1")

    def norm(a):
        a = np.asarray(a, dtype=np.float64)
        return ((a - a.min()) / (a.max() - a.min())).astype(np.float32)

    items = []
    for vid in (train_vid if train else test_vid):
        items.append((norm(text[vid]), norm(visual[vid]), norm(audio[vid]),
                      np.array([[1, 0] if x == "M" else [0, 1] for x in speakers[vid]], dtype=np.float32),
                      np.asarray(labels[vid], dtype=np.int64), vid))
    return items


def collate(items, device="cpu"):
    """pad-collate like dataloader.py:55-58 (features seq-first, umask/label batch-first)."""
    B = len(items)
    S = max(len(it[4]) for it in items)
    out = {"text": torch.zeros(S, B, 100), "visual": torch.zeros(S, B, 512), "acoustic": torch.zeros(S, B, 100),
           "qmask": torch.zeros(S, B, 2), "umask": torch.zeros(B, S), "label": torch.zeros(B, S, dtype=torch.long)}
    for b, (t, v, a, q, y, _) in enumerate(items):
        n = len(y)
        out["text"][:n, b] = torch.from_numpy(t)
        out["visual"][:n, b] = torch.from_numpy(v)
        out["acoustic"][:n, b] = torch.from_numpy(a)
        out["qmask"][:n, b] = torch.from_numpy(q)
        out["umask"][b, :n] = 1
        out["label"][b, :n] = torch.from_numpy(y)
    return {k: v.to(device) for k, v in out.items()}
