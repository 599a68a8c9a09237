"""Synthetic IEMOCAP-schema batches (the dataset pickle is absent: /root/reference/.MISSING_LARGE_BLOBS)
and a loader for the real pickle when it is supplied.

Schema follows /root/reference/dataloader.py:11-13,41-58: text (S,B,100), visual (S,B,512),
audio (S,B,100), qmask (S,B,2), umask (B,S), label (B,S); features are min-max normalised to [0,1]
per dialogue (dataloader.py:20-35) and zero-padded to the longest dialogue of the batch.
"""
import numpy as np
import torch
import torch.utils.data

DIMS = {"text": 100, "visual": 512, "acoustic": 100}
# MELD feature widths (train_MELD.py:143 text D_m = 600; audio 300 per dataloader.py:93-95's videoAudio); MELD has no visual
# modality.  Used only by the MELD-dimension extension workload (BASELINE.json configs[2]).
MELD_DIMS = {"text": 600, "acoustic": 300}


def dialogue_lengths(B, S_max=94, seed=3407, lo=8, mean=48):
    """Seeded lengths mimicking IEMOCAP (min ~8, mean ~48, max 110); one dialogue is forced to S_max so the
    batch pads to S = S_max (94 is the reference's own example, model.py:1437)."""
    rng = np.random.default_rng(seed)
    L = np.clip(rng.gamma(shape=4.0, scale=(mean - lo) / 4.0, size=B) + lo, lo, S_max).astype(np.int64)
    L[int(rng.integers(0, B))] = S_max
    return L


def synthetic_batch(B=32, S_max=94, seed=3407, device="cpu", n_classes=6, dims=None, lo=8, mean=48):
    """-> dict(text, visual, acoustic, qmask, umask, label, lengths); uniform[0,1) features, zero padding.
    dims: modality -> feature width (default: IEMOCAP's)."""
    L = dialogue_lengths(B, S_max, seed, lo=lo, mean=mean)
    S = int(L.max())
    g = torch.Generator().manual_seed(seed)
    out = {}
    valid = (torch.arange(S).unsqueeze(1) < torch.from_numpy(L).unsqueeze(0)).float()       # (S, B)
    for k, d in (dims or DIMS).items():
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


def _minmax(a):
    """(v - min v) / (max v - min v) over the WHOLE dialogue array, in the array's own dtype
    (dataloader.py:20-35 — one global min/max per dialogue and modality, not per feature)"""
    a = np.asarray(a)
    lo, hi = np.min(a), np.max(a)
    return (a - lo) / (hi - lo)


class IEMOCAPDataset(torch.utils.data.Dataset):
    """Same pickle schema, normalisation, item tuple and collate as the reference's IEMOCAPDataset
    (/root/reference/dataloader.py:8-58): item = (text (L,100), visual (L,512), audio (L,100), speaker one-hot
    (L,2) ['M' -> [1,0]], umask ones (L), labels (L) int64, vid)."""

    def __init__(self, path, train=True):
        import pickle
        (self.videoIDs, self.videoSpeakers, self.videoLabels, self.videoText, self.videoAudio, self.videoVisual,
         self.videoSentence, self.trainVid, self.testVid) = pickle.load(open(path, "rb"), encoding="latin1")
        for d in (self.videoText, self.videoAudio, self.videoVisual):     # every dialogue, train and test alike
            for key in d.keys():
                d[key] = _minmax(d[key])
        self.keys = [x for x in (self.trainVid if train else self.testVid)]
        self.len = len(self.keys)

    def __getitem__(self, index):
        vid = self.keys[index]
        return (torch.FloatTensor(self.videoText[vid]), torch.FloatTensor(self.videoVisual[vid]),
                torch.FloatTensor(self.videoAudio[vid]),
                torch.FloatTensor([[1, 0] if x == "M" else [0, 1] for x in self.videoSpeakers[vid]]),
                torch.FloatTensor([1] * len(self.videoLabels[vid])), torch.LongTensor(self.videoLabels[vid]), vid)

    def __len__(self):
        return self.len

    @staticmethod
    def collate_fn(data):
        """[text, visual, audio, qmask] padded seq-first (S,B,.), [umask, label] batch-first (B,S), vids list
        (dataloader.py:55-58)"""
        from torch.nn.utils.rnn import pad_sequence
        cols = list(zip(*data))
        return [pad_sequence(list(cols[i])) if i < 4 else pad_sequence(list(cols[i]), True) if i < 6 else list(cols[i])
                for i in range(len(cols))]


class MELDDataset(torch.utils.data.Dataset):
    """/root/reference/dataloader.py:90-124: item = (text, audio, speakers, umask ones, labels, vid); no normalisation,
    no visual modality; `classify` picks emotion (7-way) or sentiment (3-way) labels."""

    def __init__(self, path, classify="emotion", train=True):
        import pickle
        (self.videoIDs, self.videoSpeakers, self.emotion_labels, self.videoText, self.videoAudio, self.videoSentence,
         self.trainVid, self.testVid, self.sentiment_labels) = pickle.load(open(path, "rb"))
        self.videoLabels = self.emotion_labels if classify == "emotion" else self.sentiment_labels
        self.keys = [x for x in (self.trainVid if train else self.testVid)]
        self.len = len(self.keys)

    def __getitem__(self, index):
        vid = self.keys[index]
        return (torch.FloatTensor(self.videoText[vid]), torch.FloatTensor(self.videoAudio[vid]),
                torch.FloatTensor(self.videoSpeakers[vid]), torch.FloatTensor([1] * len(self.videoLabels[vid])),
                torch.LongTensor(self.videoLabels[vid]), vid)

    def __len__(self):
        return self.len

    @staticmethod
    def collate_fn(data):
        from torch.nn.utils.rnn import pad_sequence
        cols = list(zip(*data))
        return [pad_sequence(list(cols[i])) if i < 3 else pad_sequence(list(cols[i]), True) if i < 5 else list(cols[i])
                for i in range(len(cols))]


def get_train_valid_sampler(trainset, valid=0.1):
    """train_IEMOCAP.py:62-66: the FIRST `valid` fraction of the training dialogues is the validation split"""
    from torch.utils.data.sampler import SubsetRandomSampler
    size = len(trainset)
    idx = list(range(size))
    split = int(valid * size)
    return SubsetRandomSampler(idx[split:]), SubsetRandomSampler(idx[:split])


def _loaders(trainset, testset, batch_size, valid, num_workers, pin_memory):
    from torch.utils.data import DataLoader
    ts, vs = get_train_valid_sampler(trainset, valid)
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=pin_memory)
    return (DataLoader(trainset, sampler=ts, collate_fn=trainset.collate_fn, **kw),
            DataLoader(trainset, sampler=vs, collate_fn=trainset.collate_fn, **kw),
            DataLoader(testset, collate_fn=testset.collate_fn, **kw))


def get_IEMOCAP_loaders(path, batch_size=32, valid=0.2, num_workers=0, pin_memory=False):
    """(train_loader, valid_loader, test_loader), train_IEMOCAP.py:69-100"""
    return _loaders(IEMOCAPDataset(path, True), IEMOCAPDataset(path, False), batch_size, valid, num_workers, pin_memory)


def get_MELD_loaders(path, n_classes=7, batch_size=32, valid=0.1, num_workers=0, pin_memory=False, classify="emotion"):
    return _loaders(MELDDataset(path, classify, True), MELDDataset(path, classify, False), batch_size, valid,
                    num_workers, pin_memory)


def to_batch(collated, device="cuda"):
    """one collated IEMOCAP batch (list, as the loaders yield it) -> the dict the engines take, moved to `device`
    once (the reference moves and re-types each tensor per batch, train_IEMOCAP.py:333-351)"""
    textf, visuf, acouf, qmask, umask, label = collated[:6]
    mv = lambda t: t.to(device, non_blocking=True)
    return {"text": mv(textf.float()).contiguous(), "visual": mv(visuf.float()).contiguous(),
            "acoustic": mv(acouf.float()).contiguous(), "qmask": mv(qmask), "umask": mv(umask.float()).contiguous(),
            "label": mv(label.long()).contiguous(), "vids": collated[6] if len(collated) > 6 else None}


def write_synthetic_iemocap_pickle(path, n_train=12, n_test=5, seed=3407, lo=4, hi=23, dtype=np.float64):
    """A small pickle with the IEMOCAP schema and seeded content (tests, smoke runs; the real one is not shipped)."""
    import pickle
    rng = np.random.default_rng(seed)
    ids, spk, lab, txt, aud, vis, sen = {}, {}, {}, {}, {}, {}, {}
    names = ["Ses%02d_%03d" % (i % 5 + 1, i) for i in range(n_train + n_test)]
    for n in names:
        L = int(rng.integers(lo, hi + 1))
        ids[n] = ["%s_%d" % (n, j) for j in range(L)]
        spk[n] = [("M" if x else "F") for x in rng.integers(0, 2, L)]
        lab[n] = [int(x) for x in rng.integers(0, 6, L)]
        txt[n] = (rng.standard_normal((L, 100)) * 3 - 1).astype(dtype)
        aud[n] = (rng.standard_normal((L, 100)) * 0.5 + 2).astype(dtype)
        vis[n] = (rng.random((L, 512)) * 7).astype(dtype)
        sen[n] = ["utt %d" % j for j in range(L)]
    with open(path, "wb") as f:
        pickle.dump((ids, spk, lab, txt, aud, vis, sen, names[:n_train], names[n_train:]), f)
    return names[:n_train], names[n_train:]


def load_iemocap_pickle(path, train=True):
    """Real-data path: same unpickling + per-dialogue global min-max normalisation as dataloader.py:10-39.
    Returns a list of (text, visual, audio, speakers, labels, vid) numpy tuples."""
    import pickle
    (ids, speakers, labels, text, audio, visual, sentence, train_vid, test_vid) = pickle.load(open(path, "rb"), encoding="latin1")

    def norm(a):
        return _minmax(a).astype(np.float32)

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
