"""N > 1 path on CPU: world_size-2 gloo processes.  The compute kernels need a GPU, so the oracle stands in
for them here; what is under test is the data-parallel logic itself: dialogue-axis sharding, the bucket
ranges, the async bucketed all-reduce (GradReducer) and the sum-then-1/world convention — the averaged
per-rank gradient must equal the single-process gradient of the global batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gan_ffn_amd import data as D
    from gan_ffn_amd.engine import GradReducer, bucket_ranges
    from oracle import ganffn_oracle as O
    from util import formula_sd
    import formula as F_

    # global batch: 4 dialogues, S = 6; rank r owns dialogues [2r, 2r+2)
    S, B = 6, 4
    full = {"text": torch.from_numpy(F_.formula_input("ddp.text", S, B, 100, pad_from=4))}
    shard = D.shard_batch(full, rank, world)
    assert shard["text"].shape == (S, B // world, 100)
    assert torch.equal(shard["text"], full["text"][:, rank * 2:(rank + 1) * 2])

    n_layers = 2
    sd = {k: v for k, v in formula_sd("TextDiscriminator", n_layers).items()}
    net = O.OracleNet("disc", sd, 10, 0.2, torch.float64, n_layers=n_layers)

    def flat_grad(x):
        for p in net.parameters():
            p.grad = None
        prob = net(x.double())
        O.bce_mean(prob, torch.ones_like(prob)).backward()
        keys = [k for k in net.P if k.startswith("transformer_encoder.") or k.startswith("fc")]
        return torch.cat([net.P[k].grad.reshape(-1) for k in keys]), keys

    g_local, keys = flat_grad(shard["text"])
    total = g_local.numel()
    layer_floats = sum(net.P[k].numel() for k in keys if k.startswith("transformer_encoder.layers.0."))
    enc = n_layers * layer_floats
    ranges = bucket_ranges(enc, 0, total, layer_floats, n_layers, n_buckets=2)
    # ranges tile the slab exactly once, head first then layers from last to first
    cover = np.zeros(total, dtype=np.int32)
    for lo, hi in ranges:
        cover[lo:hi] += 1
    assert (cover == 1).all() and ranges[0] == (enc, total) and ranges[1][1] == enc

    red = GradReducer(dist.group.WORLD)
    for lo, hi in ranges:
        red.reduce_async(g_local[lo:hi], lo, hi)
    # per-bucket optimizer step: the callback runs bucket by bucket, in issue order, right after that bucket's
    # all-reduce has been waited for (the engine applies Adam to the slice there); stand-in update: p -= 0.1 * g / world
    param = torch.zeros_like(g_local)
    seen = []

    order = []

    class _Waited:
        """records WHEN a bucket's all-reduce is waited for (the overlap claim: Adam of bucket k is enqueued before bucket
        k + 1's all-reduce is waited for, so only the last bucket's reduce is exposed)"""

        def __init__(self, work, tag):
            self.work, self.tag = work, tag

        def wait(self):
            order.append(("wait", self.tag))
            return self.work.wait()
    red.works = [(_Waited(w, (lo, hi)), lo, hi) for (w, lo, hi) in red.works]

    def on_bucket(lo, hi):
        seen.append((lo, hi))
        order.append(("adam", (lo, hi)))
        param[lo:hi] -= 0.1 * g_local[lo:hi] / red.world
    red.finish(on_bucket)
    assert seen == list(ranges) and red.works == []
    assert order == [step for r_ in ranges for step in (("wait", r_), ("adam", r_))], order
    g_avg = g_local / red.world                       # Adam's grad_scale = 1/world
    assert torch.equal(param, -0.1 * g_avg)           # bucket-wise update == whole-slab update

    g_full, _ = flat_grad(full["text"])               # what one process would compute on the global batch
    err = float((g_avg - g_full).abs().max() / g_full.abs().max())
    torch.save({"err": err, "world": red.world}, os.path.join(tmp, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_equals_global_batch_gradient(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert res["world"] == 2
        # equal shard sizes -> mean of per-rank BCE means == global mean -> averaged grads are the global-batch grads
        assert res["err"] < 1e-10, res


def test_shard_batch_rejects_uneven_split_and_never_splits_sequence():
    from gan_ffn_amd import data as D
    b = D.synthetic_batch(B=6, S_max=12, seed=1)
    with pytest.raises(AssertionError):
        D.shard_batch(b, 0, 4)
    s0, s1 = D.shard_batch(b, 0, 2), D.shard_batch(b, 1, 2)
    assert s0["text"].shape[0] == b["text"].shape[0] == s1["visual"].shape[0]      # S untouched
    assert torch.equal(torch.cat([s0["acoustic"], s1["acoustic"]], 1), b["acoustic"])
    assert torch.equal(torch.cat([s0["umask"], s1["umask"]], 0), b["umask"])


def test_synthetic_batch_schema():
    from gan_ffn_amd import data as D
    b = D.synthetic_batch(B=32, S_max=94, seed=3407)
    S = 94
    assert b["text"].shape == (S, 32, 100) and b["visual"].shape == (S, 32, 512) and b["acoustic"].shape == (S, 32, 100)
    assert b["qmask"].shape == (S, 32, 2) and b["umask"].shape == (32, S) and b["label"].shape == (32, S)
    assert int(b["lengths"].max()) == 94 and int(b["lengths"].min()) >= 8
    # zero padding beyond each dialogue's length; features in [0, 1)
    for j in range(32):
        L = int(b["lengths"][j])
        assert float(b["text"][L:, j].abs().sum()) == 0.0 and float(b["umask"][j, :L].sum()) == L
    assert float(b["visual"].min()) >= 0.0 and float(b["visual"].max()) < 1.0
