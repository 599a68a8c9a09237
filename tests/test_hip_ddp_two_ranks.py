"""Row (e) on the device: two data-parallel ranks of the GAN engine (each owning half of the dialogues, one process
each, sharing this box's single GPU; gloo process group because RCCL wants one GPU per rank) against one process on
the global batch.  Checks: the two replicas stay bit-identical after the 12 all-reduced sub-steps; the mean of the
ranks' losses is the global-batch loss; the parameters land where the single process puts them (up to the Adam
first-step noise the other trajectory tests document)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, tmp, n_streams, tag):
    port = _free_port()
    procs, outs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", GANFFN_COMM_PER_STREAM="1" if (n_streams > 1 and world == 2) else "0")
        out = os.path.join(tmp, "%s_r%d.pt" % (tag, r))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_gpu_worker.py"), out, str(n_streams)], env=env))
    for p in procs:
        assert p.wait(timeout=280) == 0
    return [torch.load(o) for o in outs]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_streams", [1, 3])
def test_two_ranks_match_single_process_global_batch(tmp_path, n_streams):
    two = _run(2, str(tmp_path), n_streams, "w2")
    one = _run(1, str(tmp_path), n_streams, "w1")[0]
    # replicas: bit-identical parameters on both ranks (same all-reduced gradients, same Adam)
    for k in two[0]["sd"]:
        assert torch.equal(two[0]["sd"][k], two[1]["sd"][k]), k
    # losses: mean over ranks of the local means == global mean (equal local S*B)
    l2 = (two[0]["losses"] + two[1]["losses"]) / 2
    d = (l2 - one["losses"]).abs().numpy()
    assert d[:2].max() < 2e-6 and d[:9].max() < 1e-4 and d.max() < 5e-3, d
    # parameters: where the single process put them
    for k, v in one["sd"].items():
        delta = (two[0]["sd"][k] - v).abs().max()
        # one Adam step of lr 1e-4 (1.1e-4 / 0.5e-4): a sign flip of a near-zero gradient moves a weight by <= 2 lr
        assert float(delta) <= 4.5e-4, (k, float(delta))
    moved = max(float((two[0]["sd"][k] - v).abs().max()) for k, v in one["sd"].items())
    assert np.isfinite(moved)


@pytest.mark.timeout(600)
def test_four_ranks_one_dialogue_each(tmp_path):
    """world 4 (one dialogue per rank), 3 sub-step streams, default single communicator"""
    four = _run(4, str(tmp_path), 3, "w4")
    one = _run(1, str(tmp_path), 3, "w1")[0]
    for r in range(1, 4):
        for k in four[0]["sd"]:
            assert torch.equal(four[0]["sd"][k], four[r]["sd"][k]), (r, k)
    l4 = sum(f["losses"] for f in four) / 4
    d = (l4 - one["losses"]).abs().numpy()
    assert d[:2].max() < 2e-6 and d[:9].max() < 1e-4 and d.max() < 5e-3, d
