"""Training-level (statistical) parity: the GAN phase and the phase-2 classifier trained with dropout ON by the HIP
engines for 8 seeds on a learnable synthetic IEMOCAP-schema set, against the SAME protocol run on the host with stock
PyTorch modules (tests/golden/train_stats.npz, produced in the build container by tests/golden/make_train_stats.py cpu).
Dropout streams differ by construction (torch CPU generator vs Philox), initial weights are identical per seed, so the
comparison is of distributions: for every recorded metric — the six GAN losses at iterations 1, 4, 8, 12, 16; the phase-2
training loss at steps 1, 20, 40, 60, 80, 100; test loss, accuracy and weighted F1 — the HIP mean over seeds must lie within
4 standard errors of the CPU mean (plus a small absolute allowance).  The F1 target of the reference
(/root/reference/README.md:9-31) needs the IEMOCAP pickle, which is absent: this is the evidence that can be had."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_hip_training_statistics_match_the_stock_cpu_runs():
    import make_train_stats as MTS
    g = np.load(os.path.join(HERE, "golden", "train_stats.npz"))
    names, cpu = [str(x) for x in g["names"]], g["cpu"]
    assert names == MTS.metric_names() and list(g["seeds"]) == MTS.SEEDS
    lines = []
    names_h, hip = MTS.run("hip", log=lines.append)
    assert names_h == names and hip.shape == cpu.shape
    n = cpu.shape[0]
    bad = []
    for j, name in enumerate(names):
        mc, mh = cpu[:, j].mean(), hip[:, j].mean()
        se = np.sqrt(cpu[:, j].var(ddof=1) / n + hip[:, j].var(ddof=1) / n)
        atol = 3.0 if name.endswith(("acc", "f1")) else 3e-3            # percent points / loss units
        if abs(mc - mh) > 4 * se + atol:
            bad.append("%s: cpu %.4f +- %.4f, hip %.4f +- %.4f" % (name, mc, cpu[:, j].std(ddof=1), mh, hip[:, j].std(ddof=1)))
    assert not bad, "\n".join(bad + lines)
    # the task is learnable and both implementations learn it: the phase-2 training loss falls, by a similar amount
    j0, j1 = names.index("p2/train_loss/step1"), names.index("p2/train_loss/step%d" % MTS.N_P2)
    drop_c, drop_h = (cpu[:, j0] - cpu[:, j1]).mean(), (hip[:, j0] - hip[:, j1]).mean()
    assert drop_c > 0.5 and drop_h > 0.5 and abs(drop_c - drop_h) < 0.35 * max(drop_c, drop_h), (drop_c, drop_h)
