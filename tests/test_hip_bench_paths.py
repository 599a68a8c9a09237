"""GPU: bench.py contract (one JSON line with roofline + cpu_baseline) and the RCCL code path exercised on
a single GPU (1-rank nccl group, launched exactly as the driver launches N > 1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_default_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--cpu-sample-batch", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["value"] > 50 * d["cpu_baseline"]["value"]          # north_star: >= 50x the host CPU


def test_bench_rccl_path_on_one_rank():
    env = dict(os.environ, GANFFN_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["config"]["launch"] == "eager" and d["config"]["parallelism"] == "dp1"
    assert all(0.1 < v < 5 for v in d["config"]["last_losses"].values())


@pytest.mark.parametrize("config", ["meld", "drnn"])
def test_bench_other_configs_on_the_rccl_path(config):
    """configs[2] (MELD widths) and configs[4] (GAN-FFN + DialogueRNN) through the same launcher and the 1-rank nccl
    group: the line keeps the contract's keys and a roofline object"""
    env = dict(os.environ, GANFFN_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29519" if config == "meld" else "29521",
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--config", config], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "dtype", "config", "roofline"):
        assert k in d, k
    assert d["value"] > 0 and d["config"]["parallelism"] == "dp1" and 0 < d["roofline"]["frac"] < 1


def test_data_parallel_path_costs_nothing_at_one_rank():
    """the N = 1 point of the scaling curve: the data-parallel code path (1-rank RCCL group: in-line all-reduce of every
    sub-step's gradient slab on the sub-step's own stream, one communicator per stream) against the plain engine, 3 streams,
    step only, best of two interleaved runs each: within 3 %.  Rounds 1-3 issued 4-5 asynchronous
    all-reduces per sub-step on the process group's internal stream: 18-75 % slower at one rank (tools/dist1_ab.sh)."""
    def run(dist, port):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if dist:
            env.update(GANFFN_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--step-only", "--steps", "10"],
                           capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        return _last_json(r.stdout)["ms_per_step"]
    plain, dist1 = [], []
    for i in range(2):
        plain.append(run(False, 0))
        dist1.append(run(True, 29601 + i))
    assert min(dist1) <= 1.03 * min(plain), (plain, dist1)
