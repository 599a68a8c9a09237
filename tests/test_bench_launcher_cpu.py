"""`python bench.py --gpus N` starts its own N ranks and survives a bad RCCL day: the launcher half,
exercised on the CPU with a stand-in child (the real children need one MI355X each).  Replaces the reference's
multi-GPU entry, /root/reference/train_IEMOCAP.py:587-593."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_2_builds_two_rank_environments():
    envs = bench.rank_environments(2, 29999, {"GANFFN_DP_MODE": "inline"}, base={"PATH": "/usr/bin"})
    assert [e["RANK"] for e in envs] == ["0", "1"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1"]
    for e in envs:
        assert e["WORLD_SIZE"] == "2" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999"
        assert e["GANFFN_DP_MODE"] == "inline" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["GANFFN_BENCH_CHILD"] == "1"


def test_child_arguments_keep_the_callers_and_override_streams():
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--streams", "3", "--launcher", "spawn"]
    assert bench._child_argv(argv, None) == ["--gpus", "8", "--steps", "20", "--warmup", "5", "--streams", "3"]
    assert bench._child_argv(argv, 1) == ["--gpus", "8", "--steps", "20", "--warmup", "5", "--streams", "1"]
    assert bench._child_argv(["--gpus=2", "--streams=3"], 1) == ["--gpus=2", "--streams", "1"]


def test_ladder_order_is_inline3_inline1_buckets():
    assert [r[0] for r in bench.LADDER] == ["inline-3streams", "inline-1stream", "buckets"]
    assert bench.LADDER[1][2] == 1 and bench.LADDER[2][1]["GANFFN_DP_MODE"] == "buckets"


CHILD = textwrap.dedent('''
    import json, os, sys, time
    rank, world, rung = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ["GANFFN_BENCH_RUNG"]
    print("[child] rank %d of %d on rung %s argv %s" % (rank, world, rung, sys.argv[1:]), file=sys.stderr, flush=True)
    mode = os.environ["FAKE_MODE"]
    if mode == "hang-then-crash-then-ok":
        if rung == "inline-3streams":
            time.sleep(600)                      # a hung collective: silent for ever
        if rung == "inline-1stream":
            if rank == 1:
                sys.exit(3)                      # a rank that dies ...
            time.sleep(8)                        # ... while the others are still inside the collective with it
    if rank == 0:
        print(json.dumps({"metric": "m", "value": 1.0, "n_gpus": world, "config": {"dp_mode": os.environ.get("GANFFN_DP_MODE"),
                          "streams": sys.argv[sys.argv.index("--streams") + 1] if "--streams" in sys.argv else None,
                          "fallback_from": os.environ.get("GANFFN_BENCH_FALLBACK_FROM")}}), flush=True)
''')


def _run_launcher(tmp_path, mode, silence="3"):
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    code = ("import sys, argparse; sys.path.insert(0, %r); import bench; "
            "a = argparse.Namespace(gpus=2); sys.exit(bench.launch(a, ['--gpus', '2', '--steps', '2'], cmd=[sys.executable, %r]))"
            % (ROOT, str(child)))
    env = dict(os.environ, FAKE_MODE=mode, GANFFN_LAUNCH_SILENCE_S=silence)
    env.pop("GANFFN_DP_MODE", None)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)


def test_launcher_relays_rank0_line(tmp_path):
    r = _run_launcher(tmp_path, "ok")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["launcher"]["spawned_ranks"] == 2
    assert d["config"]["launcher"]["rung"] == "inline-3streams" and d["config"]["launcher"]["fallback_from"] is None
    assert d["config"]["dp_mode"] == "inline"
    assert "rank 1 of 2" in r.stderr                   # both ranks were started, stderr is relayed


def test_launcher_falls_down_the_ladder_with_fresh_children(tmp_path):
    """rung 1 hangs (silent) -> its children are ended; rung 2 loses a rank -> ended; rung 3 (buckets) delivers the line"""
    r = _run_launcher(tmp_path, "hang-then-crash-then-ok")
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["launcher"]["rung"] == "buckets"
    assert d["config"]["launcher"]["fallback_from"] == ["inline-3streams", "inline-1stream"]
    assert d["config"]["dp_mode"] == "buckets" and d["config"]["fallback_from"] == "inline-3streams,inline-1stream"
    assert "no output from any rank" in r.stderr and "exited with" in r.stderr
    # the second rung really ran on one stream
    assert "rung inline-1stream argv ['--gpus', '2', '--steps', '2', '--streams', '1']" in r.stderr


def _run_supervisors(tmp_path, mode, silence="3"):
    """two ranks of an external launcher (what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` starts):
    each process calls bench.supervise with a stand-in worker; they share the coordination directory"""
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    coord = tmp_path / "coord"
    code = ("import sys, argparse; sys.path.insert(0, %r); import bench; a = argparse.Namespace(gpus=2); "
            "sys.exit(bench.supervise(a, ['--gpus', '2', '--steps', '2'], cmd=[sys.executable, %r], coord_dir=%r))" % (ROOT, str(child), str(coord)))
    procs = []
    for r in range(2):
        env = dict(os.environ, FAKE_MODE=mode, GANFFN_LAUNCH_SILENCE_S=silence, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29987")
        env.pop("GANFFN_DP_MODE", None)
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p_.communicate(timeout=180) for p_ in procs]
    return [p_.returncode for p_ in procs], outs


def test_supervisors_under_an_external_launcher_relay_rank0_line(tmp_path):
    rcs, outs = _run_supervisors(tmp_path, "ok")
    assert rcs == [0, 0], outs
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["config"]["launcher"]["rung"] == "inline-3streams" and d["config"]["launcher"]["spawned_ranks"] == 2
    assert d["config"]["dp_mode"] == "inline"


def test_supervisors_fall_down_the_ladder_together(tmp_path):
    """the driver starts N > 1 through torch.distributed.run: the ladder must work there too — rung 1 hangs on both ranks (each
    supervisor's own watchdog fires), rung 2 loses rank 1 (rank 1's supervisor marks the rung failed, rank 0's ends its healthy
    worker), rung 3 delivers the line; both supervisors exit 0"""
    rcs, outs = _run_supervisors(tmp_path, "hang-then-crash-then-ok")
    assert rcs == [0, 0], outs
    d = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert d["config"]["launcher"]["rung"] == "buckets"
    assert d["config"]["launcher"]["fallback_from"] == ["inline-3streams", "inline-1stream"]
    assert d["config"]["dp_mode"] == "buckets" and d["config"]["fallback_from"] == "inline-3streams,inline-1stream"
    assert "rung inline-1stream argv ['--gpus', '2', '--steps', '2', '--streams', '1']" in outs[0][1]
