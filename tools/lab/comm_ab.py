"""Data-parallel GAN step with ONE communicator for all sub-step streams against one communicator PER stream
(GANFFN_COMM_PER_STREAM=1): W processes share this box's single GPU (gloo process groups — RCCL wants one GPU per rank), every
rank runs the 3-stream engine on its own B_local dialogues; wall-clock per iteration, max over ranks.
    python tools/lab/comm_ab.py            (driver: spawns the workers for W = 2, 4 and both settings)"""
import os, socket, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)


def worker():
    import torch
    import torch.distributed as dist
    from gan_ffn_amd import data as D, engine as E
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gens, discs = E.build_networks(100, 0.2, "cuda", seed=99)
    batch = D.synthetic_batch(B=int(os.environ["B_LOCAL"]), S_max=94, seed=21 + rank, device="cuda")
    eng = E.GanEngine(gens, discs, process_group=dist.group.WORLD, n_streams=3)
    for _ in range(2):
        eng.iteration(batch)
    eng.synchronize(); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        eng.iteration(batch)
    eng.synchronize(); torch.cuda.synchronize(); dist.barrier()
    dt = torch.tensor([(time.perf_counter() - t0) / n * 1e3], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print("world %d, B_local %s, comm per stream %s: %.1f ms / iteration" % (world, os.environ["B_LOCAL"], os.environ.get("GANFFN_COMM_PER_STREAM", "0"), float(dt)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


if __name__ == "__main__":
    if "RANK" in os.environ:
        worker()
        sys.exit(0)
    for world, bl in ((2, 8), (4, 4)):
        for rep in range(2):
            for per in ("0", "1"):
                port = free_port()
                procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)],
                                          env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                                   HSA_ENABLE_IPC_MODE_LEGACY="0", GANFFN_COMM_PER_STREAM=per, B_LOCAL=str(bl)))
                         for r in range(world)]
                for p in procs:
                    assert p.wait(timeout=500) == 0
