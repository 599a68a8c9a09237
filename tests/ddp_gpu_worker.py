"""Worker of tests/test_hip_ddp_two_ranks.py: one data-parallel rank of the GAN engine on the (shared) GPU, gloo
process group (RCCL needs one GPU per rank; the data-parallel logic under test is backend-agnostic)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.position_encoding.dropout.p = 0.0
    m.transformer_encoder.enc_dropout = 0.0


def run(rank, world, out_path, n_streams):
    from gan_ffn_amd import data as D, engine as E
    pg = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
    gens, discs = E.build_networks(100, 0.2, "cuda", seed=99)         # identical replicas on every rank
    for m in list(gens.values()) + list(discs.values()):
        zero_dropout(m)
    full = D.synthetic_batch(B=4, S_max=12, seed=21, device="cuda")
    batch = D.shard_batch(full, rank, world) if world > 1 else full
    eng = E.GanEngine(gens, discs, process_group=pg, n_streams=n_streams)
    eng.iteration(batch)
    eng.synchronize()
    torch.cuda.synchronize()
    losses = eng.losses.detach().cpu().clone()
    sd = {k: v.detach().cpu().clone() for k, v in gens["visual"].state_dict().items() if "layers.7." in k or k.startswith("fc")}
    sd.update({"D." + k: v.detach().cpu().clone() for k, v in discs["text"].state_dict().items() if "layers.0.self_attn" in k})
    torch.save({"losses": losses, "sd": sd}, out_path)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), sys.argv[1], int(sys.argv[2]))
