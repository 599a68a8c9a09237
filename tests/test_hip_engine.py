"""GPU parity of the fast step runner (gan_ffn_amd.engine) — the counterpart of the reference's
train_disc / train_gen / train_GAN — against the golden fixtures produced by the reference's own
functions (dropout p = 0), plus graph-replay == eager and train-mode sanity."""
import numpy as np
import pytest
import torch

import formula as F_
from util import DIN, DISC, GEN, formula_sd, golden
from test_oracle_golden import GAN_LOSS_TOL, check_first_update

pytestmark = pytest.mark.gpu


def build_all(zero_dropout):
    from gan_ffn_amd import model
    nets = {}
    for grp, table in (("G", GEN), ("D", DISC)):
        nets[grp] = {}
        for k, cls in table.items():
            m = getattr(model, cls)(100, dropout=0.2)
            m.load_state_dict({a: torch.from_numpy(b) for a, b in formula_sd(cls).items()}, strict=False)
            if zero_dropout:
                m.dropout.p = 0.0
                m.position_encoding.dropout.p = 0.0
                m.transformer_encoder.enc_dropout = 0.0
            nets[grp][k] = m.cuda()
    return nets["G"], nets["D"]


def gan_batch(S=7, B=2):
    return {k: torch.from_numpy(F_.formula_input("gan." + k, S, B, DIN[k], pad_from=5)).cuda() for k in DIN}


def test_engine_reproduces_reference_gan_trajectory():
    from gan_ffn_amd import engine
    g = golden("gan_steps")
    gens, discs = build_all(zero_dropout=True)
    eng = engine.GanEngine(gens, discs)           # lr 1e-4, betas (0.5, 0.6): train_IEMOCAP.py:603-606
    batch = gan_batch()
    eng._prepare(7, 2)
    losses, seen = [], set()
    for it in range(2):
        eng._adds = 0
        for i, (kind, who, partner) in enumerate(engine.SCHEDULE):
            (eng.train_disc if kind == "D" else eng.train_gen)(who, partner, batch, i)
            losses.append(float(eng.losses[i]))
            if (kind, who) not in seen:
                seen.add((kind, who))
                net = (discs if kind == "D" else gens)[who]
                sd = dict(net.named_parameters())
                check_first_update(g, kind, who, lambda k: sd[k].detach().cpu().numpy())
    err = np.abs(np.array(losses) - g["gan/losses"])
    assert (err <= np.array(GAN_LOSS_TOL)).all(), err


def test_graph_replay_matches_eager():
    """hipGraph replay against eager launches, EXACTLY (train mode, dropout on).  The first graph-mode call runs the
    iteration once eagerly as warm-up (dropout offsets 0 .. 47, then the device-side offset is bumped by 48), captures
    (capture executes nothing) and replays once, so replay k is the (k + 1)-th execution of the iteration: same
    parameter versions, same effective Philox offsets (device offset 48 (k + 1) + v against the eager engine's host
    block 48 (k + 1) + v) and the same deterministic kernels -> the same bits as eager iteration k + 1."""
    from gan_ffn_amd import engine, ops
    batch = gan_batch(S=9, B=2)
    res = []
    for use_graph, n in ((False, 4), (True, 3)):
        gens, discs = build_all(zero_dropout=False)
        ops.manual_seed(1234)
        eng = engine.GanEngine(gens, discs, use_graph=use_graph)
        assert eng.use_graph == use_graph
        out = []
        for _ in range(n):
            out.append(eng.iteration(batch).clone())
        torch.cuda.synchronize()
        res.append(torch.stack(out).cpu().numpy())
        final = {k: m.slab.detach().cpu().clone() for k, m in list(gens.items()) + [("D" + k, v) for k, v in discs.items()]}
        res.append(final)
    eager, eager_final, replay, replay_final = res
    assert np.isfinite(eager).all() and np.isfinite(replay).all()
    for k in range(3):
        assert np.array_equal(replay[k], eager[k + 1]), (k, replay[k], eager[k + 1])
    assert not np.allclose(replay[0], replay[1])               # the replay advances the dropout stream
    for k in eager_final:                                       # 4 executions each: identical parameters, bit for bit
        assert torch.equal(eager_final[k], replay_final[k]), k


def test_train_mode_losses_are_plausible_and_masks_advance():
    from gan_ffn_amd import engine, ops
    gens, discs = build_all(zero_dropout=False)
    ops.manual_seed(7)
    eng = engine.GanEngine(gens, discs)
    batch = gan_batch(S=12, B=4)
    a = eng.iteration(batch).clone()
    b = eng.iteration(batch).clone()
    d = eng.loss_dict()
    assert set(d) == {"acoustic_G_loss", "visual_G_loss", "text_G_loss", "visual_D_loss", "text_D_loss", "acoustic_D_loss"}
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert (a > 0.3).all() and (a < 2.5).all()
    # eager mode takes one block of dropout offsets per iteration from the device's allocator (graph mode bumps the
    # device-side offset instead)
    assert eng._adds == 6 * 4 + 6 * 4 and eng._base_add == eng._adds and ops.DeviceRng.get("cuda").counter == 2 * eng._adds
    assert not torch.allclose(a, b)


@pytest.mark.parametrize("n_streams,use_graph", [(2, False), (3, False)])
def test_multi_stream_schedule_matches_sequential(n_streams, use_graph):
    """sub-steps overlapped on several HIP streams see the same parameter versions as the sequential order"""
    from gan_ffn_amd import engine
    batch = gan_batch(S=11, B=2)
    out = {}
    for ns, ug in ((1, False), (n_streams, use_graph)):
        gens, discs = build_all(zero_dropout=True)
        eng = engine.GanEngine(gens, discs, n_streams=ns, use_graph=ug)
        ls = []
        for _ in range(2):
            losses = eng.iteration(batch)
            eng.synchronize()             # side streams -> current stream before reading the loss slots
            ls.append(losses.clone())
        eng.synchronize()
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().clone() for k, v in gens["visual"].state_dict().items()}
        out[(ns, ug)] = (torch.stack(ls).cpu().numpy(), sd)
    a, b = out[(1, False)], out[(n_streams, use_graph)]
    if not use_graph:
        # No kernel accumulates with atomics and every sub-step sees exactly the parameter versions of the sequential
        # order, so the overlapped schedule is BIT-identical to the sequential one: losses of both iterations and the
        # final parameters.
        assert np.array_equal(a[0], b[0]), np.abs(a[0] - b[0])
        for k in a[1]:
            assert torch.equal(a[1][k], b[1][k]), k
    else:
        assert np.isfinite(b[0]).all()


@pytest.mark.parametrize("n_streams", [1, 3])
def test_iteration_is_bit_reproducible(n_streams):
    """two runs from the same weights, batch and dropout seed (train mode, dropout ON) give identical losses and identical
    parameters of all six networks: weight / bias / LayerNorm gradients and the loss means are computed without
    floating-point atomics (owner-accumulated tiles, partial sums reduced in a fixed order)"""
    from gan_ffn_amd import engine, ops
    batch = gan_batch(S=19, B=4)
    res = []
    for _ in range(2):
        gens, discs = build_all(zero_dropout=False)
        ops.manual_seed(2024)
        eng = engine.GanEngine(gens, discs, n_streams=n_streams)
        ls = []
        for _ in range(2):
            losses = eng.iteration(batch)
            eng.synchronize()
            ls.append(losses.clone())
        torch.cuda.synchronize()
        sds = {("G", k): {n: v.detach().cpu().clone() for n, v in m.state_dict().items()} for k, m in gens.items()}
        sds.update({("D", k): {n: v.detach().cpu().clone() for n, v in m.state_dict().items()} for k, m in discs.items()})
        res.append((torch.stack(ls).cpu(), sds))
    assert torch.equal(res[0][0], res[1][0])
    for key in res[0][1]:
        for n in res[0][1][key]:
            assert torch.equal(res[0][1][key][n], res[1][1][key][n]), (key, n)


@pytest.mark.parametrize("S,B,n_streams", [(40, 8, 1), (40, 8, 3), (94, 32, 3)])
def test_unreduced_weight_gradient_chunks_give_the_reduce_launchs_bits(S, B, n_streams, monkeypatch):
    """round 5: on one GPU the d_model-100 weight gradients stay as token-chunk slabs that the Adam launch adds itself
    (ganffn_encoder_bwd_parts + ganffn_adam_step_parts: no tn100_reduce launch, no zero-fill of the gradient slab) — the same
    sum in the same association, so two iterations leave EVERY parameter of all six networks and every loss bit-identical to
    the path through the reduce launch (GANFFN_ADAM_PARTS=0).  (40, 8): the generators' passes (320 tokens) run one chunk,
    the discriminators' [real | fake] passes (640 tokens) two; (94, 32): three chunks everywhere."""
    from gan_ffn_amd import engine, ops
    batch = gan_batch(S=S, B=B)
    res = []
    for parts in ("1", "0"):
        monkeypatch.setenv("GANFFN_ADAM_PARTS", parts)
        gens, discs = build_all(zero_dropout=False)
        ops.manual_seed(77)
        eng = engine.GanEngine(gens, discs, n_streams=n_streams)
        ls = []
        for _ in range(2):
            losses = eng.iteration(batch)
            eng.synchronize()
            ls.append(losses.clone())
        torch.cuda.synchronize()
        slabs = {("G", k): m.slab.detach().cpu().clone() for k, m in gens.items()}
        slabs.update({("D", k): m.slab.detach().cpu().clone() for k, m in discs.items()})
        res.append((torch.stack(ls).cpu(), slabs))
        if parts == "1":
            assert eng._parts_ok(eng.D["text"], eng.pass_D2["text"]) and not eng._parts_ok(eng.G["visual"], eng.pass_G["visual"])
    assert torch.equal(res[0][0], res[1][0]), (res[0][0] - res[1][0]).abs().max()
    for key in res[0][1]:
        assert torch.equal(res[0][1][key], res[1][1][key]), key


def test_phase2_step_matches_reference_fixture():
    """GAN_FFN forward + MaskedNLLLoss + backward (A11) through the fast runner vs the reference's own numbers"""
    from gan_ffn_amd import engine, model
    from util import check_summary
    g = golden("misc")
    gens, _ = build_all(zero_dropout=True)
    net = model.GAN_FFN(gens["acoustic"], gens["visual"], gens["text"], n_classes=6).cuda()
    with torch.no_grad():
        net.fc.weight.copy_(torch.from_numpy(F_.formula_tensor("phase2.fc.weight", (6, 100))))
        net.fc.bias.copy_(torch.from_numpy(F_.formula_tensor("phase2.fc.bias", (6,))))
    eng = engine.Phase2Engine(net, lr=0.0, weight_decay=0.0)       # lr 0: keep the weights, inspect the gradients
    batch = gan_batch(S=7, B=2)
    batch["umask"] = torch.from_numpy(g["phase2/umask"]).cuda()
    batch["label"] = torch.from_numpy(g["phase2/label"]).cuda()
    loss, lp = eng.step(batch, train=True)
    torch.cuda.synchronize()
    assert np.abs(lp.cpu().numpy() - g["phase2/log_prob"]).max() < 1e-4
    assert abs(float(loss) - float(g["phase2/loss_weighted"])) < 1e-4
    gw = eng.fc_grad[:600].view(6, 100).cpu().numpy()
    assert np.abs(gw - g["phase2/grad_fc_weight"]).max() < 2e-5
    check_summary(g, "phase2/grad_text_fc2_weight", eng.G["text"].w("fc2.weight", True).view(100, 512), rtol=1e-3, atol=1e-8)
    n = 3 * 512 * 512
    check_summary(g, "phase2/grad_visual_l0_inproj", eng.G["visual"].grad[:n].view(1536, 512), rtol=1e-3, atol=1e-9)
    # the same through the nn.Module mirror + autograd
    net2_gens, _ = build_all(zero_dropout=True)
    net2 = model.GAN_FFN(net2_gens["acoustic"], net2_gens["visual"], net2_gens["text"]).cuda().eval()
    with torch.no_grad():
        net2.fc.weight.copy_(net.fc.weight)
        net2.fc.bias.copy_(net.fc.bias)
    lp2, _, _, _ = net2(batch["acoustic"], batch["visual"], batch["text"])
    lp2_ = lp2.transpose(0, 1).contiguous().view(-1, 6)
    w = torch.tensor(engine.CLASS_WEIGHTS, device="cuda")
    l2 = model.MaskedNLLLoss(w)(lp2_, batch["label"].view(-1), batch["umask"])
    assert abs(float(l2) - float(g["phase2/loss_weighted"])) < 1e-4
    l2.backward()
    assert np.abs(net2.fc.weight.grad.cpu().numpy() - g["phase2/grad_fc_weight"]).max() < 2e-5
    # eval step: loss only, predictions in the reference's batch-major order
    loss_e, lp_e = eng.step(batch, train=False)
    pred = engine.Phase2Engine.predictions(lp_e)
    assert pred.shape == (14,) and abs(float(loss_e) - float(loss)) < 1e-5


def test_phase2_training_reduces_loss():
    from gan_ffn_amd import engine, model, ops
    from gan_ffn_amd import data as D
    gens, _ = engine.build_networks(device="cuda", seed=11)
    net = model.GAN_FFN(gens["acoustic"], gens["visual"], gens["text"]).cuda()
    ops.manual_seed(5)
    eng = engine.Phase2Engine(net)                                   # lr 1e-4, l2 0.008 (train_IEMOCAP.py:453-456,661)
    batch = D.synthetic_batch(B=4, S_max=16, device="cuda")
    first = float(eng.step(batch)[0])
    for _ in range(25):
        last = float(eng.step(batch)[0])
    assert np.isfinite(last) and last < first


def test_multi_stream_with_fresh_batches_of_changing_length():
    """real loaders hand over a NEW batch (new tensors, new S) every iteration and drop the old one while the side
    streams may still be working on it: buffers are re-made per shape and batch tensors are recorded on the side
    streams.  Same data through 1 and 3 streams: first iteration equal, everything finite."""
    from gan_ffn_amd import engine
    outs = {}
    for ns in (1, 3):
        gens, discs = build_all(zero_dropout=True)
        eng = engine.GanEngine(gens, discs, n_streams=ns)
        ls = []
        for it, S in enumerate((11, 7, 13, 7)):
            batch = gan_batch(S=S, B=2)                       # fresh tensors each time
            batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
            eng.iteration(batch)
            del batch
            junk = torch.full((1 << 20,), float("nan"), device="cuda")   # try to grab the freed memory right away
            del junk
            ls.append(eng.loss_dict())
        outs[ns] = ls
    for a, b in zip(outs[1], outs[3]):
        for k in a:
            assert np.isfinite(a[k]) and np.isfinite(b[k])
    for k in outs[1][0]:
        assert abs(outs[1][0][k] - outs[3][0][k]) < 2e-3, k


def test_buffers_are_reused_for_shorter_batches():
    """capacity sizing: a shorter batch re-views the same storage (no allocation), a longer one re-allocates; results
    do not depend on the capacity the buffers happen to have"""
    from gan_ffn_amd import engine
    res = {}
    for cap in (None, 21):
        gens, discs = build_all(zero_dropout=True)
        eng = engine.GanEngine(gens, discs, n_streams=1)
        if cap:
            eng._prepare(cap, 2)
            ptr0 = eng.pass_G["text"]._saved.data_ptr()
        eng.iteration(gan_batch(S=9, B=2))
        if cap:
            assert eng.pass_G["text"]._saved.data_ptr() == ptr0 and eng._cap_S == 21 and eng._shape == (9, 2)
        res[cap] = eng.loss_dict()
        eng.iteration(gan_batch(S=13, B=2))
        assert eng._cap_S == (21 if cap else 13)
    for k in res[None]:
        assert abs(res[None][k] - res[21][k]) < 2e-3, k


def test_short_last_batch_keeps_the_reserved_buffers():
    """real loaders have no drop_last: every epoch ends on a short batch (IEMOCAP: 32, 32, 32, 12).  The step buffers
    are sized once (reserve) and re-viewed for every (S, B) within that capacity — no allocation, no device sync — and
    the results equal those of a fresh engine sized exactly for each batch."""
    from gan_ffn_amd import engine
    shapes = [(9, 4), (13, 2), (7, 4), (11, 3)]
    gens, discs = build_all(zero_dropout=True)
    eng = engine.GanEngine(gens, discs, n_streams=1)
    eng.reserve(16, 4)
    got, ptrs = [], None
    for S, B in shapes:
        eng.iteration(gan_batch(S=S, B=B))
        got.append(eng.loss_dict())
        cur = (eng.pass_G["text"]._saved.data_ptr(), eng.pass_D2["visual"]._saved.data_ptr(), eng._scratch_flat[0]["ws"].data_ptr())
        assert ptrs is None or cur == ptrs
        ptrs = cur
        assert (eng._cap_S, eng._cap_B) == (16, 4) and eng._shape == (S, B)
    # fresh engines, one per shape, run the same trajectory on exactly-sized buffers
    gens, discs = build_all(zero_dropout=True)
    for i, (S, B) in enumerate(shapes):
        eng2 = engine.GanEngine(gens, discs, n_streams=1)
        if i > 0:   # carry the optimizer state over (a fresh engine starts Adam at t = 0)
            for k in eng2.G:
                for a in ("exp_avg", "exp_avg_sq", "step"):
                    getattr(eng2.G[k], a).copy_(state["G"][k][a])
            for k in eng2.D:
                for a in ("exp_avg", "exp_avg_sq", "step"):
                    getattr(eng2.D[k], a).copy_(state["D"][k][a])
        eng2.iteration(gan_batch(S=S, B=B))
        want = eng2.loss_dict()
        state = {"G": {k: {a: getattr(n, a).clone() for a in ("exp_avg", "exp_avg_sq", "step")} for k, n in eng2.G.items()},
                 "D": {k: {a: getattr(n, a).clone() for a in ("exp_avg", "exp_avg_sq", "step")} for k, n in eng2.D.items()}}
        tol = 5e-5 if i == 0 else 0.15          # later batches: Adam chaos (same bound as GAN_LOSS_TOL[12:])
        for k in want:
            assert abs(want[k] - got[i][k]) < tol, (i, k, want[k], got[i][k])


def test_engines_of_one_process_share_their_side_streams():
    """the side streams are chosen once per process (by timing, engine._tune_streams) and every later engine of comparable
    size runs on the same ones: on ROCm a second set of streams created later can land on hardware queues that serialise
    the sub-step chains (DESIGN.md §4 iv).  A much bigger engine re-times the choice."""
    from gan_ffn_amd import engine
    handles = []
    for S, B in ((11, 2), (13, 2), (94, 8)):
        gens, discs = build_all(zero_dropout=True)
        eng = engine.GanEngine(gens, discs, n_streams=3)
        eng.iteration(gan_batch(S=S, B=B))
        eng.synchronize()
        torch.cuda.synchronize()
        # (three sub-step streams; with the early generator forward on, three helper streams follow them in the list)
        assert len(eng.streams) in (3, 6) and len({s.cuda_stream for s in eng.streams}) == len(eng.streams)
        handles.append([s.cuda_stream for s in eng.streams])
    assert handles[0] == handles[1]
    key = (str(eng.dev), (0, 0, -1) * (2 if eng.early_gen else 1))
    assert engine._STREAMS[key][1] >= 94 * 8 or handles[2] == handles[0]


def test_two_engines_over_the_same_networks_need_a_synchronize():
    """Two eager 3-stream engines built over the SAME six networks share the process's side streams but each keeps its own
    dependency records (and its own Adam state: engine._side_streams docstring).  With `synchronize()` between them the
    hand-over is race-free: the sequence "two iterations by engine A, synchronize, two by engine B" gives the same bits
    every time it is run from the same state and dropout seed."""
    from gan_ffn_amd import engine, ops
    batch = gan_batch(S=11, B=2)
    out = []
    for rep in range(2):
        gens, discs = build_all(zero_dropout=False)
        ops.manual_seed(4321)
        a = engine.GanEngine(gens, discs, n_streams=3)
        b = engine.GanEngine(gens, discs, n_streams=3)
        for i in range(4):
            (a if i < 2 else b).iteration(batch)
            if i == 1:
                a.synchronize()            # the contract: before another engine touches the same networks
                torch.cuda.synchronize()
        b.synchronize()
        torch.cuda.synchronize()
        out.append({k: m.slab.detach().cpu().clone() for k, m in list(gens.items()) + [("D" + k, v) for k, v in discs.items()]})
        assert all(bool(torch.isfinite(v).all()) for v in out[-1].values())
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k
