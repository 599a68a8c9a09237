"""Fast step runner: the counterpart of the reference's train_disc / train_gen / train_GAN
(/root/reference/train_IEMOCAP.py:200-393), driving libganffn.so directly.

Differences from running the nn.Module mirror under torch autograd — all result-preserving:
  * no autograd graph: forward/backward are explicit C-ABI calls on preallocated buffers;
  * train_disc evaluates D(real) and D(fake) as ONE pass over a [real | fake] batch of 2B dialogues
    (dialogues are independent in every op; loss = ganffn_bce2 = (BCE(real,1)+BCE(fake,0))/2);
  * the generator forward inside train_disc keeps nothing for backward (the reference builds and
    discards that graph, train_IEMOCAP.py:218-219);
  * train_gen skips the frozen discriminator's weight gradients (computed but never used by the
    reference: zero_grad at train_IEMOCAP.py:216 clears them before D's next step);
  * losses stay on the device; the host reads all 12 once per iteration instead of 12 syncs
    (train_IEMOCAP.py:224,249);
  * gradients, Adam moments and parameters are flat slabs -> one fused Adam launch and, with
    world_size > 1, a bucketed all-reduce (RCCL) per sub-step;
  * optionally the whole iteration is captured once into a hipGraph and replayed (launch-bound otherwise).
"""
import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import HeadCfg

# sub-step schedule of one batch, train_IEMOCAP.py:355-382: (kind, trained net, partner net)
SCHEDULE = [
    ("D", "visual", "acoustic"), ("G", "acoustic", "visual"),
    ("D", "visual", "text"), ("G", "text", "visual"),
    ("D", "text", "acoustic"), ("G", "acoustic", "text"),
    ("D", "acoustic", "text"), ("G", "text", "acoustic"),
    ("D", "text", "visual"), ("G", "visual", "text"),
    ("D", "acoustic", "visual"), ("G", "visual", "acoustic"),
]
LOSS_COLUMNS = ["acoustic_G_loss", "visual_G_loss", "text_G_loss", "visual_D_loss", "text_D_loss",
                "acoustic_D_loss"]  # train_IEMOCAP.py:308-316


class NetState:
    """One network on the device: parameter slab (shared with the nn.Module), gradient slab, Adam state."""

    def __init__(self, module, lr, betas, weight_decay=0.0):
        self.m = module
        self.slab = module.slab
        assert self.slab.is_cuda, "NetState needs the module on the GPU"
        total, views, enc = module.slab_layout()
        self.total, self.views, self.enc_floats = total, views, enc
        self.grad = torch.zeros_like(self.slab)
        self.exp_avg = torch.zeros_like(self.slab)
        self.exp_avg_sq = torch.zeros_like(self.slab)
        self.step = torch.zeros(1, dtype=torch.int32, device=self.slab.device)
        self.lr, self.betas, self.wd = lr, betas, weight_decay
        self.E, self.H, self.L = module.d_model, module.nhead, module.num_layers
        self.kind = 0 if module.KIND == "gen" else 1
        named = {}
        params = module._slab_params()
        names = {id(p): n for n, p in module.named_parameters()}
        for p, (off, shape) in zip(params, views):
            named[names[id(p)]] = (off, shape)
        self.named = named
        self.pe = module.position_encoding.pe
        self.p_head = float(module.dropout.p)
        self.p_pe = float(module.position_encoding.dropout.p)
        self.p_enc = float(module.transformer_encoder.enc_dropout)
        self.D1 = module.fc1.weight.shape[0]
        self.D2 = module.fc2.weight.shape[0]
        self.layer_floats = enc // self.L
        self.covered = int(_lib.load().ganffn_encoder_bwd_parts_covered(self.E, 2048))    # chunked head of a layer block (weights, biases)
        self.has_obj = bool(module.HAS_OBJECT)
        self.obj_in = int(module.OBJECT_IN) if self.has_obj else 0       # raw-modality width `object` maps to D_h
        # `object` (weight [D_h x obj_in] | bias [D_h], each padded to 4 floats) sits right after the layers
        self.obj_floats = sum((int(p.numel()) + 3) & ~3 for p in (module.object.weight, module.object.bias)) if self.has_obj else 0

    def w(self, name, grad=False):
        off, shape = self.named[name]
        n = 1
        for d in shape:
            n *= d
        return (self.grad if grad else self.slab)[off:off + n]

    def buckets(self, n_buckets):
        """[(lo, hi)] float ranges of the grad slab, in the order backward completes them:
        head (+object) first, then encoder layers from the last to the first."""
        return bucket_ranges(self.enc_floats, self.obj_floats, self.total, self.layer_floats, self.L, n_buckets)


def dp_mode():
    """how gradients cross the ranks (GANFFN_DP_MODE):
    "inline" (default, round 4): ONE all-reduce of the network's whole gradient slab per sub-step, issued as a synchronous
        collective — torch >= 2.7 runs those on the CURRENT stream, i.e. in-line on the sub-step's own HIP stream, between its
        backward and its Adam: no internal communication stream, no cross-stream events; the all-reduce of one sub-step runs
        beside the compute of the sub-steps on the other streams (each stream has its own communicator, so collectives of
        different streams never share one);
    "buckets" (rounds 1-3): 4-5 asynchronous all-reduces per sub-step on the process group's internal stream, each issued
        right after its layer range's backward, Adam per bucket.  Measured with a 1-rank RCCL group on one MI355X
        (tools/dist1_ab.sh): 40.6-60.6 ms per step against 34.5 ms for the plain engine — the internal stream's event waits
        share hardware queues with the sub-step streams and hold them up; see DESIGN.md section 7."""
    return os.environ.get("GANFFN_DP_MODE", "inline")


class GradReducer:
    """Bucketed gradient all-reduce for data parallelism over the DIALOGUE axis (one process per GPU).
    Each call sums one contiguous slice of a gradient slab across ranks, asynchronously (on RCCL's own
    stream for the nccl backend), so it overlaps whatever backward work is enqueued next; finish()
    makes the current stream wait for all of them.  The 1/world factor is applied by Adam (grad_scale).
    Backend-agnostic (nccl on MI355X, gloo in the CPU tests)."""

    def __init__(self, process_group):
        import torch.distributed as dist
        self.dist, self.pg = dist, process_group
        self.world = dist.get_world_size(process_group)
        self.works = []

    def reduce_async(self, flat_slice, lo=None, hi=None):
        w = self.dist.all_reduce(flat_slice, op=self.dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self.works.append((w, lo, hi))

    def finish(self, on_bucket=None):
        """wait for the buckets in issue order; on_bucket(lo, hi) runs right after a bucket's all-reduce has been waited
        for (the engine applies Adam to that slice there, so only the LAST bucket's reduce is exposed)"""
        for w, lo, hi in self.works:
            w.wait()
            if on_bucket is not None and lo is not None:
                on_bucket(lo, hi)
        self.works = []


def bucket_ranges(enc_floats, obj_floats, total, layer_floats, L, n_buckets):
    """[(lo, hi)] float ranges of a gradient slab in the order backward completes them: fc head first,
    then groups of encoder layers from the last layer to the first (`object`, whose gradient is produced
    last, is reduced separately by the caller)."""
    out = [(enc_floats + obj_floats, total)]
    per = max(1, (L + n_buckets - 1) // max(1, n_buckets))
    hi = L
    while hi > 0:
        lo = max(0, hi - per)
        out.append((lo * layer_floats, hi * layer_floats))
        hi = lo
    return out


class _Pass:
    """Buffers of one forward(+backward) pass of one network for up to `B` dialogues of up to `S` steps.  Storage is
    flat and sized for that capacity; `resize(S, B)` re-derives the configs and the shaped views for a smaller batch
    without touching the allocator (real loaders deliver a different S every iteration and a short last batch)."""

    def __init__(self, net, S, B, dev, need_bwd, S_cap=None):
        self.net, self.B, self.dev, self.need_bwd = net, B, dev, need_bwd
        self.S_cap = max(S, S_cap or 0)
        self.B_cap = B
        E = net.E
        cfg = ops.enc_cfg(self.S_cap, B, E, net.H, net.L, train=True, p_pe=net.p_pe, p_enc=net.p_enc)
        n_saved, n_ws = ops.enc_sizes(cfg)
        h_saved, h_ws = ops.head_sizes(HeadCfg(self.S_cap * B, E, net.D1, net.D2, net.kind, net.p_head, 1))
        f32 = dict(device=dev, dtype=torch.float32)
        Tc = self.S_cap * B
        self._enc_out = torch.empty(Tc * E, **f32)
        self._saved = torch.empty(n_saved, **f32) if need_bwd else None
        self._hsaved = torch.empty(h_saved, **f32)
        self._out = torch.empty(Tc * (net.D2 if net.kind == 0 else 1), **f32)
        self._dx = torch.empty(Tc * E, **f32) if need_bwd else None
        self.n_ws = max(n_ws, h_ws)          # at capacity: workspace needs grow with S
        self.resize(S)

    def resize(self, S, B=None):
        B = self.B_cap if B is None else B
        assert S <= self.S_cap and B <= self.B_cap
        net = self.net
        self.B = B
        E = net.E
        self.S, self.T = S, S * B
        self.cfg_train = ops.enc_cfg(S, B, E, net.H, net.L, train=True, p_pe=net.p_pe, p_enc=net.p_enc)
        self.cfg_eval = ops.enc_cfg(S, B, E, net.H, net.L, train=False, p_pe=net.p_pe, p_enc=net.p_enc)
        n_saved, _ = ops.enc_sizes(self.cfg_train)
        self.hcfg_train = HeadCfg(self.T, E, net.D1, net.D2, net.kind, net.p_head, 1)
        self.hcfg_eval = HeadCfg(self.T, E, net.D1, net.D2, net.kind, net.p_head, 0)
        h_saved, _ = ops.head_sizes(self.hcfg_train)
        Do = net.D2 if net.kind == 0 else 1
        self.enc_out = self._enc_out[:self.T * E].view(S, B, E)
        self.saved = self._saved[:n_saved] if self.need_bwd else None
        self.hsaved = self._hsaved[:h_saved]
        self.out = self._out[:self.T * Do].view(S, B, Do)
        self.dx = self._dx[:self.T * E].view(S, B, E) if self.need_bwd else None


class _Res:
    """A resource touched by sub-steps on different HIP streams (a network's parameter slab, a pass buffer):
    the event of its last writer and the events of the readers since then."""

    def __init__(self):
        self.last_write = None
        self.reads = []


# Sub-step -> stream map for n_streams = 2: the visual generator's chain (the long pole: its two D steps and two
# G steps are 45 % of the work) runs beside everything else; dependencies are enforced by events, so any map is
# correct — this one balances the two streams (tools: see DESIGN.md §4).
STREAM_MAP = {1: [0] * 12,
              2: [0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1],     # measured best of six 2-stream maps (60.9 ms vs 81.6 ms on 1)
              3: [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 2, 2]}     # visual generator's chain on its own stream: 49.9 ms

# Bi-modal schedule of the MELD-dimension extension workload (BASELINE.json configs[2]; MELD has text and audio only,
# dataloader.py:93-95): the text/acoustic sub-steps 5-8 of the reference's order (train_IEMOCAP.py:363-370).
ADDS_PER_SUBSTEP = 4     # dropout-bearing launches of one sub-step (two encoder passes and two heads)
SCHEDULE_BIMODAL = [s for s in SCHEDULE if "visual" not in s[1:]]
STREAM_MAP_BIMODAL = {1: [0, 0, 0, 0], 2: [0, 0, 1, 1]}


_STREAMS = {}


def _side_streams(dev, prios, tuner=None, work=1):
    """the side streams of a device, chosen once per process and shared by every engine built afterwards.

    Which hardware queue a new HIP stream lands on depends on how many streams the process created and used before
    (torch's pools, RCCL, other engines), and with it how well kernels of different streams overlap: the same three
    sub-step chains ran at 34.7, 37.1, 40.5, 43.5 or 54 ms per iteration depending on nothing but that
    (tools/lab/stream_order.py; 1-workgroup spin kernels overlap on every pair — only real launch mixes tell the pairs
    apart).  `tuner(prios)` picks the streams by timing a probe workload on fresh candidates (GanEngine._tune_streams);
    without it, or with GANFFN_STREAM_TUNE=0, fresh streams are taken as they come; `work` = tokens per pass of the engine
    that asks (the choice is re-timed when a much bigger engine comes along).
    Sharing the streams between the engines of one process is safe for engines over DIFFERENT networks and buffers (each
    engine orders its own sub-steps with its own events).  Two engines over the SAME networks see only their own dependency
    records: DrnnEngine joins its streams at the end of every step, but an eager multi-stream GanEngine does not (consecutive
    iterations overlap) — call `eng.synchronize()` (or `loss_dict()`) before another engine touches the same networks
    (tests/test_hip_engine.py::test_two_engines_over_the_same_networks_need_a_synchronize)."""
    key = (str(dev), tuple(prios))
    if os.environ.get("GANFFN_STREAM_CACHE", "1") == "1" and key in _STREAMS and work <= 4 * _STREAMS[key][1]:
        return _STREAMS[key][0]                 # (a choice timed on a much smaller batch is not trusted for a big one)
    # (no timing probes while the caller is capturing a graph: they synchronise)
    if tuner is not None and len(prios) > 1 and os.environ.get("GANFFN_STREAM_TUNE", "1") == "1" and \
            not torch.cuda.is_current_stream_capturing():
        streams = tuner(prios)
    else:
        streams = [torch.cuda.Stream(device=dev, priority=p_) for p_ in prios]
    _STREAMS[key] = (streams, max(1, work))
    return streams


class _Runner:
    """network-level forward / backward / Adam on preallocated buffers — shared by the GAN step runner and the
    phase-2 (classifier) step runner"""
    n_streams = 1
    early_gen = False
    _cur_stream = None
    _base_add = 0
    _adds = 0

    def _check_slabs(self):
        """the engine trains the slab it captured at construction; a module that re-packed into a NEW slab since then
        (`.to(other device)`, `.double()`, load into fresh parameters) would silently stop following — refuse instead"""
        for group in (getattr(self, "G", {}), getattr(self, "D", {})):
            for k, st in group.items():
                if st.m.slab.data_ptr() != st.slab.data_ptr():
                    raise RuntimeError("network %r re-packed its parameters after the engine was built (module.slab moved): "
                                       "build the engine after the last .to()/.cuda()/dtype change" % k)

    def _init_common(self, device, process_group, n_buckets):
        self.dev = device
        self.rng = ops.DeviceRng.get(device)
        self.pg = process_group
        self.world = 1
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
        self.n_buckets = n_buckets


class GanEngine(_Runner):
    """train_GAN's inner loop (train_IEMOCAP.py:320-382) for a fixed batch shape.

    n_streams > 1: independent sub-steps run concurrently on several HIP streams.  The 12 sub-steps form a
    DAG over the six networks (e.g. (D_t|G_a) only needs G_a from sub-step 2 and can run beside
    (D_v|G_t)/(G_t|D_v)); every sub-step waits for the writers of what it reads and, before its Adam, for
    the readers of what it writes — so each one sees exactly the parameter versions of the sequential
    schedule.  Most kernels of this workload fill a fraction of the 256 CUs, so overlapping them is free."""

    def __init__(self, gens, discs, lr=1e-4, b1=0.5, b2=0.6, process_group=None, n_buckets=3, use_graph=False,
                 n_streams=1, schedule=None):
        # optimizers: train_IEMOCAP.py:292-297 (G lr, text-G 1.1*lr, every D lr/2); call site :603-606
        self.G = {k: NetState(m, lr * (1.1 if k == "text" else 1.0), (b1, b2)) for k, m in gens.items()}
        self.D = {k: NetState(m, lr / 2, (b1, b2)) for k, m in discs.items()}
        self._init_common(next(iter(self.G.values())).slab.device, process_group, n_buckets)
        self.modalities = list(self.G.keys())
        if schedule is None:
            schedule = SCHEDULE if set(self.modalities) == {"acoustic", "visual", "text"} else \
                [s_ for s_ in SCHEDULE if s_[1] in self.G and s_[2] in self.G]
        self.schedule = list(schedule)
        self.D_h = next(iter(self.G.values())).D2          # width of the fused feature = every discriminator's d_model
        stream_maps = STREAM_MAP if len(self.schedule) == 12 else \
            {1: [0] * len(self.schedule), 2: [(i // 2) % 2 for i in range(len(self.schedule))]}
        self.use_graph = use_graph
        if n_streams not in stream_maps:
            n_streams = max(k for k in stream_maps if k <= max(1, n_streams))
        self.n_streams = n_streams
        if use_graph and self.n_streams > 1:
            # multi-stream capture is not used: replay == eager here (the step is GPU-bound, not launch-bound), and
            # eager streams additionally overlap consecutive iterations
            self.use_graph = use_graph = False
        self.stream_map = stream_maps[self.n_streams]
        self.early_gen = self.n_streams > 1 and os.environ.get("GANFFN_EARLY_GEN", "0") == "1"
        if os.environ.get("GANFFN_STREAM_MAP"):
            self.stream_map = [int(x) for x in os.environ["GANFFN_STREAM_MAP"].split(",")]
            assert len(self.stream_map) == len(self.schedule) and max(self.stream_map) < self.n_streams
        self.streams = None
        self._res = {}
        self._base_add = 0
        # One communicator per sub-step stream (default in the in-line mode, see dp_mode(); GANFFN_COMM_PER_STREAM overrides).
        self.pgs = [process_group]
        per_stream = os.environ.get("GANFFN_COMM_PER_STREAM", "1" if dp_mode() == "inline" else "0") == "1"
        if process_group is not None and self.n_streams > 1 and not per_stream and dp_mode() == "inline":
            import torch.distributed as dist
            if dist.get_backend(process_group) == "nccl":
                # in-line collectives of different sub-step streams would share ONE RCCL communicator and may be in flight
                # together: RCCL (like NCCL) does not support that.  (gloo reduces on the host, synchronously: no such limit.)
                raise RuntimeError("GANFFN_DP_MODE=inline with %d sub-step streams needs one communicator per stream: "
                                   "leave GANFFN_COMM_PER_STREAM at 1, or use n_streams=1, or GANFFN_DP_MODE=buckets" % self.n_streams)
        # Ordering assumption of the in-line mode: every rank runs the SAME host program, so the
        # collectives of the three communicators are issued in the same host order on every rank; whatever order a rank's
        # hardware queues impose is a sub-order of that one, so no two ranks can wait on each other's collectives in a cycle.
        # Never measured on more than one rank (no multi-GPU node was available): `python bench.py --gpus N` therefore runs
        # under a watchdog that falls back to one stream / one communicator and then to the bucket mode.
        if process_group is not None and self.n_streams > 1 and per_stream:
            # in-line collectives run on the sub-step streams themselves; two of them may be in flight at once, and one
            # communicator must never carry two collectives concurrently: one communicator per sub-step stream (every rank
            # creates them here, in the same order; each stream's collectives are ordered by the stream)
            import torch.distributed as dist
            ranks = list(range(dist.get_world_size(process_group)))
            self.pgs += [dist.new_group(ranks=ranks) for _ in range(self.n_streams - 1)]
        self._cur_pg = process_group
        self._shape = None
        self._graph = None
        self.losses = torch.zeros(len(self.schedule), device=self.dev)
        self._adds = 0
        self._cap_S = self._cap_B = 0

    # ------------------------------------------------------------------------------------------
    def reserve(self, S, B):
        """size every step buffer once for dialogues of up to S utterances and batches of up to B dialogues, so that
        the varying (S, B) of real loaders (no drop_last: every epoch ends on a short batch, train_IEMOCAP.py:62-100)
        never touches the allocator again"""
        self._cap_S, self._cap_B = max(self._cap_S, S), max(self._cap_B, B)

    def _prepare(self, S, B):
        if self._shape == (S, B):
            return
        if self._shape is not None and S <= self._alloc_S and B <= self._alloc_B:
            # within what the buffers were sized for: new views, no allocation, no sync.  (A pass of fewer dialogues uses
            # a prefix of the flat storage; layouts are derived from (S, B) alone.)
            self._resize_passes(S, B)
            self._shape = (S, B)
            self._graph = None
            self.static_batch = None
            self._view_scratch(S, B)
            return
        if self._shape is not None and self.n_streams > 1:
            # the buffers about to be dropped may still be in use by sub-steps queued on the side streams (eager
            # iterations overlap); the caching allocator only tracks the allocating stream
            torch.cuda.synchronize(self.dev)
        # capacity only ever grows (a short last batch must not shrink it)
        cS = self._cap_S = max(self._cap_S, S)
        cB = self._cap_B = max(self._cap_B, B)
        self._alloc_S, self._alloc_B = cS, cB
        self._shape = (S, B)
        self._graph = None
        dev = self.dev
        self.pass_G_nosave = {k: _Pass(n, cS, cB, dev, False) for k, n in self.G.items()}
        self.pass_G = {k: _Pass(n, cS, cB, dev, True) for k, n in self.G.items()}
        self.pass_D2 = {k: _Pass(n, cS, 2 * cB, dev, True) for k, n in self.D.items()}   # [real | fake]
        self.pass_D1 = {k: _Pass(n, cS, cB, dev, True) for k, n in self.D.items()}       # frozen D in train_gen
        n_ws = max(p.n_ws for d in (self.pass_G, self.pass_D2, self.pass_D1, self.pass_G_nosave) for p in d.values())
        f32 = dict(device=dev, dtype=torch.float32)
        Dh = self.D_h
        # scratch is per stream (sub-steps on different streams run concurrently); flat, viewed per (S, B)
        self._scratch_flat = [dict(ws=torch.empty(n_ws, **f32), x_cat=torch.empty(cS * 2 * cB * Dh, **f32),
                                   obj_out=torch.empty(cS * cB * Dh, **f32), dprob2=torch.empty(cS * 2 * cB, **f32),
                                   dprob1=torch.empty(cS * cB, **f32), d_real=torch.empty(cS * cB * Dh, **f32))
                              for _ in range(self.n_streams * (2 if self.early_gen else 1))]
        if (S, B) != (cS, cB):
            self._resize_passes(S, B)
        self._view_scratch(S, B)
        if self.n_streams > 1 and self.streams is None:
            # stream priorities: the visual generator's chain (stream 2 of the 3-stream map: its four sub-steps are a cycle
            # through G_v's parameters and pace the iteration) gets the high priority — measured 35.34 -> 34.90 ms per step
            # (GANFFN_STREAM_PRIO="0,0,0" restores equal priorities; "-1,-1,0" measured 35.6)
            default_prio = "0,0,-1" if (self.n_streams == 3 and len(self.schedule) == 12) else ""
            prio = [int(x) for x in os.environ.get("GANFFN_STREAM_PRIO", default_prio).split(",") if x.strip()]
            prio = (prio + [0] * self.n_streams)[:self.n_streams]
            # main streams, then (early generator forward) one helper stream per main stream
            self.streams = list(_side_streams(dev, prio + (prio if self.early_gen else []), self._tune_streams, S * B))
            self._tune_x, self._tune_pass = (None, None), {}
            self._use_scratch(0)
        self._res = {}
        self.static_batch = None

    def _tune_slot(self, i):
        """the probe work of stream slot i for _tune_streams: eval-mode forwards of one generator into its no-save pass
        buffers with the slot's scratch (nothing else is written: no parameter, gradient or RNG state changes)"""
        k = self.modalities[i % len(self.modalities)]
        if getattr(self, "_tune_x", (None, None))[0] != self._shape:
            S, B = self._shape
            self._tune_x = (self._shape, {m: torch.zeros(S, B, self.G[m].E, device=self.dev) for m in self.modalities})
            self._tune_pass = {}
        # slots beyond the modalities (the early-generator helper streams) probe the same networks: every such slot gets pass
        # buffers of its own, so that concurrent probes never write the same memory
        if i < len(self.modalities):
            ps = self.pass_G_nosave[k]
        else:
            if i not in self._tune_pass:
                self._tune_pass[i] = _Pass(self.G[k], self._shape[0], self._shape[1], self.dev, False)
            ps = self._tune_pass[i]
        self._use_scratch(i)
        for _ in range(1 if self.G[k].E > 256 else 3):                      # (the 512-wide generator is ~3x a 100-wide one)
            self._net_fwd(self.G[k], ps, self._tune_x[1][k], train=False, save=False, adds=(0, 1))

    def _tune_streams(self, prios, n_cand=6, reps=2):
        """choose one stream per entry of `prios` among n_cand fresh candidates per priority by TIMING them on this engine's
        own kernels: slot i runs _tune_slot(i) beside the slots already chosen; the first two slots are chosen jointly over
        all candidate pairs, every further slot greedily.  ~0.15 s, once per process and device."""
        dev = self.dev
        n_cand = int(os.environ.get("GANFFN_STREAM_CAND", n_cand))
        cands = {p_: [torch.cuda.Stream(device=dev, priority=p_) for _ in range(n_cand)] for p_ in sorted(set(prios))}
        cur = torch.cuda.current_stream(dev)

        def probe(streams):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for i, st in enumerate(streams):
                st.wait_event(e0)
                with torch.cuda.stream(st):
                    self._tune_slot(i)
            for st in streams:
                cur.wait_stream(st)
            e1.record(cur)
            e1.synchronize()
            return e0.elapsed_time(e1)

        def best(fixed, pool):
            timed = []
            for c_ in pool:
                group = fixed + (list(c_) if isinstance(c_, tuple) else [c_])
                timed.append((min(probe(group) for _ in range(reps)), c_))
            return min(timed, key=lambda t_: t_[0])[1]

        if self.pg is not None:
            # RCCL creates its internal stream(s) at the first collective of a communicator; a stream that appears AFTER the
            # choice below can land on a hardware queue one of the chosen streams uses.  Issue one tiny all-reduce per
            # communicator first, so that the candidates are timed with RCCL's queue already taken.
            for g_ in getattr(self, "pgs", [self.pg]):
                if g_ is not None:
                    self.dist_warm = torch.zeros(8, device=dev)
                    import torch.distributed as dist
                    dist.all_reduce(self.dist_warm, group=g_, async_op=(dp_mode() != "inline"))
                    if dp_mode() != "inline":
                        torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)
        saved_add = self._base_add
        self._base_add = 0
        try:
            probe([cands[prios[0]][0]])                                      # warm-up (lazy module / allocator state)
            if prios[0] == prios[1]:
                pool = [(a_, b_) for i, a_ in enumerate(cands[prios[0]]) for b_ in cands[prios[0]][i + 1:]]
            else:
                pool = [(a_, b_) for a_ in cands[prios[0]] for b_ in cands[prios[1]]]
            chosen = list(best([], pool))
            for p_ in prios[2:]:
                chosen.append(best(chosen, [c_ for c_ in cands[p_] if c_ not in chosen]))
        finally:
            self._base_add = saved_add
            torch.cuda.synchronize(dev)
        return chosen

    def _resize_passes(self, S, B):
        for d in (self.pass_G_nosave, self.pass_G, self.pass_D1):
            for p_ in d.values():
                p_.resize(S, B)
        for p_ in self.pass_D2.values():
            p_.resize(S, 2 * B)

    def _view_scratch(self, S, B):
        Dh = self.D_h
        self.scratch = [dict(ws=f["ws"], x_cat=f["x_cat"][:S * 2 * B * Dh].view(S, 2 * B, Dh),
                             obj_out=f["obj_out"][:S * B * Dh].view(S, B, Dh),
                             dprob2=f["dprob2"][:S * 2 * B].view(S, 2 * B, 1), dprob1=f["dprob1"][:S * B].view(S, B, 1),
                             d_real=f["d_real"][:S * B * Dh].view(S, B, Dh)) for f in self._scratch_flat]
        self._use_scratch(0)

    def _use_scratch(self, i):
        sc = self.scratch[i]
        self.ws, self.x_cat, self.obj_out = sc["ws"], sc["x_cat"], sc["obj_out"]
        self.dprob2, self.dprob1, self.d_real = sc["dprob2"], sc["dprob1"], sc["d_real"]

    # ---- cross-stream dependencies -------------------------------------------------------------
    def _r(self, key):
        r = self._res.get(key)
        if r is None:
            r = self._res[key] = _Res()
        return r

    def _wait_writers(self, stream, keys):
        for k in keys:
            ev = self._r(k).last_write
            if ev is not None:
                stream.wait_event(ev)

    def _wait_readers(self, stream, keys):
        for k in keys:
            for ev in self._r(k).reads:
                stream.wait_event(ev)

    def _done(self, stream, reads, writes):
        ev = torch.cuda.Event()
        ev.record(stream)
        for k in reads:
            self._r(k).reads.append(ev)
        for k in writes:
            r = self._r(k)
            r.last_write, r.reads = ev, []

    def _next_add(self):
        v = self._adds
        self._adds += 1
        return self._base_add + v

    # ------------------------------------------------------------------------------------------
    def _net_fwd(self, net, ps, x, train, save, adds=None):
        """encoder + head forward into ps.out; returns (enc_add, head_add) rng offsets used.  adds: the two dropout
        offsets (relative to the iteration's block) when the caller assigns them by sub-step; else the next two."""
        cfg = ps.cfg_train if train else ps.cfg_eval
        hcfg = ps.hcfg_train if train else ps.hcfg_eval
        if adds is None:
            a0, a1 = self._next_add(), self._next_add()
        else:
            a0, a1 = self._base_add + adds[0], self._base_add + adds[1]
        ops.encoder_fwd_raw(cfg, x, net.pe, net.slab, ps.enc_out, ps.saved if save else None, self.ws, self.rng.state, a0)
        w = net.w
        w3 = w("fc3.weight") if net.kind == 1 else None
        b3 = w("fc3.bias") if net.kind == 1 else None
        ops.head_fwd_raw(hcfg, ps.enc_out, w("fc1.weight"), w("fc1.bias"), w("fc2.weight"), w("fc2.bias"), w3, b3,
                         ps.out, ps.hsaved, self.ws, self.rng.state, a1)
        return a0, a1

    def _net_bwd(self, net, ps, d_out, train, adds, want_wgrad, reduce_cb=None, need_dx=None, parts=False):
        """head + encoder backward; ps.dx <- dL/d(network input) when `need_dx` (default: exactly when the network is the
        frozen one that only passes gradient through).  Weight grads accumulate into net.grad."""
        cfg = ps.cfg_train if train else ps.cfg_eval
        hcfg = ps.hcfg_train if train else ps.hcfg_eval
        a0, a1 = adds
        w = net.w
        g = (lambda n: net.w(n, grad=True)) if want_wgrad else (lambda n: None)
        w3 = w("fc3.weight") if net.kind == 1 else None
        ops.head_bwd_raw(hcfg, d_out, ps.enc_out, w("fc1.weight"), w("fc2.weight"), w3,
                         g("fc1.weight"), g("fc1.bias"), g("fc2.weight"), g("fc2.bias"),
                         g("fc3.weight") if net.kind == 1 else None, g("fc3.bias") if net.kind == 1 else None,
                         ps.dx, ps.hsaved, self.ws, self.rng.state, a1)
        gslab = net.grad if want_wgrad else None
        # the network being trained reads a raw modality or detached fakes: nobody consumes dL/d(its input)
        # (train_IEMOCAP.py:200-252) — except the visual discriminator's `object` layer; the frozen discriminator of
        # train_gen passes its input gradient on to the generator
        if need_dx is None:
            need_dx = not want_wgrad
        if parts and want_wgrad and reduce_cb is None:
            # single GPU, d_model 100: the weight gradients stay as token-chunk slabs for Adam to add (no reduce launch, and the
            # caller did not zero the encoder region of net.grad) -> (parts view of self.ws, stride, chunks)
            return ops.encoder_bwd_parts_raw(cfg, ps.dx, net.slab, gslab, ps.saved, self.ws, self.rng.state, a0, need_dx)
        if reduce_cb is None or not want_wgrad:
            ops.encoder_bwd_raw(cfg, 0, net.L, ps.dx, net.slab, gslab, ps.saved, self.ws, self.rng.state, a0, need_dx)
        else:
            # bucketed: backward a group of layers, then hand that slice of the grad slab to the all-reduce
            bks = net.buckets(self.n_buckets)
            reduce_cb(*bks[0], last=False)                       # head (+object handled by caller before this)
            for i, (lo_f, hi_f) in enumerate(bks[1:]):
                lo, hi = lo_f // net.layer_floats, hi_f // net.layer_floats
                ops.encoder_bwd_raw(cfg, lo, hi, ps.dx, net.slab, gslab, ps.saved, self.ws, self.rng.state, a0, need_dx)
                reduce_cb(lo_f, hi_f, last=(i == len(bks) - 2))

    def _adam(self, net, parts=None):
        if parts is not None and parts[2] > 1:
            ops.adam_step_parts_raw(net.slab, net.grad, net.exp_avg, net.exp_avg_sq, net.step, net.total, net.lr, net.betas[0],
                                    net.betas[1], parts[0], parts[1], parts[2], net.enc_floats, net.layer_floats, net.covered,
                                    1e-8, net.wd, 1.0 / self.world)
            return
        ops.adam_step_raw(net.slab, net.grad, net.exp_avg, net.exp_avg_sq, net.step, net.total, net.lr,
                          net.betas[0], net.betas[1], 1e-8, net.wd, 1.0 / self.world)

    def _parts_ok(self, net, ps):
        """single GPU, d_model 100: leave the weight gradients of this backward pass as token-chunk slabs and let Adam add them
        (ganffn_encoder_bwd_parts / ganffn_adam_step_parts): no reduce launch, no zero-fill of the encoder region of net.grad.
        With a process group the all-reduce needs the summed slab: today's path.  GANFFN_ADAM_PARTS=0 switches it off."""
        return self.pg is None and os.environ.get("GANFFN_ADAM_PARTS", "1") == "1" and net.total % 4 == 0 and \
            ops.encoder_bwd_parts_supported(ps.cfg_train)

    def _zero_grad(self, net, parts):
        """opt.zero_grad() (train_IEMOCAP.py:216,245); with unreduced weight gradients only what still accumulates: heads, `object`"""
        if parts:
            net.grad[net.enc_floats:].zero_()
        else:
            net.grad.zero_()

    def _adam_slice(self, net, lo, hi):
        ops.adam_update_raw(net.slab[lo:hi], net.grad[lo:hi], net.exp_avg[lo:hi], net.exp_avg_sq[lo:hi], net.step, hi - lo,
                            net.lr, net.betas[0], net.betas[1], 1e-8, net.wd, 1.0 / self.world)

    def _make_reducer(self, net):
        """returns (callback, finish_and_step): async all-reduce (sum) of grad-slab slices on RCCL's own stream,
        overlapping the rest of backward.  finish_and_step(net_key) waits bucket by bucket and applies Adam (which
        divides by world: grad_scale) to each slice as soon as its reduce is done; without a process group it is the
        plain whole-slab Adam step."""
        if self.pg is None:
            def plain(net_key, parts=None):
                self._pre_write(net_key)
                self._adam(net, parts)
            return None, plain
        if dp_mode() == "inline":
            import torch.distributed as dist
            group = getattr(self, "_cur_pg", None) or self.pg

            def inline(net_key, parts=None):
                # on the CURRENT stream (the sub-step's own): sum over ranks, then Adam divides by world (grad_scale)
                dist.all_reduce(net.grad, op=dist.ReduceOp.SUM, group=group, async_op=False)
                self._pre_write(net_key)
                self._adam(net)
            return None, inline
        red = GradReducer(getattr(self, "_cur_pg", None) or self.pg)

        def cb(lo, hi, last):
            red.reduce_async(net.grad[lo:hi], lo, hi)

        def finish_and_step(net_key, parts=None):
            self._pre_write(net_key)            # other streams' readers of these parameters first (WAR)
            red.finish(lambda lo, hi: self._adam_slice(net, lo, hi) if hi > lo else None)
            ops.adam_bump_raw(net.step)
        return cb, finish_and_step

    # ------------------------------------------------------------------------------------------
    def train_disc(self, who, partner, batch, loss_slot):
        """train_IEMOCAP.py:200-227."""
        S, B = batch[who].shape[:2]
        Dn, Gn = self.D[who], self.G[partner]
        pg_, pd = self.pass_G_nosave[partner], self.pass_D2[who]
        # dropout offsets are assigned by sub-step (4 per sub-step: generator encoder / head, discriminator encoder / head), so
        # they do not depend on the order in which streams issue their launches
        a = ADDS_PER_SUBSTEP * loss_slot
        # fusion = G(real_gen) in eval mode, nothing saved (detach(), :218-219)
        self._net_fwd(Gn, pg_, batch[partner], train=False, save=False, adds=(a, a + 1))
        # real input of D_m is raw modality m; VisualDiscriminator maps 512 -> 100 first (model.py:1355-1356)
        x_real = batch[who]
        if Dn.has_obj:
            ops.linear_fwd_raw(x_real, Dn.w("object.weight"), Dn.w("object.bias"), self.obj_out, S * B, Dn.obj_in, self.D_h)
            x_real = self.obj_out
        torch.cat((x_real, pg_.out), dim=1, out=self.x_cat)
        adds = self._net_fwd(Dn, pd, self.x_cat, train=True, save=True, adds=(a + 2, a + 3))
        n = S * 2 * B
        ops._lib.call("ganffn_bce2_fwd", ops._ptr(pd.out), C.c_float(1.0), C.c_float(0.0), 2 * B, B, n, C.c_float(1.0),
                      ops._ptr(self.losses[loss_slot:loss_slot + 1]), 0, ops._stream())
        ops._lib.call("ganffn_bce2_bwd", ops._ptr(pd.out), C.c_float(1.0), C.c_float(0.0), 2 * B, B, n, C.c_float(1.0),
                      ops._ptr(self.dprob2), ops._stream())
        parts_mode = self._parts_ok(Dn, pd)
        self._zero_grad(Dn, parts_mode)                              # opt.zero_grad(), :216
        cb, finish = self._make_reducer(Dn)
        parts = self._net_bwd(Dn, pd, self.dprob2, True, adds, True, cb, need_dx=Dn.has_obj, parts=parts_mode)
        if Dn.has_obj:
            self.d_real.copy_(pd.dx[:, :B])                      # gradient of the real half of the batch
            # (scratch: the workspace BELOW the unreduced weight-gradient chunks the Adam launch is about to read)
            scratch = self.ws[:parts[3]] if (parts is not None and parts[2] > 1) else self.ws
            ops.linear_bwd_raw(self.d_real, batch[who], Dn.w("object.weight"), None, Dn.w("object.weight", True),
                               Dn.w("object.bias", True), S * B, Dn.obj_in, self.D_h, scratch)
            if cb is not None:
                cb(Dn.enc_floats, Dn.enc_floats + Dn.obj_floats, last=True)
        finish(("D", who), parts)

    def train_gen_forward(self, who, batch, loss_slot):
        """the generator's own forward of train_gen (train mode, saved for backward): it reads nothing but the generator's
        parameters and the batch, so the multi-stream scheduler may issue it ahead of the sub-step"""
        a = ADDS_PER_SUBSTEP * loss_slot
        return self._net_fwd(self.G[who], self.pass_G[who], batch[who], train=True, save=True, adds=(a, a + 1))

    def train_gen(self, who, partner, batch, loss_slot, g_adds=None):
        """train_IEMOCAP.py:230-252.  g_adds: the generator forward has already been issued (train_gen_forward)."""
        S, B = batch[who].shape[:2]
        Gn, Dn = self.G[who], self.D[partner]
        pg_, pd = self.pass_G[who], self.pass_D1[partner]
        a = ADDS_PER_SUBSTEP * loss_slot
        if g_adds is None:
            g_adds = self.train_gen_forward(who, batch, loss_slot)
        d_adds = self._net_fwd(Dn, pd, pg_.out, train=False, save=True, adds=(a + 2, a + 3))       # disc.eval(), :243
        n = S * B
        ops.bce_fwd_raw(pd.out, 1.0, n, 1.0, self.losses[loss_slot:loss_slot + 1], False)
        ops.bce_bwd_raw(pd.out, 1.0, n, 1.0, self.dprob1)
        self._net_bwd(Dn, pd, self.dprob1, False, d_adds, False)             # through the frozen D: dgrad only
        parts_mode = self._parts_ok(Gn, pg_)
        self._zero_grad(Gn, parts_mode)
        cb, finish = self._make_reducer(Gn)
        parts = self._net_bwd(Gn, pg_, pd.dx, True, g_adds, True, cb, parts=parts_mode)
        finish(("G", who), parts)

    def _pre_write(self, net_key):
        """called right before a sub-step's Adam: wait for other streams' readers of that network (WAR)"""
        if self.n_streams > 1 and self._cur_stream is not None:
            self._wait_readers(self._cur_stream, [net_key])

    # ------------------------------------------------------------------------------------------
    _cur_stream = None

    def _iteration_body(self, batch, device_rng_advance):
        self._adds = 0
        self._check_slabs()
        if not device_rng_advance:
            # eager: this iteration's block of dropout offsets comes from the device's one allocator (shared with the
            # module path and every other engine); iterations may overlap, the offsets are host-side arguments
            self._base_add = self.rng.next_add(ADDS_PER_SUBSTEP * len(self.schedule))
        self._adds = ADDS_PER_SUBSTEP * len(self.schedule)       # (offsets are assigned by sub-step: train_disc / train_gen)
        if self.n_streams == 1:
            for i, (kind, who, partner) in enumerate(self.schedule):
                (self.train_disc if kind == "D" else self.train_gen)(who, partner, batch, i)
        else:
            origin = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(origin)
            smap = self.stream_map
            for st in self.streams:
                st.wait_event(fork)
                # the caller may drop this batch as soon as we return while the side streams still read it: tell the
                # caching allocator, so the memory is not handed out again before those streams are done with it
                for k in self.modalities:
                    if batch[k].is_cuda:
                        batch[k].record_stream(st)
            early = {}
            nsub = len(self.schedule)
            for i, (kind, who, partner) in enumerate(self.schedule):
                st = self.streams[smap[i]]
                # Early generator forward: the train-mode forward of the NEXT sub-step's generator reads only that generator's
                # parameters (which this sub-step does not write) and the batch, so it is issued now, on this stream's helper
                # stream, and runs beside this sub-step instead of after it.  (The visual generator's four sub-steps are a
                # cycle — each needs the parameters the previous one wrote: its two train-mode forwards leave that chain.)
                # Results do not change: same parameters, same dropout offsets (the multi-stream tests pass bit for bit with
                # it on).  NO GAIN, therefore OFF by default (GANFFN_EARLY_GEN=1 enables it): 35.4 against 35.3 ms per step with
                # its six streams chosen by _tune_streams (37.8 and 47 ms before that: helper streams that shared a hardware
                # queue with a main stream).
                j = i + 1
                if self.early_gen and j < nsub and self.schedule[j][0] == "G" and smap[j] == smap[i] and \
                        (kind, who) != ("G", self.schedule[j][1]):
                    gwho = self.schedule[j][1]
                    hs = self.streams[self.n_streams + smap[j]]
                    self._use_scratch(self.n_streams + smap[j])
                    with torch.cuda.stream(hs):
                        self._wait_writers(hs, [("G", gwho), ("buf", "G", gwho)])
                        self._wait_readers(hs, [("buf", "G", gwho)])
                        early[j] = self.train_gen_forward(gwho, batch, j)
                        ev = torch.cuda.Event()
                        ev.record(hs)
                        self._r(("G", gwho)).reads.append(ev)        # a later writer of these parameters waits for this read
                        rb = self._r(("buf", "G", gwho))
                        rb.last_write, rb.reads = ev, []             # (the sub-step itself waits for it as the buffer's writer)
                trained = (kind, who)
                other = ("G" if kind == "D" else "D", partner)
                # pass buffers are resources too (same (net, role) buffer reused by a later sub-step)
                if kind == "D":
                    bufs = [("buf", "G_nosave", partner), ("buf", "D2", who)]
                else:
                    bufs = [("buf", "G", who), ("buf", "D1", partner)]
                self._use_scratch(smap[i])
                self._cur_stream = st
                self._cur_pg = self.pgs[smap[i]] if len(self.pgs) > 1 else self.pg
                with torch.cuda.stream(st):
                    self._wait_writers(st, [trained, other] + bufs)
                    self._wait_readers(st, bufs)
                    if kind == "D":
                        self.train_disc(who, partner, batch, i)
                    else:
                        self.train_gen(who, partner, batch, i, g_adds=early.get(i))
                    self._done(st, reads=[other], writes=[trained] + bufs)
                self._cur_stream = None
            self._cur_pg = self.pg
            self._use_scratch(0)
            if device_rng_advance:
                # graph mode: join everything (the device-side offset bump below must follow every kernel)
                for st in self.streams:
                    origin.wait_stream(st)
                self._res = {}
        if device_rng_advance:
            ops.rng_advance_raw(self.rng.state, self._adds)
        else:
            assert self._adds <= ADDS_PER_SUBSTEP * len(self.schedule), self._adds

    def synchronize(self):
        """make the current stream wait for all side-stream work (call before reading results / timing)"""
        if self.n_streams > 1 and self.streams is not None:
            cur = torch.cuda.current_stream()
            for st in self.streams:
                cur.wait_stream(st)

    def iteration(self, batch):
        """One batch = 12 sub-steps.  Returns the device tensor of the 12 sub-step losses (no host sync).
        With n_streams > 1 call synchronize() (or loss_dict()) before reading it on the current stream."""
        S, B = batch[self.modalities[0]].shape[:2]
        self._prepare(S, B)
        if not self.use_graph:
            self._iteration_body(batch, device_rng_advance=False)
            return self.losses
        if self.static_batch is None:
            self.static_batch = {k: batch[k].clone() for k in self.modalities}
        else:
            for k in self.static_batch:
                self.static_batch[k].copy_(batch[k])
        if self._graph is None:
            # warm up on a side stream (allocations, lazy module init), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._iteration_body(self.static_batch, device_rng_advance=True)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._iteration_body(self.static_batch, device_rng_advance=True)
        self._graph.replay()
        return self.losses

    def loss_dict(self):
        """host copy of the last iteration's losses under the reference's column names (last value per key,
        as train_IEMOCAP.py:355-382 keeps)."""
        self.synchronize()
        v = self.losses.tolist()
        out = {}
        for (kind, who, _), x in zip(self.schedule, v):
            out["%s_%s_loss" % (who, kind)] = x
        return out


def build_networks(D_h=100, dropout=0.2, device="cuda", seed=None):
    """the six networks as train_IEMOCAP.py:580-585 builds them"""
    from . import model
    if seed is not None:
        torch.manual_seed(seed)
    discs = {"acoustic": model.AcousticDiscriminator(D_h, dropout=dropout),
             "visual": model.VisualDiscriminator(D_h, dropout=dropout),
             "text": model.TextDiscriminator(D_h, dropout=dropout)}
    gens = {"acoustic": model.AcousticGenerator(D_h, dropout=dropout),
            "visual": model.VisualGenerator(D_h, dropout=dropout),
            "text": model.TextGenerator(D_h, dropout=dropout)}
    for d in (gens, discs):
        for k in d:
            d[k] = d[k].to(device)
    return gens, discs


def train_GAN(gens, discs, batches, epochs=1, lr=1e-4, b1=0.5, b2=0.6, process_group=None, use_graph=False, log=None,
              n_streams=3, reserve_S=None):
    """Counterpart of train_GAN (train_IEMOCAP.py:255-393) over an iterable of batches per epoch.
    Returns rows of the GAN_loss table (columns train_IEMOCAP.py:308-316): last batch of each epoch.
    reserve_S: size the step buffers once for dialogues up to this length (PositionalEncoding allows 110), so that
    batches of varying length never re-allocate."""
    eng = GanEngine(gens, discs, lr, b1, b2, process_group, use_graph=use_graph, n_streams=n_streams)
    rows = []
    for epoch in range(epochs):
        last = None
        for batch in batches:
            if reserve_S and eng._shape is None:
                S0, B0 = batch["text"].shape[:2]
                eng.reserve(max(reserve_S, S0), B0)
            eng.iteration(batch)
            last = eng.loss_dict()
            if log:
                log(epoch, last)
        if last is not None:
            rows.append(dict(epoch=epoch, **{c: last[c] for c in LOSS_COLUMNS}))
    return rows


# ================================================================================================
# Phase 2: GAN_FFN classifier step      (/root/reference/model.py:1434-1462, train_IEMOCAP.py:103-197,653-661)
# ================================================================================================
CLASS_WEIGHTS = [1.2, 0.60072, 0.38066, 0.94019, 0.67924, 0.34332]   # train_IEMOCAP.py:653


class Phase2Engine(GanEngine):
    """One step of train_or_eval_model on GAN_FFN: log_softmax(fc(G_a(a) + G_v(v) + G_t(t))), MaskedNLLLoss with
    class weights, backward through the three generators, Adam(lr, weight_decay=l2) on everything.
    The reference re-creates a LambdaLR every batch, which pins the effective lr to its base value."""

    def __init__(self, ffn_module, lr=1e-4, weight_decay=0.008, class_weights=CLASS_WEIGHTS, process_group=None,
                 n_buckets=3):
        gens = {"acoustic": ffn_module.acoustic_generator, "visual": ffn_module.visual_generator,
                "text": ffn_module.text_generator}
        self.module = ffn_module
        self.G = {k: NetState(m, lr, (0.9, 0.999), weight_decay) for k, m in gens.items()}
        self.D = {}
        self._init_common(next(iter(self.G.values())).slab.device, process_group, n_buckets)
        self.n_streams, self.use_graph = 1, False
        self.n_classes = ffn_module.fc.weight.shape[0]
        dev = self.dev
        # fc (100 -> n_classes) parameters as one small slab [weight | bias], 16-byte aligned pieces
        self.fc_w, self.fc_b = ffn_module.fc.weight, ffn_module.fc.bias
        nw = self.fc_w.numel()
        self.fc_off_b = (nw + 3) & ~3
        self.fc_total = self.fc_off_b + ((self.fc_b.numel() + 3) & ~3)
        self.fc_slab = torch.zeros(self.fc_total, device=dev)
        with torch.no_grad():
            self.fc_slab[:nw].copy_(self.fc_w.detach().reshape(-1))
            self.fc_slab[self.fc_off_b:self.fc_off_b + self.fc_b.numel()].copy_(self.fc_b.detach())
            self.fc_w.data = self.fc_slab[:nw].view_as(self.fc_w)
            self.fc_b.data = self.fc_slab[self.fc_off_b:self.fc_off_b + self.fc_b.numel()]
        self.fc_grad = torch.zeros_like(self.fc_slab)
        self.fc_m, self.fc_v = torch.zeros_like(self.fc_slab), torch.zeros_like(self.fc_slab)
        self.fc_step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr, self.wd = lr, weight_decay
        self.class_w = torch.tensor(class_weights, device=dev, dtype=torch.float32) if class_weights is not None else None
        self._shape = None
        self.loss = torch.zeros(1, device=dev)
        self._adds = 0
        self._base_add = 0

    def reserve(self, S, B):
        """size the pass buffers once for batches of up to (S, B): train / valid / test loaders then never re-allocate"""
        self._cap_S, self._cap_B = max(getattr(self, "_cap_S", 0), S), max(getattr(self, "_cap_B", 0), B)

    def _prepare2(self, S, B):
        if self._shape == (S, B):
            return
        C_ = self.n_classes
        if self._shape is None or S > self._alloc_S or B > self._alloc_B:
            cS = self._cap_S = max(getattr(self, "_cap_S", 0), S)
            cB = self._cap_B = max(getattr(self, "_cap_B", 0), B)
            self._alloc_S, self._alloc_B = cS, cB
            dev = self.dev
            self.pass_G = {k: _Pass(n, cS, cB, dev, True) for k, n in self.G.items()}
            n_ws = max(p.n_ws for p in self.pass_G.values())
            f32 = dict(device=dev, dtype=torch.float32)
            self.ws = torch.empty(n_ws, **f32)
            self._flat = dict(fusion=torch.empty(cS * cB * 100, **f32), logits=torch.empty(cS * cB * C_, **f32),
                              log_prob=torch.empty(cS * cB * C_, **f32), dlogits=torch.empty(cS * cB * C_, **f32),
                              d_fusion=torch.empty(cS * cB * 100, **f32))
            self.ws2 = torch.zeros(4, **f32)
        self._shape = (S, B)
        for p_ in self.pass_G.values():
            p_.resize(S, B)
        f = self._flat
        self.fusion = f["fusion"][:S * B * 100].view(S, B, 100)
        self.logits = f["logits"][:S * B * C_].view(S, B, C_)
        self.log_prob = f["log_prob"][:S * B * C_].view(S, B, C_)
        self.dlogits = f["dlogits"][:S * B * C_].view(S, B, C_)
        self.d_fusion = f["d_fusion"][:S * B * 100].view(S, B, 100)

    def step(self, batch, train=True):
        """batch: text/visual/acoustic (S,B,.), umask (B,S) float, label (B,S) int64.  Returns (loss tensor, log_prob).
        train=False: forward + loss only (model.eval())."""
        S, B = batch["text"].shape[:2]
        self._prepare2(S, B)
        self._check_slabs()
        if self.fc_w.data_ptr() != self.fc_slab.data_ptr():
            raise RuntimeError("GAN_FFN.fc was re-allocated after the engine was built: build Phase2Engine after the last .to()")
        T, C_ = S * B, self.n_classes
        self._adds = 0
        self._base_add = self.rng.next_add(8)      # 3 generators x (encoder, head): one block from the device allocator
        adds = {}
        for k in ("acoustic", "visual", "text"):                    # model.py:1441-1443
            adds[k] = self._net_fwd(self.G[k], self.pass_G[k], batch[k], train=train, save=train)
        ops._lib.call("ganffn_add3", ops._ptr(self.pass_G["acoustic"].out), ops._ptr(self.pass_G["visual"].out),
                      ops._ptr(self.pass_G["text"].out), ops._ptr(self.fusion), C.c_int64(T * 100), ops._stream())
        ops.linear_fwd_raw(self.fusion, self.fc_w, self.fc_b, self.logits, T, 100, C_)          # model.py:1448
        ops.logsoftmax_nll_raw(self.logits, batch["label"], batch["umask"], self.class_w, self.log_prob, self.loss,
                               self.dlogits if train else None, self.ws2, S, B, C_)                # model.py:1449, :74-81
        if train:
            self.fc_grad.zero_()
            ops.linear_bwd_raw(self.dlogits, self.fusion, self.fc_w, self.d_fusion, self.fc_grad[:self.fc_w.numel()],
                               self.fc_grad[self.fc_off_b:], T, 100, C_, self.ws)
            for k in ("acoustic", "visual", "text"):
                net = self.G[k]
                net.grad.zero_()
                cb, finish = self._make_reducer(net)
                self._net_bwd(net, self.pass_G[k], self.d_fusion, True, adds[k], True, cb)
                finish(("G", k))
            if self.pg is not None:
                if dp_mode() == "inline":
                    import torch.distributed as dist
                    dist.all_reduce(self.fc_grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)
                else:
                    red = GradReducer(self.pg)
                    red.reduce_async(self.fc_grad)
                    red.finish()
            ops.adam_step_raw(self.fc_slab, self.fc_grad, self.fc_m, self.fc_v, self.fc_step, self.fc_total, self.lr,
                              0.9, 0.999, 1e-8, self.wd, 1.0 / self.world)
        assert self._adds <= 8, self._adds
        return self.loss, self.log_prob

    @staticmethod
    def predictions(log_prob):
        """argmax over classes in the reference's batch-major flattening (train_IEMOCAP.py:154,158)"""
        return log_prob.transpose(0, 1).reshape(-1, log_prob.shape[2]).argmax(1)


# ================================================================================================
# Configuration 5: GAN_FFN_DialogueRNN classifier step     (/root/reference/train_IEMOCAP_DialogueRNN.py:705-760,
#                                                          model.py:975-1062, 1465-1528)
# ================================================================================================
SITE_JOIN_F, SITE_JOIN_B, SITE_HIDDEN = 5, 6, 7       # dropout sites of the head (the recurrence uses 8..14, the encoders 16+)


class DrnnEngine(GanEngine):
    """One train / eval step of GAN_FFN_DialogueRNN on the C ABI, no autograd graph: the three generators (n_streams = 3
    runs their forward and backward passes on three HIP streams chosen like GanEngine's; measured NOT faster than one
    stream on this workload — 15.0-16.5 against 14.9-15.0 ms per step: kernels of different streams do not run side by
    side, only the dead time between launches overlaps, and the run-to-run spread grows —
    so one stream is the default), fusion = their sum, BiModel's two
    DialogueRNN directions through one chain of launches (ganffn_drnn_fwd / _bwd), the matching attention
    (ganffn_general2_attention_*), linear + ReLU + dropout, the class head, MaskedNLLLoss with class weights, and Adam
    (lr, L2-coupled weight decay; train_IEMOCAP_DialogueRNN.py:746) on flat slabs: one fused launch per generator and one
    for the whole head.  Data-parallel: the generators' gradients go through the bucketed GradReducer (all-reduce of a
    bucket overlaps the rest of that generator's backward, Adam per bucket), the head's slab is one more bucket.
    Supports the configuration the reference script trains (general context attention, no listener, two parties); the
    module path (model.GAN_FFN_DialogueRNN.forward under autograd) stays available for everything else."""

    def __init__(self, net, lr=1e-4, weight_decay=1e-5, class_weights=CLASS_WEIGHTS, process_group=None, n_buckets=3,
                 n_streams=1):
        from . import dialogue_rnn as DR
        self.module = net
        bm = net.bi_model
        cf, cr = bm.dialog_rnn_f.dialogue_cell, bm.dialog_rnn_r.dialogue_cell
        if cf.listener_state or getattr(cf.attention, "att_type", None) != "general" or cf.D_g != cf.D_p or cf.D_g > 512:
            raise ValueError("DrnnEngine runs the trained configuration only (general context attention, no listener, "
                             "D_g = D_p <= 512); use the module path for the other variants")
        gens = {"acoustic": net.acoustic_generator, "visual": net.visual_generator, "text": net.text_generator}
        self.G = {k: NetState(m, lr, (0.9, 0.999), weight_decay) for k, m in gens.items()}
        self.D = {}
        self._init_common(next(iter(self.G.values())).slab.device, process_group, n_buckets)
        dev = self.dev
        self.lr, self.wd = lr, weight_decay
        self.Dm, self.H, self.He, self.Dh2 = cf.D_m, cf.D_g, cf.D_e, bm.linear.weight.shape[0]
        self.n_classes = bm.smax_fc.weight.shape[0]
        self.p_rec, self.p_join, self.p_hid = float(cf.dropout.p), float(bm.dropout_rec.p), float(bm.dropout.p)
        # ---- head parameters -> one slab (views keep the module's parameters alive on it)
        plist = []
        for cell in (cf, cr):
            sd = dict(cell.named_parameters())
            plist += [sd[k] for k in ops.DRNN_KEYS]
        plist += [bm.matchatt.transform.weight, bm.matchatt.transform.bias, bm.linear.weight, bm.linear.bias,
                  bm.smax_fc.weight, bm.smax_fc.bias]
        offs, total = [], 0
        for p_ in plist:
            offs.append(total)
            total += (p_.numel() + 3) & ~3
        self.h_slab = torch.zeros(total, device=dev)
        with torch.no_grad():
            for p_, o in zip(plist, offs):
                self.h_slab[o:o + p_.numel()].copy_(p_.detach().reshape(-1))
                p_.data = self.h_slab[o:o + p_.numel()].view_as(p_)
        self._hparams, self._hoffs, self.h_total = plist, offs, total
        self.h_grad = torch.zeros_like(self.h_slab)
        self.h_m, self.h_v = torch.zeros_like(self.h_slab), torch.zeros_like(self.h_slab)
        self.h_step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.class_w = torch.tensor(class_weights, device=dev, dtype=torch.float32) if class_weights is not None else None
        self.n_streams = max(1, min(3, n_streams))
        if process_group is not None:
            self.n_streams = 1           # (one communicator: its in-line collectives must not run on several streams at once)
        self.streams = None                                  # (n_streams > 1: chosen in _prepare5, once the buffers exist)
        self.loss = torch.zeros(1, device=dev)
        self._shape = None
        self._cap_S = self._cap_B = 0
        self._adds = 0
        self._base_add = 0

    def _hp(self, i, grad=False):
        o, n = self._hoffs[i], self._hparams[i].numel()
        return (self.h_grad if grad else self.h_slab)[o:o + n]

    def reserve(self, S, B):
        self._cap_S, self._cap_B = max(self._cap_S, S), max(self._cap_B, B)

    def _prepare5(self, S, B):
        if self._shape == (S, B):
            return
        if B > 32 or S > 112:
            raise ValueError("DrnnEngine: at most 32 dialogues of at most 112 utterances per step (the recurrence's tile and the "
                             "attention kernels' sequence limit); got S = %d, B = %d — split the batch, or run the module path "
                             "(model.GAN_FFN_DialogueRNN under autograd, which chunks by itself)" % (S, B))
        if self._shape is None or S > self._alloc_S or B > self._alloc_B:
            cS = self._cap_S = max(self._cap_S, S)
            cB = self._cap_B = max(self._cap_B, B)
            self._alloc_S, self._alloc_B = cS, cB
            dev = self.dev
            self.pass_G = {k: _Pass(n, cS, cB, dev, True) for k, n in self.G.items()}
            f32 = dict(device=dev, dtype=torch.float32)
            self.ws3 = {k: torch.empty(p_.n_ws, **f32) for k, p_ in self.pass_G.items()}     # one workspace per generator stream
            cfgc = _lib.DrnnCfg(cS, cB, self.Dm, self.H, self.He, self.p_rec, 1)
            lib = _lib.load()
            n_saved, n_ws = int(lib.ganffn_drnn_saved_floats(C.byref(cfgc))), int(lib.ganffn_drnn_workspace_floats(C.byref(cfgc)))
            if n_saved < 0 or n_ws < 0:
                _lib.check(-1, "ganffn_drnn_*_floats")
            T, D2, Cn = cS * cB, 2 * self.He, self.n_classes
            z = lambda n: torch.empty(n, **f32)
            self._f = dict(fusion=z(T * self.Dm), rev_U=z(T * self.Dm), e_f=z(T * self.He), e_b=z(T * self.He),
                           alpha_f=z(cB * cS * cS), alpha_b=z(cB * cS * cS), emotions=z(T * D2), xq=z(T * D2), att=z(T * D2),
                           alpha2=z(cB * cS * cS), tanh_s=z(cB * cS * cS), du=z(cB * cS * cS), hidden=z(T * self.Dh2),
                           logits=z(T * Cn), log_prob=z(T * Cn), dlogits=z(T * Cn), d_hidden=z(T * self.Dh2), d_att=z(T * D2),
                           d_xq=z(T * D2), d_mem=z(T * D2), d_em=z(T * D2), d_e_f=z(T * self.He), d_e_b=z(T * self.He),
                           dU_f=z(T * self.Dm), dU_b=z(T * self.Dm), saved_f=z(n_saved), saved_b=z(n_saved), ws_f=z(n_ws),
                           ws_b=z(n_ws), lin_ws=z(int(lib.ganffn_linear_bwd_workspace_floats(T, D2, D2)) + 64))
            self.ws2 = torch.zeros(4, **f32)
        self._shape = (S, B)
        for p_ in self.pass_G.values():
            p_.resize(S, B)
        self.cfg_train = _lib.DrnnCfg(S, B, self.Dm, self.H, self.He, self.p_rec, 1)
        self.cfg_eval = _lib.DrnnCfg(S, B, self.Dm, self.H, self.He, self.p_rec, 0)
        if self.n_streams > 1 and self.streams is None:
            self.streams = list(_side_streams(self.dev, [0, 0, 0], self._tune_streams, S * B))
            self._tune_x = (None, None)

    def _tune_slot(self, i):
        k = ("acoustic", "visual", "text")[i % 3]
        if getattr(self, "_tune_x", (None, None))[0] != self._shape:
            S, B = self._shape
            self._tune_x = (self._shape, {m: torch.zeros(S, B, self.G[m].E, device=self.dev) for m in self.G})
        self.ws = self.ws3[k]
        # (save=True: these pass buffers and their workspace were sized for the saving mode; the next real forward overwrites)
        self._net_fwd(self.G[k], self.pass_G[k], self._tune_x[1][k], train=False, save=True, adds=(0, 1))

    # ------------------------------------------------------------------------------------------
    def _drnn_ptrs(self, grad):
        out = []
        for z in range(2):
            s = _lib.DrnnPtrs()
            for j, name in enumerate(_lib.DRNN_PARAM_FIELDS):
                setattr(s, name, self._hp(13 * z + j, grad).data_ptr())
            out.append(s)
        return (_lib.DrnnPtrs * 2)(*out)

    def step(self, batch, train=True):
        """batch: acoustic/visual/text (S,B,.), qmask (S,B,2) one-hot (zero rows on padding), umask (B,S), label (B,S) int64.
        Returns (loss tensor, log_prob (S,B,C)).  train=False: forward + loss only (model.eval())."""
        S, B = batch["text"].shape[:2]
        self._prepare5(S, B)
        if self.streams is None:
            return self._step(batch, train)
        # n_streams = 3: the WHOLE step runs on the tuned streams — the visual generator's stream carries the recurrence and
        # the head as well, the two 100-wide generators run beside it — and the caller's stream only waits at both ends
        # (round 3 ran the recurrence on the caller's stream, a fourth stream the tuner had not chosen).  Section timing
        # (tools/lab/drnn_sections.py, gpurun_out/r4_drnn_sections*.txt): the three generators' forward 3.75 -> 3.3 ms and
        # backward + Adam 5.69 -> 4.9 ms on three streams, but the recurrence + head — a chain of ~400 latency-sized launches —
        # 5.84 -> 6.3-7.3 ms as soon as the process has several active hardware queues, whichever stream it runs on:
        # 14.45-15.6 ms against 14.8-15.3 on one stream depending on the box.  One stream stays the default.
        cur = torch.cuda.current_stream()
        main = self.streams[1]
        main.wait_stream(cur)
        for t_ in batch.values():
            if torch.is_tensor(t_) and t_.is_cuda:
                t_.record_stream(main)
        with torch.cuda.stream(main):
            out = self._step(batch, train)
        cur.wait_stream(main)
        return out

    def _step(self, batch, train=True):
        S, B = batch["text"].shape[:2]
        self._check_slabs()
        if self._hparams[0].data_ptr() != self.h_slab.data_ptr():
            raise RuntimeError("the DialogueRNN head was re-allocated after the engine was built: build DrnnEngine after the last .to()")
        P, st_ = ops._ptr, ops._stream
        T, Dm, He, D2, Cn = S * B, self.Dm, self.He, 2 * self.He, self.n_classes
        f = self._f
        self._adds = 0
        self._base_add = self.rng.next_add(10)          # 3 generators x (encoder, head), the recurrence, the head's dropouts
        a_rec, a_head = self._base_add + 6, self._base_add + 7
        rng = self.rng.state
        umask, qmask = batch["umask"], batch["qmask"]
        if ops._CHECK_QMASK:
            # GANFFN_CHECK_QMASK=1 (one host sync per batch): the gate kernels use argmax / max of a qmask row and the lengths
            # umask.sum(1) — a soft or multi-hot speaker row or a mask with holes would silently differ from the reference
            row = qmask.sum(2)
            if not bool((((row == 1) & (qmask.max(2).values == 1)) | (row == 0)).all()):
                raise ValueError("DrnnEngine: qmask rows must be one-hot (or all zero on padding)")
            L_ = umask.sum(1).long()
            if not bool((umask == (torch.arange(S, device=umask.device).unsqueeze(0) < L_.unsqueeze(1)).to(umask.dtype)).all()):
                raise ValueError("DrnnEngine: umask rows must be prefixes (1 .. 1 0 .. 0)")
        # per-batch index data (tiny): dialogue lengths, speaker index / value per step, in both directions
        lens = umask.sum(1).to(torch.int32)
        spk_f = torch.argmax(qmask, 2).to(torch.int32).contiguous()
        mval_f = qmask.max(2).values.contiguous()
        t_idx = torch.arange(S, device=self.dev).unsqueeze(1)
        src = (lens.to(torch.long).unsqueeze(0) - 1 - t_idx).clamp(min=0)
        valid = (t_idx < lens.unsqueeze(0))
        spk_b = (spk_f.gather(0, src) * valid).to(torch.int32).contiguous()
        mval_b = (mval_f.gather(0, src) * valid).contiguous()

        # ---- three generators, concurrently
        keys = ("acoustic", "visual", "text")                        # model.py:1521-1523
        adds = {}
        cur = torch.cuda.current_stream()
        if self.streams is not None:
            fork = torch.cuda.Event()
            fork.record(cur)
        for i, k in enumerate(keys):
            self.ws = self.ws3[k]
            if self.streams is not None and self.streams[i].cuda_stream != cur.cuda_stream:
                self.streams[i].wait_event(fork)
                if batch[k].is_cuda:
                    batch[k].record_stream(self.streams[i])      # the caller may drop the batch while this stream still reads it
                with torch.cuda.stream(self.streams[i]):
                    adds[k] = self._net_fwd(self.G[k], self.pass_G[k], batch[k], train=train, save=train)
            else:
                adds[k] = self._net_fwd(self.G[k], self.pass_G[k], batch[k], train=train, save=train)
        if self.streams is not None:
            for s_ in self.streams:
                if s_.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(s_)
        st = st_()
        _lib.call("ganffn_add3", P(self.pass_G["acoustic"].out), P(self.pass_G["visual"].out), P(self.pass_G["text"].out),
                  P(f["fusion"]), C.c_int64(T * Dm), st)
        _lib.call("ganffn_seq_reverse", P(f["fusion"]), P(lens), P(f["rev_U"]), S, B, Dm, 0, st)
        # ---- the recurrence, both directions
        cfg = self.cfg_train if train else self.cfg_eval
        arr = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
        U_, spk_, mval_ = arr([f["fusion"], f["rev_U"]]), arr([spk_f, spk_b]), arr([mval_f, mval_b])
        e_, al_, sv_, ws_ = arr([f["e_f"], f["e_b"]]), arr([f["alpha_f"], f["alpha_b"]]), arr([f["saved_f"], f["saved_b"]]), arr([f["ws_f"], f["ws_b"]])
        Pp = self._drnn_ptrs(False)
        _lib.call("ganffn_drnn_fwd", C.byref(cfg), 2, U_, spk_, mval_, Pp, e_, al_, sv_, ws_, P(rng), C.c_uint64(a_rec), st)
        # ---- head: emotions -> matching attention -> linear/relu/dropout -> classes -> loss
        tr = 1 if train else 0
        _lib.call("ganffn_drnn_join_fwd", P(f["e_f"]), P(f["e_b"]), P(lens), P(f["emotions"]), S, B, He, C.c_float(self.p_join),
                  C.c_uint32(SITE_JOIN_F), C.c_uint32(SITE_JOIN_B), P(rng), C.c_uint64(a_head), tr, st)
        w_t, b_t, w_l, b_l, w_s, b_s = (self._hp(26 + j) for j in range(6))
        ops.linear_fwd_raw(f["emotions"], w_t, b_t, f["xq"], T, D2, D2)
        _lib.call("ganffn_general2_attention_fwd", P(f["xq"]), P(f["emotions"]), P(umask), P(f["att"]), P(f["alpha2"]), P(f["tanh_s"]),
                  S, B, D2, st)
        _lib.call("ganffn_ffn_linear1_fwd", P(f["att"]), P(w_l), P(b_l), P(f["hidden"]), T, D2, self.Dh2, C.c_float(self.p_hid),
                  C.c_uint32(SITE_HIDDEN), P(rng), C.c_uint64(a_head), tr, st)
        ops.linear_fwd_raw(f["hidden"], w_s, b_s, f["logits"], T, self.Dh2, Cn)
        log_prob = f["log_prob"][:T * Cn].view(S, B, Cn)
        ops.logsoftmax_nll_raw(f["logits"], batch["label"], umask, self.class_w, log_prob, self.loss,
                               f["dlogits"] if train else None, self.ws2, S, B, Cn)
        if not train:
            return self.loss, log_prob
        # ---- backward through the head
        self.h_grad.zero_()
        g_t, gb_t, g_l, gb_l, g_s, gb_s = (self._hp(26 + j, True) for j in range(6))
        ops.linear_bwd_raw(f["dlogits"], f["hidden"], w_s, f["d_hidden"], g_s, gb_s, T, self.Dh2, Cn, f["lin_ws"])
        mscale = 1.0 / (1.0 - self.p_hid) if self.p_hid > 0 else 1.0
        _lib.call("ganffn_mask_pos_inplace", P(f["d_hidden"]), P(f["hidden"]), C.c_float(mscale), C.c_int64(T * self.Dh2), st)
        ops.linear_bwd_raw(f["d_hidden"], f["att"], w_l, f["d_att"], g_l, gb_l, T, D2, self.Dh2, f["lin_ws"])
        _lib.call("ganffn_general2_attention_bwd", P(f["d_att"]), P(f["xq"]), P(f["emotions"]), P(umask), P(f["alpha2"]), P(f["tanh_s"]),
                  P(f["du"]), P(f["d_xq"]), P(f["d_mem"]), S, B, D2, st)
        ops.linear_bwd_raw(f["d_xq"], f["emotions"], w_t, f["d_em"], g_t, gb_t, T, D2, D2, f["lin_ws"])
        f["d_em"][:T * D2].add_(f["d_mem"][:T * D2])          # memory path of the attention + its transform(mem) path
        _lib.call("ganffn_drnn_join_bwd", P(f["d_em"]), P(lens), P(f["d_e_f"]), P(f["d_e_b"]), S, B, He, C.c_float(self.p_join),
                  C.c_uint32(SITE_JOIN_F), C.c_uint32(SITE_JOIN_B), P(rng), C.c_uint64(a_head), tr, st)
        # ---- the recurrence backward (weight gradients accumulate into the zeroed head slab)
        Gp = self._drnn_ptrs(True)
        de_, dU_ = arr([f["d_e_f"], f["d_e_b"]]), arr([f["dU_f"], f["dU_b"]])
        _lib.call("ganffn_drnn_bwd", C.byref(cfg), 2, de_, U_, spk_, mval_, Pp, Gp, dU_, al_, sv_, ws_, P(rng), C.c_uint64(a_rec), st)
        # d fusion = dU_f + reverse(dU_b)
        _lib.call("ganffn_seq_reverse", P(f["dU_b"]), P(lens), P(f["dU_f"]), S, B, Dm, 1, st)
        d_fusion = f["dU_f"][:T * Dm].view(S, B, Dm)
        # ---- head optimizer step (its all-reduce, when data-parallel, runs beside the generators' backward)
        red_h = None
        if self.pg is not None:
            if dp_mode() == "inline":
                # in-line on this stream (a communicator never carries two collectives at once: the generators' all-reduces
                # below use the same one)
                import torch.distributed as dist
                dist.all_reduce(self.h_grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)
            else:
                red_h = GradReducer(self.pg)
                red_h.reduce_async(self.h_grad)
        # ---- three generator backward passes + Adam, concurrently
        if self.streams is not None:
            fork = torch.cuda.Event()
            fork.record(cur)
        for i, k in enumerate(keys):
            net = self.G[k]
            self.ws = self.ws3[k]

            def bwd():
                net.grad.zero_()
                cb, finish = self._make_reducer(net)
                self._net_bwd(net, self.pass_G[k], d_fusion, True, adds[k], True, cb)
                finish(("G", k))
            if self.streams is not None and self.streams[i].cuda_stream != cur.cuda_stream:
                self.streams[i].wait_event(fork)
                with torch.cuda.stream(self.streams[i]):
                    bwd()
            else:
                bwd()
        if red_h is not None:
            red_h.finish()
        ops.adam_step_raw(self.h_slab, self.h_grad, self.h_m, self.h_v, self.h_step, self.h_total, self.lr, 0.9, 0.999, 1e-8,
                          self.wd, 1.0 / self.world)
        if self.streams is not None:
            for s_ in self.streams:
                if s_.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(s_)
        assert self._adds <= 6, self._adds
        return self.loss, log_prob

    @staticmethod
    def predictions(log_prob):
        return log_prob.transpose(0, 1).reshape(-1, log_prob.shape[2]).argmax(1)
