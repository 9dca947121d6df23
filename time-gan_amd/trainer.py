"""Three-phase TimeGAN training steps, mirroring the loop bodies of
outdated_files/TimeGANv1.py:573-812 (embedding_trainer, supervisor_trainer, joint_trainer), with
optional data parallelism (one process per GPU, gradients summed over ranks with RCCL).

A "step" here is the body of the reference's `for idx, real_data in enumerate(dl)` loop for one
batch; the plotting / wandb block of the joint phase (TimeGANv1.py:814-896) is not part of it.
"""
from typing import Dict, List, Optional, Sequence

from contextlib import contextmanager

import torch
from torch.optim import Adam

from . import losses
from .deferred import GradArena, deferred_reductions
from .modules import build_networks
from .optim import AdamStep, init_adam_state
import torch.distributed as dist

from .parallel import (all_gather_rows, all_reduce_grads, broadcast_int, broadcast_parameters, equal_shard_extent,
                       multi_rank, shard_batch)


class _GraphedStep:
    """Runs `fn(*tensors) -> tuple of tensors` eagerly for the first `warmup` calls (they are real
    steps), then captures it once into a HIP graph and replays it: one graph launch instead of
    hundreds of kernel launches.  Inputs are copied into static buffers; outputs are static tensors.
    A call with different shapes (a ragged last batch) runs eagerly.  A tensor that IS the static buffer
    (the trainer passes its own persistent input buffers) is not copied."""

    def __init__(self, fn, warmup: int = 2, capture_mode: str = "global"):
        self.fn, self.warmup, self.calls, self.capture_mode = fn, warmup, 0, capture_mode
        self.graph, self.static_in, self.static_out, self.shapes = None, None, None, None

    def __call__(self, *tensors):
        shapes = tuple(tuple(t.shape) for t in tensors)
        if self.calls < self.warmup or (self.shapes is not None and shapes != self.shapes):
            self.calls += 1
            return self.fn(*tensors)
        if self.graph is None:
            self.shapes = shapes
            self.static_in = list(tensors)      # the caller's buffers become the graph's inputs
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode=self.capture_mode):
                self.static_out = self.fn(*self.static_in)
        for s, t in zip(self.static_in, tensors):
            if s is not t:
                s.copy_(t)
        self.graph.replay()
        return self.static_out


class TimeGANTrainer:
    """Owns the five networks and the six Adam optimisers of TimeGANv1.py:926-938.

    The embedder has TWO optimisers, one used by the embedding phase and one by the joint phase
    (`emb_opt_side` / `emb_opt_main`, TimeGANv1.py:933-934), each with its own moment estimates.
    """

    def __init__(self, cfg: Dict, device=None, rnn_cls=None, group=None, nets=None, use_graphs: bool = False,
                 all_gradients: bool = False, counter_noise: Optional[bool] = None):
        """`all_gradients`: also form the gradients of the networks a sub-step runs through but does not
        update, as the reference does before it zeroes them unseen (see `_frozen`); same parameter updates,
        more work -- for loops that want to look at those gradients.
        `counter_noise`: draw the joint phase's noise from the library's counter-based generator (see
        `_noise_like`) instead of torch's; default: with several ranks, where it lets every rank draw just its
        shard's rows of the global batch's noise.  True in a single process gives that process the very noise a
        data-parallel run of the same seed uses."""
        self.cfg = cfg
        self.all_gradients = bool(all_gradients)
        self.device = torch.device(device if device is not None else cfg["system"]["device"])
        kw = {} if rnn_cls is None else {"rnn_cls": rnn_cls}
        self.nets = nets if nets is not None else build_networks(cfg, **kw)
        for k in self.nets:
            self.nets[k] = self.nets[k].to(self.device)
        self.group = group
        self._params = {}
        self._arenas = {}              # per optimiser group: the arena its gradients live in (several ranks only)
        self.counter_noise = multi_rank(group) if counter_noise is None else bool(counter_noise)
        self._noise_seed = None        # seed of the counter-based noise, drawn at its first use
        self._noise_draws = 0          # noise tensors drawn so far: the stream number of the next one
        self._extent = None            # (start, stop, total rows) of the current shard, when fit() knows it
        broadcast_parameters(self.nets.values(), group=group)
        lr = float(cfg["system"]["lr"])
        n = self.nets
        # HIP-graph mode: every optimiser sub-step (zero_grad + forward + backward [+ all-reduces] + Adam)
        # is one captured graph.  With several ranks the collectives are captured too, which needs the
        # RCCL ("nccl") backend; every rank takes the same host-side decisions (the discriminator
        # loss it compares with the threshold is the global one), so all ranks replay the same graphs.
        dp = multi_rank(group)
        self.use_graphs = (bool(use_graphs) and self.device.type == "cuda"
                           and (not dp or dist.get_backend(group) == "nccl"))
        # fused=True keeps Adam's step counters on the device, where the one-launch update (optim.AdamStep)
        # and graph capture need them.
        kw_opt = {"fused": True} if self.device.type == "cuda" else {}
        if self.use_graphs:
            kw_opt["capturable"] = True
        self.opt = {"emb_side": Adam(n["emb"].parameters(), lr=lr, **kw_opt),
                    "emb_main": Adam(n["emb"].parameters(), lr=lr, **kw_opt),
                    "rec": Adam(n["rec"].parameters(), lr=lr, **kw_opt), "sup": Adam(n["sup"].parameters(), lr=lr, **kw_opt),
                    "g": Adam(n["g"].parameters(), lr=lr, **kw_opt), "d": Adam(n["d"].parameters(), lr=lr, **kw_opt)}
        self.d_threshold = float(cfg["d"]["threshold"])
        self.seq_len = int(cfg["system"]["seq_len"])
        if self.use_graphs:
            # with collectives inside, the backend's own threads touch the device during capture
            mode = "thread_local" if dp else "global"
            self._g_emb = _GraphedStep(self._embedding_core, capture_mode=mode)
            self._g_sup = _GraphedStep(self._supervisor_core, capture_mode=mode)
            self._g_joint = _GraphedStep(self._joint_iter_core, capture_mode=mode)
            self._thr = torch.tensor(self.d_threshold, device=self.device)   # the threshold the D graph reads
            self._thr_host = self.d_threshold
            self._bufs = {}
        # On the GPU the optimisers of a sub-step are stepped together by ONE kernel that works on their
        # own state (optim.AdamStep), so the state is created up front.
        self._adam = {}
        self._one = torch.ones((), device=self.device)     # d loss / d loss, instead of a ones_like() per backward
        if self.device.type == "cuda":
            for o in self.opt.values():
                init_adam_state(o)

    # -- helpers ---------------------------------------------------------------------------------
    def _step(self, *names, skip=None):
        """Adam update of the named optimisers: one launch for all of them on the GPU."""
        if self.device.type != "cuda":
            for k in names:
                self.opt[k].step()
            return
        stepper = self._adam.get(names)
        if stepper is None:
            stepper = self._adam[names] = AdamStep([self.opt[k] for k in names], self.device)
        stepper.step(skip)

    def _input_buffer(self, name: str, like: torch.Tensor) -> torch.Tensor:
        """Persistent device buffer that the captured graphs read their input from."""
        buf = self._bufs.get(name)
        if buf is None or buf.shape != like.shape:
            buf = torch.empty_like(like)
            self._bufs[name] = buf
        return buf

    def _staged(self, name: str, src: torch.Tensor) -> torch.Tensor:
        buf = self._input_buffer(name, src)
        buf.copy_(src)
        return buf

    @contextmanager
    def _frozen(self, *names):
        """The networks a sub-step runs through but does not update.  The reference computes their gradients
        and zeroes them unseen before their next use (TimeGANv1.py:765-769,780-782,796-799; :667-668 for the
        supervisor phase); here their parameters ask for none, so the weight-gradient kernels of those
        networks are not launched, and a network fed by data alone (the embedder in the generator and
        supervisor sub-steps) has no backward pass at all.  The updated parameters are the same."""
        ps = [] if self.all_gradients else [p for k in names for p in self.nets[k].parameters() if p.requires_grad]
        for p in ps:
            p.requires_grad_(False)
        try:
            yield
        finally:
            for p in ps:
                p.requires_grad_(True)

    def _zero(self, *names):
        """zero_grad() of the named networks (gradients set to None, torch's default), from cached
        parameter lists: Module.zero_grad() walks the module tree on every call."""
        for k in names:
            params = self._params.get(k)
            if params is None:
                params = self._params[k] = list(self.nets[k].parameters())
            for p in params:
                p.grad = None

    def reset_optimizers(self) -> None:
        """Adam state (moment estimates, step counters) of all six optimisers back to zero, in place -- the
        state a freshly built optimiser has.  The reference runs ONE phase per time_gan_trainer() call with
        newly built optimisers (TimeGANv1.py:932-938) and hands only state_dicts of the networks from phase
        to phase (:951-1009)."""
        for opt in self.opt.values():
            for st in opt.state.values():
                for k in ("exp_avg", "exp_avg_sq", "step"):
                    v = st.get(k)
                    if isinstance(v, torch.Tensor):
                        v.zero_()
                    elif v is not None:
                        st[k] = 0
        if self.device.type != "cuda":          # torch's own step() re-creates state that is absent
            for opt in self.opt.values():
                opt.state.clear()

    def _pass(self, *names):
        """Context of one backward pass whose gradients for the networks `names` are exchanged afterwards: the
        partial-gradient reductions of the pass are deferred into one launch, and with several ranks the pass's
        gradient buffers come from one arena per optimiser group, which `_sync_grads` all-reduces in place."""
        arena = None
        if multi_rank(self.group) and self.device.type == "cuda":
            arena = self._arenas.get(names)
            if arena is None:
                # room for every gradient of the group twice over: the head kernels' gradient blocks also hold
                # the slots of stages a head does not have, and every view starts on a 16-byte boundary
                n = sum(p.numel() + 64 for k in names for p in self.nets[k].parameters())
                arena = self._arenas[names] = GradArena(2 * n, self.device)
        return deferred_reductions(arena)

    def _sync_grads(self, *names):
        all_reduce_grads([self.nets[k] for k in names], group=self.group, arena=self._arenas.get(names))

    def _noise_like(self, x: torch.Tensor, count: int, out: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """`count` noise tensors shaped like x, U[0,1) as torch.rand_like(x) draws them
        (TimeGANv1.py:762,793); into the rows of `out` (count, *x.shape) when given.  With several ranks, x is this
        rank's shard of a global batch and the ranks must use different rows of ONE stream -- the rows a single
        process would draw for the concatenated batch (identically seeded ranks drawing rand_like(shard) would all
        get the same rows: a generated batch of duplicates, whose batch statistics degenerate).  On the GPU the
        library's counter-based generator draws exactly the shard's rows (element = global row x row width +
        offset, one stream per noise tensor); on the CPU every rank draws the whole batch from its identically
        seeded torch generator and keeps its rows.  Where the shard sits comes from `fit` (arithmetic on the
        global batch); a direct caller's shards are taken to be equal unless it sets `self._extent`."""
        if not self.counter_noise and not multi_rank(self.group):
            if out is not None:
                out.uniform_()
                return list(out)
            return [torch.rand_like(x) for _ in range(count)]
        lo, hi, total = self._extent if self._extent is not None else equal_shard_extent(int(x.shape[0]), self.group)
        assert hi - lo == x.shape[0], "the shard extent does not match the batch"
        if x.is_cuda and self.counter_noise:
            from .functional import uniform_
            buf = out if out is not None else torch.empty((count,) + tuple(x.shape), device=x.device, dtype=torch.float32)
            row = x[0].numel()
            if self._noise_seed is None:
                # from torch's generator (so torch.manual_seed governs it); every rank uses rank 0's
                self._noise_seed = broadcast_int(int(torch.randint(0, 2 ** 62, (1,)).item()), self.device, group=self.group)
            for i in range(count):
                uniform_(buf[i], self._noise_seed, self._noise_draws + i, lo * row)
            self._noise_draws += count
            return list(buf)
        shape = (total,) + tuple(x.shape[1:])
        zs = [torch.rand(shape, device=x.device, dtype=x.dtype)[lo:hi].contiguous() for _ in range(count)]
        if out is not None:
            for i in range(count):
                out[i].copy_(zs[i])
            return list(out)
        return zs

    def lengths_of(self, x: torch.Tensor) -> List[int]:
        """What Energy.extract_time returns for a batch (data/Energy.py:12-19): every length = T."""
        return [x.shape[1]] * x.shape[0]

    # -- sub-step cores: tensors in, loss tensors out, no host synchronisation ------------------------
    def _embedding_core(self, x):
        n = self.nets
        self._zero("emb", "rec", "sup")
        e_loss0, _ = losses.embedding_forward_side(n["emb"], n["rec"], x, self.lengths_of(x), group=self.group)
        with self._pass("emb", "rec"):     # one launch sums the partial gradients of the whole pass
            e_loss0.backward(self._one)
        self._sync_grads("emb", "rec")
        self._step("emb_side", "rec")
        return (e_loss0.detach(),)

    def _supervisor_core(self, x):
        n = self.nets
        self._zero("emb", "sup")
        with self._frozen("emb"), self._pass("sup"):     # one launch sums the partial gradients of the whole pass
            s_loss, _, _ = losses.supervisor_forward(n["emb"], n["sup"], x, self.lengths_of(x), group=self.group)
            s_loss.backward(self._one)
        self._sync_grads("sup")
        self._step("sup")
        return (s_loss.detach(),)

    def _latent(self, x):
        """emb(x) for a generator sub-step AND the embedder sub-step that follows it: the generator sub-step
        updates the generator and the supervisor only (TimeGANv1.py:776-777), so the embedder sub-step's
        emb(x) (:784) is this very tensor.  It is evaluated once, in grad mode; the generator sub-step reads
        its value, the embedder sub-step differentiates it.  None when every gradient is wanted (the generator
        loss then differentiates the embedder too, in a graph of its own, as in the reference)."""
        if self.all_gradients:
            return None
        return self.nets["emb"](x, self.lengths_of(x))

    def _joint_g_core(self, x, z, h=None):
        n = self.nets
        self._zero("emb", "rec", "sup", "g", "d")
        with self._frozen("emb", "rec", "d"), self._pass("g", "sup"):   # one launch sums the partial gradients of the pass
            g_loss = losses.generator_forward(n["emb"], n["sup"], n["g"], n["d"], n["rec"], x, self.lengths_of(x), z,
                                              group=self.group, h=None if h is None else h.detach())
            g_loss.backward(self._one)
        self._sync_grads("g", "sup")
        self._step("g", "sup")
        return (g_loss.detach(),)

    def _joint_e_core(self, x, h=None):
        n = self.nets
        self._zero("emb", "rec", "sup")
        with self._frozen("sup"), self._pass("emb", "rec"):     # one launch sums the partial gradients of the whole pass
            e_loss, _, _ = losses.embedding_forward_main(n["emb"], n["sup"], n["rec"], x, self.lengths_of(x),
                                                         group=self.group, h=h)
            e_loss.backward(self._one)
        self._sync_grads("emb", "rec")
        self._step("emb_main", "rec")
        return (e_loss.detach(),)

    def _joint_d_core(self, x, z):
        """Graph mode: the whole discriminator sub-step with the `d_loss > threshold` decision
        (TimeGANv1.py:805-808) taken ON THE DEVICE: forward and backward always run (an unused backward
        changes nothing: the gradients are zeroed before their next use, :765-769,796-799) and the
        Adam kernel receives `d_loss <= threshold` as its skip flag, which leaves parameters, moments and
        step counters untouched.
        No host synchronisation inside the iteration."""
        n = self.nets
        self._zero("emb", "sup", "g", "d")
        d_loss = losses.discriminator_forward(n["emb"], n["sup"], n["g"], n["d"], x, self.lengths_of(x), z,
                                              group=self.group)
        with self._pass("d"):     # one launch sums the partial gradients of the whole pass
            d_loss.backward(self._one)
        self._sync_grads("d")
        skip = (d_loss.detach() <= self._thr).to(torch.float32)
        self._step("d", skip=skip)
        return (d_loss.detach(), 1.0 - skip)

    def _joint_iter_core(self, x, zb):
        """Graph mode: a WHOLE joint iteration -- 10 x (generator + embedder sub-step) and the discriminator
        sub-step with its on-device decision -- as one function of x and 11 noise tensors zb[0..10], so that
        it can be captured into a single graph: one launch per iteration, every loss in its own tensor.
        Returns (g_0, e_0, ..., g_9, e_9, d_loss, d_stepped)."""
        outs = []
        for i in range(10):
            h = self._latent(x)
            outs += self._joint_g_core(x, zb[i], h)
            outs += self._joint_e_core(x, h)
        outs += self._joint_d_core(x, zb[10])
        return tuple(outs)

    # -- phase 1: embedding / recovery autoencoder (TimeGANv1.py:585-608) ---------------------------
    def embedding_step(self, x: torch.Tensor, t=None) -> float:
        if t is None:
            (loss,) = self._g_emb(self._staged("x", x)) if self.use_graphs else self._embedding_core(x)
            return float(loss)
        n = self.nets
        self._zero("emb", "rec", "sup")
        e_loss0, _ = losses.embedding_forward_side(n["emb"], n["rec"], x, t, group=self.group)
        with self._pass("emb", "rec"):     # one launch sums the partial gradients of the whole pass
            e_loss0.backward(self._one)
        self._sync_grads("emb", "rec")
        self._step("emb_side", "rec")
        return float(e_loss0.detach())

    # -- phase 2: supervisor (TimeGANv1.py:662-683) -------------------------------------------------
    def supervisor_step(self, x: torch.Tensor, t=None) -> float:
        if t is None:
            (loss,) = self._g_sup(self._staged("x", x)) if self.use_graphs else self._supervisor_core(x)
            return float(loss)
        n = self.nets
        self._zero("emb", "sup")
        with self._frozen("emb"), self._pass("sup"):     # one launch sums the partial gradients of the whole pass
            s_loss, _, _ = losses.supervisor_forward(n["emb"], n["sup"], x, t, group=self.group)
            s_loss.backward(self._one)
        self._sync_grads("sup")
        self._step("sup")
        return float(s_loss.detach())

    # -- phase 3: joint adversarial training (TimeGANv1.py:750-812) --------------------------------
    def joint_step(self, x: torch.Tensor, t=None, noise: Optional[Sequence[torch.Tensor]] = None) -> Dict:
        """One joint iteration: 10 x (generator step + embedder step) and one discriminator step.

        `noise`: optional 11 tensors shaped like x used in place of torch.rand_like(x)
        (TimeGANv1.py:762,793), so runs can be reproduced exactly.
        """
        if t is not None:
            return self._joint_step_with_lengths(x, t, noise)
        noise = list(noise) if noise is not None else None
        if self.use_graphs:
            # one graph holds the whole iteration; it reads x and the 11 noise tensors from persistent
            # buffers (x staged once, the noise drawn or copied straight into its buffer)
            x = self._staged("x", x)
            zb = self._bufs.get("z11")
            if zb is None or zb.shape[1:] != x.shape:
                zb = self._bufs["z11"] = torch.empty((11,) + tuple(x.shape), device=x.device, dtype=x.dtype)
            if noise is not None:
                for i in range(11):
                    zb[i].copy_(noise[i])
            else:
                self._noise_like(x, 11, out=zb)
            if float(self._thr_host) != self.d_threshold:     # the threshold was changed since the last call
                self._thr.fill_(self.d_threshold)
                self._thr_host = self.d_threshold
            outs = self._g_joint(x, zb)
            vals = torch.stack(list(outs)).tolist()           # one copy to the host for all 22 numbers
            return {"g_loss": vals[0:20:2], "e_loss": vals[1:20:2], "d_loss": vals[20], "d_stepped": vals[21] != 0.0}

        if noise is None:
            noise = self._noise_like(x, 11)

        def draw():
            return noise.pop(0)
        g_losses, e_losses = [], []
        for _ in range(10):
            z = draw()
            h = self._latent(x)
            (g_loss,) = self._joint_g_core(x, z, h)
            g_losses.append(g_loss)
            (e_loss,) = self._joint_e_core(x, h)
            e_losses.append(e_loss)
        z = draw()
        n = self.nets
        self._zero("emb", "sup", "g", "d")
        d_loss = losses.discriminator_forward(n["emb"], n["sup"], n["g"], n["d"], x, self.lengths_of(x), z,
                                              group=self.group)
        stepped = bool(d_loss > self.d_threshold)
        if stepped:
            with self._pass("d"):     # one launch sums the partial gradients of the whole pass
                d_loss.backward(self._one)
            self._sync_grads("d")
            self._step("d")
        return {"g_loss": [float(v) for v in torch.stack(g_losses).tolist()],
                "e_loss": [float(v) for v in torch.stack(e_losses).tolist()],
                "d_loss": float(d_loss.detach()), "d_stepped": stepped}

    def _joint_step_with_lengths(self, x, t, noise):
        n = self.nets
        noise = list(noise) if noise is not None else self._noise_like(x, 11)
        out = {"g_loss": [], "e_loss": []}
        for _ in range(10):
            z = noise.pop(0)
            self._zero("emb", "rec", "sup", "g", "d")
            with self._frozen("emb", "rec", "d"), self._pass("g", "sup"):   # one launch sums the pass's partial gradients
                g_loss = losses.generator_forward(n["emb"], n["sup"], n["g"], n["d"], n["rec"], x, t, z,
                                                  group=self.group)
                g_loss.backward(self._one)
            out["g_loss"].append(float(g_loss.detach()))
            self._sync_grads("g", "sup")
            self._step("g", "sup")
            self._zero("emb", "rec", "sup")
            with self._frozen("sup"), self._pass("emb", "rec"):     # one launch sums the pass's partial gradients
                e_loss, _, _ = losses.embedding_forward_main(n["emb"], n["sup"], n["rec"], x, t, group=self.group)
                e_loss.backward(self._one)
            out["e_loss"].append(float(e_loss.detach()))
            self._sync_grads("emb", "rec")
            self._step("emb_main", "rec")
        z = noise.pop(0)
        self._zero("emb", "sup", "g", "d")
        d_loss = losses.discriminator_forward(n["emb"], n["sup"], n["g"], n["d"], x, t, z, group=self.group)
        stepped = bool(d_loss > self.d_threshold)
        if stepped:
            with self._pass("d"):     # one launch sums the partial gradients of the whole pass
                d_loss.backward(self._one)
            self._sync_grads("d")
            self._step("d")
        out["d_loss"] = float(d_loss.detach())
        out["d_stepped"] = stepped
        return out

    @torch.no_grad()
    def generate(self, z: torch.Tensor, t=None) -> torch.Tensor:
        n = self.nets
        t = [z.shape[1]] * z.shape[0] if t is None else t
        return losses.inference(n["sup"], n["g"], n["rec"], z, t)

    @torch.no_grad()
    def generate_all(self, n_total: int, seed: int = 0, chunk: int = 4096) -> torch.Tensor:
        """n_total synthetic sequences (the inference path of TimeGANv1.py:556-570 on noise U[0,1)), sharded
        over the ranks of the group: every rank draws the SAME noise from `seed`, generates its contiguous
        shard (in chunks of `chunk` sequences) and one all-gather hands every rank all n_total sequences,
        equal to what a single process generates from that seed."""
        feats = int(self.cfg["g"]["dim_latent"])
        z = torch.rand(n_total, self.seq_len, feats, generator=torch.Generator().manual_seed(seed))
        z = shard_batch(z, self.group).to(self.device)
        parts = [self.generate(z[i:i + chunk]) for i in range(0, z.shape[0], chunk)]
        x = torch.cat(parts) if parts else z.new_zeros((0, self.seq_len, int(self.cfg["rec"]["dim_output"])))
        return all_gather_rows(x, n_total, self.group)

    # -- the three phases over a dataset (TimeGANv1.py:573-812 loops, :940-1009 driver) ---------------
    def fit(self, dataset, phases: Sequence[str] = ("embedding", "supervisor", "joint"), batch_size: Optional[int] = None,
            seed: int = 0, on_batch=None, fresh_optimizers: bool = True) -> Dict[str, List]:
        """Runs the named phases in order, each for its configured number of epochs
        (cfg['emb']['num_epochs'], cfg['sup']['num_epochs'], cfg['system']['jointly_num_epochs'],
        TimeGANv1.py:581,658,745) over `dataset` (data.ResidentDataset, or anything with the same
        `batches(batch_size, shuffle, generator, rank, world)`); `batch_size` is the GLOBAL batch, of which
        every rank of the group takes its shard.  Returns the per-batch losses of each phase.
        `on_batch(phase, epoch, index, result)` is called after every batch (logging hooks, early stops by
        raising StopIteration).  `fresh_optimizers` (default): every phase starts from zeroed Adam state, as
        in the reference, where each phase is a run of its own with newly built optimisers
        (TimeGANv1.py:932-938); False carries the moment estimates of `rec` and `sup` into the joint phase."""
        from .parallel import rank as rank_of, world_size
        bs = int(batch_size if batch_size is not None else self.cfg["system"]["batch_size"])
        epochs = {"embedding": int(self.cfg["emb"]["num_epochs"]), "supervisor": int(self.cfg["sup"]["num_epochs"]),
                  "joint": int(self.cfg["system"]["jointly_num_epochs"])}
        steps = {"embedding": self.embedding_step, "supervisor": self.supervisor_step, "joint": self.joint_step}
        gen = torch.Generator().manual_seed(seed)       # same permutation on every rank
        r, w = rank_of(self.group), world_size(self.group)
        history = {}
        try:
            for phase in phases:
                if phase not in steps:
                    raise ValueError("The step should be: embedding, supervisor or joint")   # TimeGANv1.py:1012
                log = history.setdefault(phase, [])
                if fresh_optimizers:
                    self.reset_optimizers()
                for epoch in range(epochs[phase]):
                    try:        # where the shard sits in its global batch: the dataset knows by arithmetic
                        it = dataset.batches(bs, shuffle=True, generator=gen, rank=r, world=w, with_extent=True)
                        extents = True
                    except TypeError:
                        it = dataset.batches(bs, shuffle=True, generator=gen, rank=r, world=w)
                        extents = False
                    for i, item in enumerate(it):
                        x, self._extent = item if extents else (item, None)
                        if not extents and w > 1:
                            from .parallel import shard_extent
                            self._extent = shard_extent(int(x.shape[0]), x.device, self.group)
                        out = steps[phase](x)
                        log.append(out)
                        if on_batch is not None:
                            on_batch(phase, epoch, i, out)
        except StopIteration:
            pass
        finally:
            self._extent = None
        return history

    def save_models(self, directory: str, dataset: Optional[str] = None, nets: Optional[Sequence[str]] = None) -> List[str]:
        """state_dicts as the files the reference writes and reads back between its phases:
        `<directory>/<net>_<dataset>.pt` for emb, rec, sup, g, d (TimeGANv1.py:953-954,976,1006,1009);
        tensors are moved to the CPU first, as there.  Parameter names and shapes are torch's, so the files
        load into the reference's modules and the reference's files load here."""
        import os
        name = dataset if dataset is not None else self.cfg["system"].get("dataset", "synthetic")
        os.makedirs(directory, exist_ok=True)
        paths = []
        for k in (nets if nets is not None else ("emb", "rec", "sup", "g", "d")):
            path = os.path.join(directory, f"{k}_{name}.pt")
            torch.save({n: t.detach().cpu() for n, t in self.nets[k].state_dict().items()}, path)
            paths.append(path)
        return paths

    def load_models(self, directory: str, dataset: Optional[str] = None, nets: Optional[Sequence[str]] = None) -> None:
        """Inverse of save_models (tensor-only files: loaded with weights_only=True)."""
        import os
        name = dataset if dataset is not None else self.cfg["system"].get("dataset", "synthetic")
        for k in (nets if nets is not None else ("emb", "rec", "sup", "g", "d")):
            sd = torch.load(os.path.join(directory, f"{k}_{name}.pt"), map_location=self.device, weights_only=True)
            self.nets[k].load_state_dict(sd)

    def state_dicts(self) -> Dict[str, Dict[str, torch.Tensor]]:
        return {k: m.state_dict() for k, m in self.nets.items()}

    def load_state_dicts(self, sds: Dict[str, Dict[str, torch.Tensor]]) -> None:
        for k, sd in sds.items():
            self.nets[k].load_state_dict(sd)
