"""The score-matching training step as the build runs it (replaces the loop
body MSGM_higherDim.py:803-809): perturb (K1) -> probe -> fused
forward/tangent/loss/backward (K5) -> [RCCL all-reduce of the flat gradient
bucket] -> fused Adam (K13).  Everything is enqueued on one stream with no
host synchronisation, so a step can be captured once and replayed as a hipGraph
(single-GPU) — the Philox offset and the Adam step count live on the device.

Both trainers expose ``state_dict()`` / ``load_state_dict()`` in the layout of ``torch.optim.Adam`` (``state`` =
{index: step / exp_avg / exp_avg_sq}, ``param_groups``) plus a ``philox`` entry, so the fast path goes through
``NN.save_checkpoint`` / ``load_checkpoint`` like the reference loop does (MSGM_higherDim.py:794-826, NN.py:13-42)
and a resumed run continues the SAME noise stream.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from . import ops
from . import parallel
from ._lib import MsgmError


class _TrainerState:
    """Checkpoint surface shared by the trainers (Adam layout of torch.optim.Adam + the device Philox state).

    The reference builds its optimizer over ``gen_sde.parameters()`` (MSGM_higherDim.py:792,899-900): index 0 of that
    list is the non-trainable horizon ``T`` (an nn.Parameter(requires_grad=False), MSGM_higherDim.py:728), the score
    net's tensors follow.  The trainers write / read their Adam state under THOSE indices (``T`` gets a slot in
    ``param_groups[0]['params']`` and no state, exactly as torch.optim.Adam leaves it), so a trainer checkpoint loads
    into ``torch.optim.Adam(gen_sde.parameters())`` / ``FusedAdam`` and vice versa."""

    def _param_index(self):
        """(all parameters of gen_sde in optimizer order, [(index, flat offset, parameter)] of the net's tensors)."""
        offs, off = {}, 0
        for p in self.net.parameters():
            offs[id(p)] = off
            off += p.numel()
        allp = list(self.gen_sde.parameters())
        mine = [(i, offs[id(p)], p) for i, p in enumerate(allp) if id(p) in offs]
        if len(mine) != len(offs):
            raise MsgmError("the score net's parameters are not all reachable from gen_sde.parameters()")
        return allp, mine

    def state_dict(self) -> dict:
        step = float(self.step_dev.item())
        allp, mine = self._param_index()
        st = {}
        if step > 0:                              # a never-stepped torch.optim.Adam has an empty state too
            for i, off, p in mine:
                k = p.numel()
                st[i] = {"step": torch.tensor(step), "exp_avg": self.m[off:off + k].view(p.shape).clone(),
                         "exp_avg_sq": self.v[off:off + k].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(allp)))}
        return {"state": st, "param_groups": [group], "philox": self.rng.state_dict()}

    def load_state_dict(self, sd: dict) -> None:
        allp, mine = self._param_index()
        saved = sd["param_groups"][0]["params"] if sd.get("param_groups") else list(range(len(allp)))
        if len(saved) == len(allp):
            shift = 0
        elif len(saved) == len(mine):             # written by a round-2 trainer: net tensors only, indexed from 0
            shift = mine[0][0]
        else:
            raise MsgmError(f"optimizer state has {len(saved)} parameters; gen_sde.parameters() has {len(allp)}")
        step = 0
        for i, off, p in mine:
            k = p.numel()
            e = sd["state"].get(i - shift)
            if e is None:                       # never-stepped checkpoint: empty Adam state
                self.m[off:off + k].zero_(); self.v[off:off + k].zero_()
            else:
                self.m[off:off + k].copy_(e["exp_avg"].reshape(-1)); self.v[off:off + k].copy_(e["exp_avg_sq"].reshape(-1))
                step = int(float(e["step"]))
        self.step_dev.fill_(step)
        if sd.get("param_groups"):
            lr = float(sd["param_groups"][0]["lr"])
            if lr != self.lr and self.graph is not None:
                raise MsgmError("the captured step has the learning rate baked in; build a new trainer for another lr")
            self.lr = lr
        if "philox" in sd:                      # absent in checkpoints written by torch.optim.Adam: keep the stream
            self.rng.load_state_dict(sd["philox"])      # stream position only; the shard base stays this rank's


def _msgm_draws(tr):
    """(y, t, v, u, cst) of one training step under the MULTIPLICATIVE SDE, drawn in the order and from the stream
    positions ``PluginReverseSDE.ssm`` uses (SDEs.py:684-693 sample_t, :78-122 sample_scheme, :514-515 probe): uniform t
    clamped at t_epsilon, the device-resident masked RK4 forward perturbation over the nsf-step grid
    (sde_scheme.msgm_forward_perturb), the Rademacher probe, then u = G(y)^T v, cst of the general loss form
    (msgm_ssm_terms).  Every launch is a kernel on the current stream, no host synchronisation: capturable.
    The base SDE draws from the TRAINER's Philox stream (shard placement included)."""
    from .sde_scheme import msgm_forward_perturb
    base, rng = tr.base, tr.rng
    ops.fill_uniform(tr.u, rng, L.RNG_STREAM_T)
    torch.mul(tr.u, tr.T_host, out=tr.t)
    tr.t.clamp_(min=base.t_epsilon)                  # == m*t_eps + (1-m)*t of SDEs.py:690-692, bit for bit
    y = msgm_forward_perturb(base, tr.t.view(-1, 1), tr.x)          # advances the stream by 2 nsf + 2
    v = ops.rademacher((tr.B, tr.d), tr.dev, rng=rng)
    rng.advance(1)
    uu, cst = ops.ssm_terms(y, v, tr.t, tr.st)
    return y, tr.t, v, uu, cst


def _adopt_stream(tr):
    """The multiplicative SDE's forward perturbation draws through ``base.philox()``: hand it the trainer's stream."""
    tr.msgm = tr.base.kind != L.SDE_SGM
    if tr.msgm:
        tr.base.rng = tr.rng
        tr.T_host = tr.base.T_float()
        tr.u = torch.empty(tr.B, dtype=torch.float32, device=tr.dev)


class MLPScoreTrainer(_TrainerState):
    """MLP score net (configs C1/C2) under the additive (SGMsde) or the multiplicative SDE (MSGMsde, dense or sparse
    tensor: MSGM_higherDim.py:733-746).  ``x`` is this rank's shard (B_local, d) and stays resident; each ``step()`` draws
    fresh (t, eps | forward-SDE noise, v) on the device."""

    def __init__(self, gen_sde, batch_local: int, lr: float = 1e-3, world: int = 1, use_graph: bool = True,
                 seed: int = 0, row_base: int = 0):
        from .NN import MLP
        net, base = gen_sde.a, gen_sde.base_sde
        if not isinstance(net, MLP):
            raise MsgmError("MLPScoreTrainer is built for the MLP score net (SGMsde or MSGMsde)")
        self.gen_sde, self.net, self.base = gen_sde, net, base
        self.dev = next(net.parameters()).device
        self.B, self.d, self.world = batch_local, net.input_dim, world
        self.flat, self.gflat = net.flat_parameters()
        self.n = self.flat.numel()
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)
        # same seed on every rank + the shard's first global row: the N-rank run draws what the 1-rank run draws
        self.rng = L.PhiloxState(seed * 1000003 + 17, self.dev, row_base=row_base, n=self.d)
        self.ws = ops.mlp_ssm_workspace(self.d, net.pre is not None, self.dev)
        # ONE all-reduce bucket per step: [flat gradient | mean loss] = the net's own gradient bucket
        self.gbuf = net.grad_bucket()
        self.loss = self.gbuf[self.n:]
        self.x = torch.zeros(batch_local, self.d, dtype=torch.float32, device=self.dev)
        self.y = torch.empty_like(self.x)
        self.t = torch.empty(batch_local, dtype=torch.float32, device=self.dev)
        self.vp = torch.empty_like(self.x)
        self.lr = lr
        self.P = net.kernel_params()
        self.st = base.struct()
        self.inv_batch = 1.0 / (batch_local * world)        # mean over the GLOBAL batch
        self.graph = None
        self.use_graph = use_graph and not parallel.multi(world)
        _adopt_stream(self)

    def _body(self):
        """3 launches on one GPU: prep (K1 + probe + step tick) -> fused SSM kernel ->
        slab reduction fused with Adam (+ Philox advance).  With N>1 ranks the
        reduction and Adam are split around the RCCL all-reduce of the flat bucket."""
        import ctypes as C
        lib, s = ops.lib(), ops.stream()
        if self.msgm:
            # multiplicative SDE: no closed-form perturbation — masked RK4 on the device, then the general loss form
            y, t, vp, uu, cst = _msgm_draws(self)
            ops.counter_inc(self.step_dev)
            pu, pc, adv = uu.data_ptr(), cst.data_ptr(), None       # the draws advanced the stream themselves
            self._live = (y, vp, uu, cst)
        else:
            ops.check(lib.msgm_ssm_prep(self.x.data_ptr(), self.y.data_ptr(), self.t.data_ptr(), self.vp.data_ptr(), self.B,
                                        self.d, self.st, self.rng.ptr(), self.step_dev.data_ptr(), s), "msgm_ssm_prep")
            y, t, vp, pu, pc, adv = self.y, self.t, self.vp, None, None, self.rng.ptr()
        nsl = C.c_int32(0)
        ops.check(lib.msgm_mlp_ssm_partial(self.P, y.data_ptr(), t.data_ptr(), vp.data_ptr(), pu, pc, self.B,
                                           self.st, self.inv_batch, None, self.ws.data_ptr(), self.ws.numel() * 4,
                                           C.byref(nsl), s), "msgm_mlp_ssm_partial")
        pre = int(self.net.pre is not None)
        if not parallel.multi(self.world):
            ops.check(lib.msgm_mlp_ssm_reduce_adam(self.d, pre, self.ws.data_ptr(), nsl.value, self.inv_batch,
                                                   self.gflat.data_ptr(), self.loss.data_ptr(), self.flat.data_ptr(),
                                                   self.m.data_ptr(), self.v.data_ptr(), self.lr, 0.9, 0.999, 1e-8,
                                                   self.step_dev.data_ptr(), adv, s), "msgm_mlp_ssm_reduce_adam")
        else:
            ops.check(lib.msgm_mlp_ssm_reduce(self.d, pre, self.ws.data_ptr(), nsl.value, self.inv_batch,
                                              self.gflat.data_ptr(), self.loss.data_ptr(), s), "msgm_mlp_ssm_reduce")
            parallel.allreduce_sum_(self.gbuf)    # grads and loss already carry 1/global_batch
            ops.adam_step(self.flat, self.gflat, self.m, self.v, step=0, lr=self.lr, step_dev=self.step_dev)
            if not self.msgm:
                self.rng.advance(1)

    def capture(self):
        """One ordinary step on a side stream (loads code objects; it is a REAL training step: parameters, Adam state,
        Philox offset and step counter all live on the device), then the capture — which records and executes nothing."""
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            self._body()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        self.graph = ops.new_graph()
        with torch.cuda.graph(self.graph):
            self._body()

    def set_data(self, x: torch.Tensor):
        self.x.copy_(x)

    def step(self):
        if self.use_graph:
            if self.graph is None:
                self.capture()           # its warm-up IS this call's training step (one update per call)
                return self.loss
            self.graph.replay()
        else:
            self._body()
        return self.loss


class UNetScoreTrainer(_TrainerState):
    """A U-Net score net exposing ``ssm_grad`` (configs C3/C4) under the additive SDE or the multiplicative one (the
    paper's model: MSGMsde, sparse tensor for d >= 256, nsf forward RK4 steps — MSGM_higherDim.py:733-746): prep kernel
    (K1 + probe) -> dual-number forward / hand-written backward -> [RCCL all-reduce
    of the flat gradient bucket] -> fused Adam on the flat parameter bucket.

    The step enqueues ~2000 small launches for the 2-D U-Net; at the C4 shard size
    (32 samples per GPU) the launch gaps cost as much as the kernels, so everything
    before the collective is captured ONCE as a hipGraph and replayed (``use_graph``);
    with one rank the Adam update and the Philox advance are inside the graph too."""

    def __init__(self, gen_sde, batch_local: int, dim: int, lr: float = 1e-4, world: int = 1, seed: int = 0,
                 use_graph: Optional[bool] = None, row_base: int = 0):
        net, base = gen_sde.a, gen_sde.base_sde
        if not hasattr(net, "ssm_grad"):
            raise MsgmError("UNetScoreTrainer needs a HIP U-Net score net (SGMsde or MSGMsde)")
        self.gen_sde, self.net, self.base = gen_sde, net, base
        self.dev = next(net.parameters()).device
        self.B, self.d, self.world, self.lr = batch_local, dim, world, lr
        self.flat, self.gflat = net.flat_parameters()
        self.n = self.flat.numel()
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.gbuf = net.grad_bucket()            # ONE bucket [grads | mean loss]: the net's gradient bucket IS the
        self.loss = self.gbuf[self.n:]           # collective buffer (no 16 MB staging copy per step)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.rng = L.PhiloxState(seed * 1000003 + 29, self.dev, row_base=row_base, n=dim)
        self.x = torch.zeros(batch_local, dim, dtype=torch.float32, device=self.dev)
        self.y, self.vp = torch.empty_like(self.x), torch.empty_like(self.x)
        self.t = torch.empty(batch_local, dtype=torch.float32, device=self.dev)
        self.st = base.struct()
        self.inv_batch = 1.0 / (batch_local * world)
        # default: graph everywhere.  With N ranks the graph ends before the collective (capture_error_mode
        # "thread_local": the process group's watchdog thread may touch the runtime during the capture), the
        # all-reduce and Adam follow eagerly on the same stream
        self.use_graph = True if use_graph is None else bool(use_graph)
        self.graph = None
        _adopt_stream(self)

    def set_data(self, x):
        self.x.copy_(x)

    def _fwd_bwd(self):
        """Everything before the collective; leaves [grads | loss] (already x 1/global_batch) in ``gbuf``."""
        lib, s = ops.lib(), ops.stream()
        if self.msgm:
            y, t, vp, u, cst = _msgm_draws(self)
            ops.counter_inc(self.step_dev)
        else:
            ops.check(lib.msgm_ssm_prep(self.x.data_ptr(), self.y.data_ptr(), self.t.data_ptr(), self.vp.data_ptr(), self.B,
                                        self.d, self.st, self.rng.ptr(), self.step_dev.data_ptr(), s), "msgm_ssm_prep")
            y, t, vp = self.y, self.t, self.vp
            u, cst = ops.ssm_terms(y, vp, t, self.st)
        per = self.net.ssm_grad(y, t, vp, u, cst, self.inv_batch)
        flat, gflat = self.net.flat_parameters()
        if flat.data_ptr() != self.flat.data_ptr() or gflat.data_ptr() != self.gbuf.data_ptr():
            raise MsgmError("the flat parameter bucket moved; rebuild the trainer")
        # kernel nodes only inside the captured step (no D2D memcpy / memset nodes: see msgm_zero_async in csrc/common.h)
        torch.sum(per.view(1, -1), dim=1, out=self.loss)
        self.loss.mul_(self.inv_batch)

    def _update(self):
        ops.adam_step(self.flat, self.gbuf[: self.n], self.m, self.v, step=0, lr=self.lr, step_dev=self.step_dev)
        if not self.msgm:                        # the multiplicative SDE's draws advance the stream as they go
            self.rng.advance(1)

    def _collective_update(self):
        if parallel.multi(self.world):
            parallel.allreduce_sum_(self.gbuf)       # ONE collective: flat gradient bucket + the loss scalar
        self._update()

    def capture(self):
        """One ordinary eager step with the pre-collective part on a side stream (allocations, packed-weight
        images, one-time kernel attributes), then the capture.  Parameters, Adam state, Philox offset and step
        counter all live on the device, so the warm-up is a real training step and every replay is the next one.
        With one rank the graph holds the whole step; with N ranks it ends before the RCCL all-reduce."""
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self._fwd_bwd()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self._collective_update()
        torch.cuda.synchronize(self.dev)
        self._eager_step_done = True
        graph = ops.new_graph()
        mode = "global" if not parallel.multi(self.world) else "thread_local"      # a collective backend's watchdog thread must not
        with torch.cuda.graph(graph, capture_error_mode=mode):       # invalidate the capture
            self._fwd_bwd()
            if not parallel.multi(self.world):
                self._update()
        self.graph = graph

    def step(self):
        if not self.use_graph:
            self._fwd_bwd()
            self._collective_update()
            return self.loss
        if self.graph is None:
            try:
                self.capture()                       # performs this call's step eagerly
            except Exception as e:                   # noqa: BLE001
                # capture() runs the eager step FIRST, so this call's update is already done when the capture itself
                # fails.  With N ranks a failed capture next to a live RCCL communicator must not take the job down:
                # log it and stay on the eager path in the same process (N > 1 RCCL capture is unverified on hardware).
                if not parallel.multi(self.world) or not getattr(self, "_eager_step_done", False):
                    raise
                import sys
                print(f"[msgm] hipGraph capture of the train step failed with {self.world} ranks ({type(e).__name__}: {e}); "
                      "continuing eagerly", file=sys.stderr)
                self.graph, self.use_graph = None, False
            return self.loss
        self.graph.replay()
        if parallel.multi(self.world):
            self._collective_update()
        return self.loss
