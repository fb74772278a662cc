"""Training engine: the G/D alternating step of Trainer._train_one_epoch (train.py:240-326) on the
HIP path, with MI355X-first memory layout: all trainable parameters of a network live in ONE flat
fp32 buffer (parameters are views), gradients in a second flat buffer, Adam state in two more, so
the optimiser is a single fused kernel launch and data-parallel gradient exchange is a single
flat-bucket all-reduce over RCCL/xGMI per network per step.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .models import p2igan as net_fns
from .modules.losses import ReconstructionLoss, discriminator_loss, generator_adv_loss
from .parallel import BucketedAllReduce, FlatParams, allreduce_mean_, broadcast_module_state


class FusedAdam:
    """torch.optim.Adam(lr, betas) semantics (train.py:125-136) as one p2i_adam launch over a FlatParams."""

    def __init__(self, fp: FlatParams, lr: float, betas=(0.0, 0.99), eps: float = 1e-8):
        self.fp, self.lr, self.betas, self.eps = fp, lr, betas, eps
        self.m = torch.zeros_like(fp.flat)
        self.v = torch.zeros_like(fp.flat)
        self.step_count = 0
        self.step_dev = self.coef = None          # graph mode: step counter + bias corrections on the device

    def use_device_step(self):
        self.step_dev = torch.tensor([self.step_count], device=self.fp.flat.device, dtype=torch.int32)
        self.coef = torch.zeros(2, device=self.fp.flat.device, dtype=torch.float32)

    def step(self):
        self.step_count += 1
        if self.step_dev is not None:
            ops.adam_step_dev(self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                              self.step_dev, self.coef)
        else:
            ops.adam_step(self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.step_count)
        inval = getattr(self.fp.module, "invalidate_weight_cache", None)   # raw-pointer update: version counters do not see it
        if inval is not None:
            inval()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.m, "exp_avg_sq": self.v, "lr": self.lr, "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        if self.step_dev is not None:
            self.step_dev.fill_(self.step_count)


def _allreduce_mean(buf: torch.Tensor, world: int):
    """Flat-bucket gradient exchange; the 1/world scaling is the p2i_axpy kernel on GPUs."""
    allreduce_mean_(buf, world, (lambda t, a: ops.axpy_(t, t, a - 1.0)) if buf.is_cuda else None)


class TrainEngine:
    def __init__(self, generator: nn.Module, discriminator: Optional[nn.Module], cfg: Dict, distributed: bool = False):
        self.G, self.D = generator, discriminator
        loss_cfg, opt_cfg = cfg["loss"], cfg["train"]["optimizer"]
        self.use_gan = bool(loss_cfg.get("use_gan", 0)) and discriminator is not None
        self.gan_type = loss_cfg.get("gan_loss", "hinge")
        self.real_label = loss_cfg.get("target_real_label", 1.0)
        self.fake_label = loss_cfg.get("target_fake_label", 0.0)
        self.adv_weight = loss_cfg.get("adversarial_weight", 0.01)
        self.rec_loss = ReconstructionLoss(k1_alpha=loss_cfg.get("k1_weight", 0.0))
        betas = (opt_cfg.get("beta1", 0.0), opt_cfg.get("beta2", 0.99))
        self.gp = FlatParams(self.G)
        self.opt_g = FusedAdam(self.gp, opt_cfg["lr"], betas)
        # direct mode: the step calls the networks' plain forward / backward functions (models/p2igan.py) itself and has
        # every gradient written straight into the flat buffers -- no autograd graph, no AccumulateGrad / add / fill / copy
        # kernels from ATen on the step.  P2I_ENGINE_AUTOGRAD=1 keeps the autograd-driven step (same kernels underneath).
        import os
        self.direct = os.environ.get("P2I_ENGINE_AUTOGRAD", "0") != "1" and isinstance(self.G, net_fns.P2IGenerator) and (
            discriminator is None or isinstance(discriminator, net_fns.P2IDiscriminator))
        self.dp = self.opt_d = None
        if self.use_gan:
            self.dp = FlatParams(self.D)
            self.opt_d = FusedAdam(self.dp, opt_cfg["lr"], betas)
        self.distributed = distributed and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.distributed else 1
        if self.distributed:
            self.broadcast_state()
        # data-parallel exchange of the generator's gradients: per-Decoder-level buckets launched from inside the backward
        # (P2I_DP_OVERLAP=0: one flat all-reduce after the whole backward, the round-2 behaviour)
        self.dp_overlap = os.environ.get("P2I_DP_OVERLAP", "1") != "0"
        self._level_ranges = None
        # parameter-only work of G's forward and of D's fake / real forwards on a side stream beside attention + IDW (_step_direct)
        self.prep_overlap = os.environ.get("P2I_PREP_OVERLAP", "1") != "0"

    def broadcast_state(self):
        """Rank 0's weights, spectral-norm u/v and frozen tensors to every rank, once."""
        broadcast_module_state(self.G, self.gp)
        if self.D is not None:
            broadcast_module_state(self.D, self.dp)

    def level_range(self, lvl: int):
        """[lo, hi) of Decoder[lvl]'s trainable parameters in the flat gradient buffer (contiguous: construction order)."""
        if self._level_ranges is None:
            base, rngs = self.gp.grad.data_ptr(), []
            for blk in self.G.Decoder:
                ps = [p for p in blk.parameters() if p.requires_grad]
                lo = min((p.grad.data_ptr() - base) // 4 for p in ps)
                hi = max((p.grad.data_ptr() - base) // 4 + p.numel() for p in ps)
                if hi - lo != sum(p.numel() for p in ps):
                    raise RuntimeError("Decoder level is not one contiguous slice of the flat gradient buffer")
                rngs.append((int(lo), int(hi)))
            self._level_ranges = rngs
        return self._level_ranges[lvl]

    # ------------------------------------------------------------------ hipGraph replay of the whole step
    def capture(self, frames, masked, masks, warmup: int = 3, mode: str = "graph"):
        """Capture ONE full G+D iteration (forward, both backward passes, both Adam steps: ~800 launches) and replay it from
        then on.  `warmup` eager steps run first (they are real training steps) so that every lazily built table and kernel
        attribute exists before the capture.  Single process only: the data-parallel path keeps eager launches around its RCCL
        exchange.
        mode "graph": replay the captured hipGraph (hipGraphLaunch of the ~800-node graph costs the host as much as the launches
        themselves on ROCm 7.2).
        mode "tape" (round 4, the native step sequencer of include/p2i_hip.h): while the step is being captured the library also
        RECORDS its launches, memsets and stream dependencies on a launch tape; train_step then re-enqueues the tape with one C
        call (p2i_tape_replay) -- ~2 us per launch instead of ~9 us of Python / ctypes work.  The capture is kept for its private
        memory pool only (every buffer the tape names lives there, at a fixed address); the hipGraph itself is never launched."""
        if self.distributed:
            raise RuntimeError("graph capture is for the single-GPU step")
        if mode not in ("graph", "tape"):
            raise ValueError(mode)
        for o in (self.opt_g, self.opt_d):
            if o is not None:
                o.use_device_step()
        self._static_in = [torch.empty_like(t) for t in (frames, masked, masks)]
        for dst, src in zip(self._static_in, (frames, masked, masks)):
            dst.copy_(src)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step_impl(*self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        self._tape = None
        import ctypes
        lib = ops._hip.load()
        with torch.cuda.graph(self._graph):
            if mode == "tape":
                ops._hip.check(lib.p2i_tape_begin(torch.cuda.current_stream().cuda_stream), "p2i_tape_begin")
            try:
                self._static_out = self._step_impl(*self._static_in)
            finally:
                if mode == "tape":
                    handle = ctypes.c_void_p()
                    ops._hip.check(lib.p2i_tape_end(ctypes.byref(handle)), "p2i_tape_end")
                    self._tape = handle
        self._host_steps_per_replay = 1
        # the capture itself executed nothing: undo the host-side counters it bumped
        for o in (self.opt_g, self.opt_d):
            if o is not None:
                o.step_count -= 1
        return warmup

    def tape_info(self):
        """(kernels, memsets, event operations, streams) of the recorded step, or None."""
        if getattr(self, "_tape", None) is None:
            return None
        import ctypes
        c = (ctypes.c_int * 4)()
        ops._hip.check(ops._hip.load().p2i_tape_info(self._tape, c), "p2i_tape_info")
        return tuple(c)

    # Automatic graph replay (opt-in: P2I_AUTO_GRAPH=1; P2I_AUTO_TAPE=1: the same trigger, replay through the launch tape).  Measured in round 3 (gpurun_out/r03a/loader_probe.log, profiles/README.md):
    # hipGraphLaunch of the ~800-node step costs the host 6-10 ms, i.e. as much as enqueueing the launches one by one (B=1: 8.97 ms
    # replayed vs 8.21 ms eager; B=8: 15.8 vs 15.6), so replay does not lift the launch bound of small steps on ROCm 7.2 and is
    # not the default.  With P2I_AUTO_GRAPH=1 the engine captures by itself once the same input shapes have repeated
    # AUTO_GRAPH_AFTER times (single process, direct engine); a batch of another shape (tail batch) then runs eagerly.
    AUTO_GRAPH_AFTER = 3

    def _auto_graph_wanted(self, frames) -> bool:
        import os
        if os.environ.get("P2I_AUTO_GRAPH", "0") != "1" and os.environ.get("P2I_AUTO_TAPE", "0") != "1":
            return False
        return not (self.distributed or not self.direct or self.phase_marks is not None or ops.PROFILE is not None)

    def train_step(self, frames, masked, masks) -> Dict[str, torch.Tensor]:
        """One iteration of train.py:240-326.  Returns 0-dim DEVICE tensors (no host sync here)."""
        if getattr(self, "_graph", None) is not None:
            if tuple(frames.shape) == tuple(self._static_in[0].shape):
                for dst, src in zip(self._static_in, (frames, masked, masks)):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
                if getattr(self, "_tape", None) is not None:
                    ops._hip.check(ops._hip.load().p2i_tape_replay(self._tape, torch.cuda.current_stream().cuda_stream), "p2i_tape_replay")
                else:
                    self._graph.replay()
                for o in (self.opt_g, self.opt_d):
                    if o is not None:
                        o.step_count += 1
                self.G.invalidate_weight_cache() if hasattr(self.G, "invalidate_weight_cache") else None
                return self._static_out
            return self._step_impl(frames, masked, masks)        # another shape (a tail batch): eager, same device-side Adam counters
        if self._auto_graph_wanted(frames):
            shp = tuple(frames.shape)
            self._same_shape_steps = getattr(self, "_same_shape_steps", 0) + 1 if getattr(self, "_last_shape", None) == shp else 0
            self._last_shape = shp
            if self._same_shape_steps >= self.AUTO_GRAPH_AFTER:
                import os
                self.capture(frames, masked, masks, warmup=0, mode="tape" if os.environ.get("P2I_AUTO_TAPE", "0") == "1" else "graph")
                return self.train_step(frames, masked, masks)
        return self._step_impl(frames, masked, masks)

    def _mark(self):
        """Phase boundary for bench.py's G-step / D-step split (HIP events on the launch stream; off unless asked for)."""
        if self.phase_marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.phase_marks.append(e)

    phase_marks = None      # set to [] to collect 4 events per step: start, after G fwd + rec loss, after the D step, end
    exchange_marks = None   # set to [] to collect (kind, start event, end event) around the data-parallel exchanges the launch stream
    #                         WAITS for ("d": D's flat all-reduce; "g": the wait for G's buckets / its flat all-reduce) -- bench.py's rccl{}

    def _exchange(self, kind, fn):
        if self.exchange_marks is None or not self.gp.grad.is_cuda:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.exchange_marks.append((kind, e0, e1))
        return out

    def _step_direct(self, frames, masked, masks) -> Dict[str, torch.Tensor]:
        """train.py:240-326 with explicit forward / backward calls (see __init__).  Gradient bookkeeping:
        G: every parameter receives exactly one gradient per step, written (or, for the few atomically accumulated ones, added)
        into its zeroed view of gp.grad.  D: loss_d.backward() sums the gradients of two forward passes (fake, real) whose
        spectral-norm sigma / u / v differ (one power iteration per forward): the second pass accumulates."""
        G, D = self.G, self.D
        self._mark()
        if not G.training:                            # (Module.train() walks the whole tree: 0.4 ms of host time per call)
            G.train()
        with torch.no_grad():
            # Parameter-only work of the step's first three forwards -- G's DO-Conv folds / packs / bf16 splits, and the power
            # iteration + pack + split of D's fake and real pass -- goes to a side stream and runs beside the attention block and
            # the IDW, whose latency-bound kernels leave most of the chip idle (P2I_PREP_OVERLAP=0: inline, as before round 3).
            # D's REAL pass needs nothing of the generator either: its forward and backward follow on the side stream and share the
            # chip with the generator's convolutions (fill their tails); the main stream joins before D's fake pass.
            side = net_fns._side_of(G, frames.device) if (self.prep_overlap and frames.is_cuda) else None
            real_box = [None]
            step_start = ops.mark_stream() if side is not None else None      # the side work depends on the state at the step's start only

            def enqueue_side_work():
                # (called by generator_forward right after the attention block and the IDW are enqueued on the main stream: the
                # GPU is already busy with the step's head while the host issues these ~100 launches)
                b_, t_, c_, h_, w_ = masked.shape
                gprep = side.run(lambda: net_fns.generator_prepare(G, b_, h_, w_), after=step_start)
                ready = side.mark()                   # the main stream waits for the prepared weights, not for the rest of the side work
                if self.use_gan:
                    if not D.training:
                        D.train()
                    dprep_f = side.run(lambda: net_fns.discriminator_prepare(D, tuple(masked.shape), frames.device, False, True, pool=True), after=step_start)
                    dprep_r = side.run(lambda: net_fns.discriminator_prepare(D, tuple(frames.shape), frames.device, False, True, pool=True), after=step_start)

                    def real_pass():
                        # forward AND backward of the real half of the D loss: every GAN loss here is a sum of a real and a fake
                        # term, so d loss_d / d logits_real needs the real logits only (p2i_gan_loss with the real logits in both
                        # slots gives exactly that gradient); the fake half adds its weight gradients later -- two addends, the
                        # same sum in either order
                        lr2, cr2 = net_fns.discriminator_forward(D, frames, need_x=False, need_p=True, pool=True, prep=dprep_r)
                        # (the backward's weight gradients run in line here: a side stream forked from this side stream ends hipGraph
                        # capture in a segmentation fault on ROCm 7.2 -- ops._SIDE_DEPTH, tools/graph_nested_fork.py)
                        _, dlr2, _ = ops.gan_loss_d(lr2, lr2, self.gan_type, self.real_label, self.fake_label)
                        self.dp.zero_grad()
                        net_fns.discriminator_backward(D, cr2, dlr2, need_x=False, inplace=True, accumulate=False)
                        return lr2, None

                    real_box[0] = (dprep_f, side.run(real_pass, frames, after=step_start))
                return gprep, ready

            preds, S = net_fns.generator_forward(G, masked, masks, need_grad=True, after_head=enqueue_side_work if side is not None else None)
            dprep_f, real_out = real_box[0] if real_box[0] is not None else (None, None)
            if side is not None and self.use_gan:
                # the reconstruction loss (three short latency-bound kernels) is needed at the generator's backward only: it queues
                # behind D's real half on the side stream while D's fake forward starts on the main stream right away
                out3, dpred = side.run(lambda: ops.recloss(preds.contiguous(), frames.contiguous().float(), self.rec_loss.k1_alpha), preds, frames)
            else:
                out3, dpred = ops.recloss(preds.contiguous(), frames.contiguous().float(), self.rec_loss.k1_alpha)
            out = {"rec": out3[2], "pool": out3[0], "reg": out3[1]}
            self._mark()
            dgen = dpred
            loss_g = out3[2:3]
            if self.use_gan:
                if not D.training:
                    D.train()
                lf, cf = net_fns.discriminator_forward(D, preds, need_x=False, need_p=True, pool=True, prep=dprep_f)
            if side is not None:
                side.join()                                  # D's real half (and the reconstruction loss) are complete
            if self.use_gan:
                if real_out is not None and real_out[1] is None:      # the real half is done (forward and backward, on the side stream)
                    lr_ = real_out[0]
                    loss_d, _, dlf = ops.gan_loss_d(lr_, lf, self.gan_type, self.real_label, self.fake_label)
                    net_fns.discriminator_backward(D, cf, dlf, need_x=False, inplace=True, accumulate=True)
                    del cf
                else:
                    lr_, cr = real_out if real_out is not None else net_fns.discriminator_forward(D, frames, need_x=False, need_p=True, pool=True)
                    loss_d, dlr, dlf = ops.gan_loss_d(lr_, lf, self.gan_type, self.real_label, self.fake_label)
                    self.dp.zero_grad()
                    net_fns.discriminator_backward(D, cf, dlf, need_x=False, inplace=True, accumulate=False)
                    net_fns.discriminator_backward(D, cr, dlr, need_x=False, inplace=True, accumulate=True)
                    del cf, cr
                if self.distributed:
                    self._exchange("d", lambda: _allreduce_mean(self.dp.grad, self.world))
                self.opt_d.step()
                self._mark()
                lg, cg = net_fns.discriminator_forward(D, preds, need_x=True, need_p=False, pool=True)
                adv, dlg = ops.gan_loss_g(lg, self.adv_weight, self.gan_type, self.real_label)
                # d(rec)/d(preds) rides into the discriminator's first-layer dgrad as its additive term
                dgen, _, _, _ = net_fns.discriminator_backward(D, cg, dlg, need_x=True, needs=[False] * (2 * sum(map(len, D.layers())) + 1), dx_add=dpred)
                del cg
                loss_g = ops.add2(out3[2:3], adv)
                out.update(loss_d=loss_d.reshape(()), adv=adv.reshape(()), logits_real=lr_, logits_fake=lf)
            self.gp.zero_grad()
            if self.distributed and self.dp_overlap:
                bk = BucketedAllReduce(self.gp.grad, self.world, (lambda t, a: ops.axpy_(t, t, a - 1.0)) if self.gp.grad.is_cuda else None)
                net_fns.generator_backward(G, S, dgen, inplace=True, on_level_done=lambda lvl: bk.launch(*self.level_range(lvl)))
                del S
                self._exchange("g", bk.finish)
            else:
                net_fns.generator_backward(G, S, dgen, inplace=True)
                del S
                if self.distributed:
                    self._exchange("g", lambda: _allreduce_mean(self.gp.grad, self.world))
            self.opt_g.step()
            out.update(loss_g=loss_g.reshape(()), preds=preds)
            self._mark()
        return out

    def _step_impl(self, frames, masked, masks) -> Dict[str, torch.Tensor]:
        if self.direct:
            with ops.step_stream():                     # one stream look-up per step instead of one per launch
                return self._step_direct(frames, masked, masks)
        self._mark()
        self.G.train()
        preds = self.G(masked, masks)
        loss_g, parts = self.rec_loss(preds, frames, masks)
        out = {"rec": loss_g.detach(), "pool": parts["pool"].tensor(), "reg": parts["reg"].tensor()}
        self._mark()
        if self.use_gan:
            self.D.train()
            for p in self.dp.params:
                p.requires_grad_(True)
            logits_fake = self.D(preds.detach())
            logits_real = self.D(frames)
            loss_d = discriminator_loss(logits_real, logits_fake, self.gan_type, self.real_label, self.fake_label)
            self.dp.zero_grad()
            loss_d.backward()
            if self.distributed:
                _allreduce_mean(self.dp.grad, self.world)
            self.opt_d.step()
            for p in self.dp.params:
                p.requires_grad_(False)
            self._mark()
            logits_g = self.D(preds)
            adv = generator_adv_loss(logits_g, self.adv_weight, self.gan_type, self.real_label)
            loss_g = loss_g + adv
            out.update(loss_d=loss_d.detach(), adv=adv.detach(), logits_real=logits_real.detach(), logits_fake=logits_fake.detach())
        self.gp.zero_grad()
        loss_g.backward()
        if self.distributed:
            _allreduce_mean(self.gp.grad, self.world)
        self.opt_g.step()
        if self.use_gan:
            for p in self.dp.params:
                p.requires_grad_(True)
        out.update(loss_g=loss_g.detach(), preds=preds.detach())
        self._mark()
        return out

    @torch.no_grad()
    def eval_rec_loss(self, frames, masked, masks) -> torch.Tensor:
        self.G.eval()
        preds = self.G(masked, masks)
        out3, _ = ops.recloss(preds.contiguous(), frames.contiguous(), self.rec_loss.k1_alpha)
        return out3[2]

    def checkpoint(self, epoch: int, global_step: int) -> Dict:
        """Checkpoint dict with the reference's keys (train.py:475-485)."""
        state = {"epoch": epoch, "global_step": global_step,
                 "generator": {k: v.detach().clone() for k, v in self.G.state_dict().items()},
                 "optimizer_g": self.opt_g.state_dict()}
        if self.use_gan:
            state["discriminator"] = {k: v.detach().clone() for k, v in self.D.state_dict().items()}
            state["optimizer_d"] = self.opt_d.state_dict()
        return state
