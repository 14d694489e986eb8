"""Training / evaluation harness (API of nnet/model.py).  Host-side Python like the reference; what differs is what a step launches:
the forward/backward are fused HIP sequences, the optimizer is one launch over the flat arena, and data parallelism is one process per
GPU exchanging the flat gradient buffer (and BatchNorm statistics) over RCCL -- no DistributedDataParallel wrapper, no per-step buffer
broadcast."""
import os
import time

import torch
import torch.nn as nn

from .. import ops
from .. import runtime as rt
from .decoders import decoder_dict
from .losses import loss_dict
from .metrics import metric_dict
from .module import Module
from .normalizations import SyncBatchNorm
from .optimizers import optim_dict
from .schedulers import ConstantScheduler, Scheduler


class Model(Module):
    def __init__(self, name="model"):
        super().__init__()
        self.is_distributed, self.rank, self.is_parallel = False, 0, False
        self.compiled, self.built = False, False
        self.name = name
        self.ema_model, self.ema_tau, self.grad_max_norm = None, 0.0, None
        self.arena = None
        self.world_size = 1

    @staticmethod
    def _pre_forward(model, _args):
        """before every forward pass: in-place weight edits -> stale shadows; fp8 operands -> fresh activation amax slots"""
        if model.arena is None:
            return None
        model.arena.check_versions()
        if getattr(model.arena, "_fp8", None) is not None:
            model.arena._fp8.begin_pass()
        return None

    # -- placement ------------------------------------------------------------------------------
    def to(self, device):
        out = super().to(device)
        if self.device.type == "cuda":
            self.arena = rt.ParamArena(self)
            if not getattr(self, "_avec_version_hook", False):       # every forward (training, evaluation, a bare model(x)): weights edited in place since the last pass -> refresh the shadows
                self._avec_version_hook = True
                self.register_forward_pre_hook(Model._pre_forward)
            if self.compiled and hasattr(self.optimizer, "attach_arena"):
                self.optimizer.attach_arena(self.arena)
        return out

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) else device)))

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        if self.arena is not None:
            self.arena.mark_dirty()
        return out

    def distribute_strategy(self, rank, sync_batch_norm=True):
        """nnet/model.py:59-65.  Parameters/buffers are broadcast once from rank 0; afterwards each step all-reduces the flat gradient
        arena (and, with sync_batch_norm, the BatchNorm statistic vectors)."""
        import torch.distributed as dist
        if sync_batch_norm:
            SyncBatchNorm.convert_sync_batchnorm(self)
        self.rank, self.is_distributed = rank, True
        self.world_size = dist.get_world_size()
        rt.ensure_branch_group()          # second communicator for the collectives of the audio branch's stream (runtime.collective_group)
        if sync_batch_norm and self.device.type == "cuda":
            from .. import peer
            peer.setup(self.device)       # SyncBatchNorm statistics by peer writes over xGMI (one small kernel per exchange, no host work); None -> torch.distributed collectives
        if self.arena is not None:
            dist.broadcast(self.arena.master, 0)
            self.arena.mark_dirty()
        else:
            for p in self.parameters():
                dist.broadcast(p.data, 0)
        for b in self.buffers():
            if b.is_floating_point() or b.dtype == torch.int64:
                dist.broadcast(b, 0)

    def parallel_strategy(self):
        raise RuntimeError("single-process DataParallel is a legacy path of the reference (nnet/model.py:67-69); use one process per GPU")

    # -- compile / build (nnet/model.py:80-225) --------------------------------------------------
    def compile(self, losses, loss_weights=None, optimizer="Adam", metrics=None, decoders=None):
        self.optimizer = optim_dict[optimizer](params=self.parameters()) if isinstance(optimizer, str) else optimizer
        self.model_step = self.optimizer.scheduler.model_step
        self.compiled_losses = loss_dict[losses]() if isinstance(losses, str) else ([] if losses is None else losses)
        if loss_weights is None:
            self.compiled_loss_weights = ConstantScheduler(1.0)
        elif isinstance(loss_weights, float):
            self.compiled_loss_weights = ConstantScheduler(loss_weights)
        else:
            assert isinstance(loss_weights, (dict, list))
            it = loss_weights.items() if isinstance(loss_weights, dict) else enumerate(loss_weights)
            for k, v in list(it):
                if not isinstance(v, Scheduler):
                    loss_weights[k] = ConstantScheduler(v)
            self.compiled_loss_weights = loss_weights
        self.compiled_metrics = metric_dict[metrics]() if isinstance(metrics, str) else ([] if metrics is None else metrics)
        self.compiled_decoders = decoder_dict[decoders]() if isinstance(decoders, str) else ([] if decoders is None else decoders)
        self.compiled = True
        if self.arena is not None and hasattr(self.optimizer, "attach_arena"):
            self.optimizer.attach_arena(self.arena)

    def build(self, outputs):
        self.losses = self.map_to_outputs(outputs, self.compiled_losses)
        self.loss_weights = self.map_to_outputs(outputs, self.compiled_loss_weights)
        self.decoders = self.map_to_outputs(outputs, self.compiled_decoders)
        self.metrics = self.map_to_outputs(outputs, self.compiled_metrics)
        self.built = True

    def map_to_outputs(self, outputs, struct):
        """dict -> fill missing keys with None; list -> positional over the output order; single item -> every output."""
        if struct is None:
            return struct
        if isinstance(struct, dict):
            for key in struct:
                if key not in outputs:
                    raise Exception("Found unexpected dict key: {}. Valid output names are: {}".format(key, outputs.keys()))
            for key in outputs:
                struct.setdefault(key, None)
            return struct
        if isinstance(struct, list):
            return {key: (struct[i] if i < len(struct) else None) for i, key in enumerate(outputs)}
        return {key: struct for key in outputs}

    def _fused_ctc_losses(self, outputs, targets):
        """The CTC heads of an InterCTC model share the batch and the labels: when they all fit the LDS-resident kernel they are evaluated by ONE launch
        (ops.CTCLossMultiFn) instead of one 32-workgroup launch per head.  Returns ({key: loss} for the heads it covered (same values as losses.CTCLoss; for logging), their weighted sum (differentiable)) or ({}, None)."""
        from .losses import CTCLoss
        keys = [k for k in outputs if isinstance(self.losses.get(k), CTCLoss) and isinstance(outputs[k], (list, tuple)) and len(outputs[k]) == 2
                and torch.is_tensor(outputs[k][0]) and outputs[k][0].is_cuda and outputs[k][0].dim() == 3]
        if len(keys) < 2 or len(keys) > 8:
            return {}, None
        l0, t0 = self.losses[keys[0]], targets[keys[0]]
        same = all(self.losses[k].blank == l0.blank and self.losses[k].zero_infinity == l0.zero_infinity and self.losses[k].reduction == "mean" and not self.losses[k].assert_shorter
                   and targets[k][0] is t0[0] and targets[k][1] is t0[1] and outputs[k][0].shape[0] == outputs[keys[0]][0].shape[0]
                   and outputs[k][0].shape[2] == outputs[keys[0]][0].shape[2] for k in keys)
        y, y_len = t0
        Lmax = y.shape[1] if y.dim() == 2 else y.numel() // outputs[keys[0]][0].shape[0]
        if not same or not all(ops.ctc_multi_fits(outputs[k][0].shape[1], outputs[k][0].shape[2], Lmax) for k in keys):
            return {}, None
        flat = []
        for k in keys:
            flat += [outputs[k][0], outputs[k][1]]
        weights = tuple(self.loss_weights[k].get_val_step(self.model_step + 1) for k in keys)
        res = ops.CTCLossMultiFn.apply(l0.blank, l0.zero_infinity, y, y_len, weights, *flat)
        return dict(zip(keys, res[1:])), res[0]

    # -- one forward + losses (nnet/model.py:227-344) ---------------------------------------------
    def forward_model(self, inputs, targets, compute_metrics=True, verbose=0):
        batch_losses, batch_metrics, batch_truths, batch_preds = {}, {}, {}, {}
        total_loss = None                        # (no zero-filled accumulator: with the fused CTC heads the weighted total comes out of their launch as is)
        Model._pre_forward(self, None)       # self.forward() is called directly (no __call__): the pre-forward hook would not fire on the training / evaluation paths
        outputs = self.forward(inputs)
        if isinstance(outputs, list):
            outputs = {"output_" + str(k): v for k, v in enumerate(outputs)}
        elif not isinstance(outputs, dict):
            outputs = {"output": outputs}
        targets = self.map_to_outputs(outputs, targets)
        if not self.built:
            self.build(outputs)
        fused, fused_total = self._fused_ctc_losses(outputs, targets)
        if fused_total is not None:
            total_loss = fused_total
        for key in outputs:
            if self.losses[key] is not None:
                if key in fused:                     # already inside fused_total with its weight
                    batch_losses["loss_" + key] = fused[key]
                else:
                    l = self.losses[key](targets[key], outputs[key])
                    batch_losses["loss_" + key] = l
                    wl = l * self.loss_weights[key].get_val_step(self.model_step + 1)
                    total_loss = wl if total_loss is None else total_loss + wl
            if compute_metrics and self.metrics and self.metrics[key] is not None:
                metric, decoder = self.metrics[key], (self.decoders[key] if self.decoders else None)
                name = metric.name if metric.name not in batch_metrics else metric.name + "_" + key
                if decoder is not None:
                    batch_truths[name] = decoder(targets[key], from_logits=False) if targets[key] is not None else None
                    batch_preds[name] = decoder(outputs[key])
                else:
                    batch_truths[name], batch_preds[name] = targets[key], outputs[key]
                batch_metrics[name] = metric(batch_truths[name], batch_preds[name])
        for module in self.modules():
            if getattr(module, "added_losses", None):
                for key, value in module.added_losses.items():
                    batch_losses["loss_" + key] = value["loss"]
                    wl = value["loss"] * value["weight"]
                    total_loss = wl if total_loss is None else total_loss + wl
                module.reset_losses()
        if total_loss is None:
            total_loss = torch.zeros((), device=self.device)
        batch_losses = dict({"loss": total_loss}, **batch_losses) if len(batch_losses) > 1 else {"loss": total_loss}
        return batch_losses, batch_metrics, batch_truths, batch_preds

    # -- one optimisation micro-step (nnet/model.py:346-409) ---------------------------------------
    def train_step(self, inputs, targets, precision=torch.float32, grad_scaler=None, accumulated_steps=1, acc_step=0, eval_training=False):
        rt.set_compute_dtype(precision)
        rt.reset_zero_pool(self.device)
        if self.is_distributed and self.arena is not None:       # overlap part of the gradient exchange with the backward pass of the last micro-step
            self.arena.arm_early_all_reduce(acc_step + 1 >= accumulated_steps and os.environ.get("AVEC_EARLY_ALLREDUCE", "1") != "0")
        batch_losses, batch_metrics, _, _ = self.forward_model(inputs, targets, compute_metrics=eval_training)
        (batch_losses["loss"] / accumulated_steps).backward()
        rt.advance_rng(self.device)
        batch_losses = self._own_losses(batch_losses)
        acc_step += 1
        if acc_step < accumulated_steps:
            return batch_losses, batch_metrics, acc_step
        if self.is_distributed:
            # the reference all-reduces on every micro-batch (no no_sync); summing once per optimizer step is numerically the same
            self.arena.all_reduce_grads()
            self.optimizer.grad_scale = 1.0 / self.world_size
        if self.grad_max_norm is not None:
            self.add_info("grad_norm", self.clip_gradients(self.grad_max_norm))
        self.optimizer.step()
        self.optimizer.zero_grad()
        self.add_info("lr", float(self.optimizer.param_groups[0]["lr"]))
        self.add_info("step", int(self.model_step))
        return batch_losses, batch_metrics, 0

    def clip_gradients(self, max_norm):
        """global-norm clipping of the flat gradient arena (torch.nn.utils.clip_grad_norm_ semantics, nnet/model.py:381-383); returns the norm before clipping"""
        norm = self.arena.grad.norm() * getattr(self.optimizer, "grad_scale", 1.0)
        self.arena.grad.mul_((max_norm / (norm + 1e-6)).clamp(max=1.0))
        return norm

    def make_graphed_train_step(self, inputs, targets, precision=torch.bfloat16, warmup=2):
        """Capture ONE optimisation step into a hipGraph: ~1600 launches replay from a single submission, removing the host-side launch overhead.
        Static shapes: later batches are copied into the captured buffers.  Returns step(inputs, targets) -> losses dict (device scalars).
        * single process: shadow refresh, forward, 6 losses, backward and Adam are all inside the graph;
        * data parallel (one process per GPU): the graph holds shadow refresh + forward + losses + backward -- the SyncBatchNorm statistic exchanges are
          peer-write kernels (avec_amd/peer.py), so no host-driven collective interrupts it; the gradient all-reduce (RCCL) and the Adam launch follow the
          replay.  Requires the peer exchange (otherwise every BatchNorm layer needs a host-issued collective: use train_step)."""
        from .. import peer
        dist_mode = self.is_distributed
        # RCCL collectives are capturable: over the nccl backend the WHOLE data-parallel step -- SyncBatchNorm exchanges (peer writes, or RCCL when the peer exchange is
        # off / refused), the range-wise gradient all-reduce (fusion + audio-visual ranges when the gradient crosses the fusion inputs, the audio encoder's from the audio
        # branch's stream beside the visual backward, the visual encoder's at the end) and the Adam launch -- is ONE hipGraph.  AVEC_GRAPH_ALLREDUCE=0: all-reduce and
        # Adam follow the replay (the round-3 behaviour; also what gloo debugging runs use).
        in_graph = dist_mode and torch.distributed.get_backend() == "nccl" and os.environ.get("AVEC_GRAPH_ALLREDUCE", "1") != "0"
        assert not dist_mode or in_graph or peer.active() is not None, "graph capture of a data-parallel step needs capturable SyncBatchNorm exchanges (RCCL, or the peer exchange); use train_step"
        rt.set_compute_dtype(precision)
        warmup = max(int(warmup), 1)                        # lazily created constants (DFT matrix, sinusoid tables, loss weights) must exist before the capture
        static_in = [t.clone() for t in inputs]
        static_tg = tuple(t.clone() for t in targets)
        if dist_mode:
            self.arena.arm_early_all_reduce(False)          # (armed per pass inside body() when the collectives are captured)
        state = {"in_graph": in_graph}

        def body():
            rt.reset_zero_pool(self.device)
            if dist_mode and state["in_graph"]:
                self.arena.arm_early_all_reduce(os.environ.get("AVEC_EARLY_ALLREDUCE", "1") != "0", sync=True)
            losses, _, _, _ = self.forward_model(static_in, static_tg, compute_metrics=False)
            ops.stamp("loss_done:f")
            losses["loss"].backward()
            ops.stamp("bwd_done:b")
            rt.advance_rng(self.device)
            if not dist_mode:
                self.optimizer.launch_step()
            elif state["in_graph"]:
                self.arena._sync_collectives = True
                self.arena.all_reduce_grads()
                self.optimizer.grad_scale = 1.0 / self.world_size
                self.optimizer.launch_step()
            return losses

        def finish():                                       # data parallel without captured collectives: average the gradients, then the optimizer launch
            if dist_mode and not state["in_graph"]:
                self.arena.all_reduce_grads()
                self.optimizer.grad_scale = 1.0 / self.world_size
                self.optimizer.launch_step()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        warm = None
        with torch.cuda.stream(side):
            for _ in range(warmup):             # warm-up on a side stream: lazy kernel attributes, caches, allocator
                self.optimizer.prepare_step()
                warm = body()
                finish()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # data parallel: the process group's watchdog thread polls the events of collectives it still tracks (hipEventQuery, every ~100 ms); under a "global"-mode capture
        # such a poll is a capture violation and aborts the process, and in "thread_local" mode HIP reports the backward pass's collectives (issued from the autograd
        # thread) as not capturing, so they ARE tracked and their captured events get polled.  So: global mode, and the watchdog's list drained before the capture starts
        # (everything issued so far has completed -- synchronize above -- and is reaped within one watchdog period).
        gkw = {}
        if dist_mode and torch.distributed.get_backend() == "nccl":
            torch.cuda.synchronize()
            time.sleep(0.6)
        cap_err = None
        try:
            with torch.cuda.graph(graph, **gkw):        # (the optimizer's device-side {step, lr} pair exists since the warm-up; each replay is preceded by prepare_step)
                static_losses = body()
        except Exception as e:
            if not (dist_mode and state["in_graph"]):
                raise
            cap_err = e
        if dist_mode and state["in_graph"]:
            # the fallback must be the SAME on every rank: a rank that keeps captured collectives beside one that issues them eagerly is a collective mismatch
            flag = torch.tensor([0.0 if cap_err is None else 1.0], device=self.device)
            torch.cuda.synchronize()
            torch.distributed.all_reduce(flag)
            torch.cuda.synchronize()
            if cap_err is None and float(flag.item()) > 0:
                cap_err = RuntimeError("another rank could not capture the collectives")
        if cap_err is not None:
            e = cap_err
            ops.reset_backward_state()                  # the aborted capture left a half-run backward pass behind: queued weight gradients, hand-over tags, group counters
            # the collectives could not be captured on this stack: capture forward + backward only (peer-write SyncBatchNorm), all-reduce + Adam after each replay
            print("[avec_amd] rank %d: capture with in-graph RCCL collectives failed (%s: %s); capturing forward + backward only" % (self.rank, type(e).__name__, e), flush=True)
            state["in_graph"] = False
            self.arena.arm_early_all_reduce(False)
            torch.cuda.synchronize()
            time.sleep(0.6)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, **gkw):
                static_losses = body()

        replays = [0]

        def step(new_inputs=None, new_targets=None):
            replays[0] += 1
            px = peer.active() if dist_mode else None       # None: SyncBatchNorm over RCCL (peer exchange off / refused / more than one node)
            if px is not None and replays[0] % 100 == 0:
                px.check()                          # (synchronises) a lost rank: the guarded Adam launch skipped those steps, say so instead of training on
            if new_inputs is not None:
                for d, s_ in zip(static_in, new_inputs):
                    d.copy_(s_, non_blocking=True)
                for d, s_ in zip(static_tg, new_targets):
                    d.copy_(s_, non_blocking=True)
            self.optimizer.prepare_step()
            graph.replay()
            finish()
            if not dist_mode or state["in_graph"]:
                self.arena.mark_dirty()             # host-side view of what the replay's closing Adam launch left behind: whatever runs next OUTSIDE the graph
            return static_losses                    # (evaluation, an eager step) must refresh the weight shadows first -- the captured refresh sits at the START of a replay

        if dist_mode:
            self.arena.arm_early_all_reduce(os.environ.get("AVEC_EARLY_ALLREDUCE", "1") != "0")      # later eager train_steps keep their overlapped exchange
        step.graph = graph
        step.collectives_in_graph = bool(dist_mode and state["in_graph"])
        step.warm_losses = {k: v.detach().clone() for k, v in warm.items()} if warm is not None else None
        return step

    # -- hipGraph replay for ragged training batches: one captured step per batch SHAPE --------------------------------------------------------------------
    @staticmethod
    def pad_av_batch(inputs, targets, bucket_frames):
        """zero-pad an audio-visual batch [video (B,Tv,H,W,C), video_len, audio (B,Ta), audio_len], (labels (B,L), label_len) to the next multiple of
        `bucket_frames` video frames (audio to the longest clip with that many frames: 640 * Tv' - 1 samples, labels to a multiple of 8): the true lengths stay, so
        attention masks and CTC are unchanged; like the reference's own padding to the batch maximum (nnet/collate_fn.py:143-146) the zero frames do enter the
        BatchNorm batch statistics -- a coarser bucket is a (slightly) different batch composition, which is why bucketing is opt-in."""
        video, vlen, audio, alen = inputs
        labels, llen = targets
        Tv = video.shape[1]
        Tvp = (Tv + bucket_frames - 1) // bucket_frames * bucket_frames
        Tap = 640 * Tvp - 1
        if Tvp > Tv:
            video = torch.nn.functional.pad(video, (0, 0, 0, 0, 0, 0, 0, Tvp - Tv))
        if Tap > audio.shape[1]:
            audio = torch.nn.functional.pad(audio, (0, Tap - audio.shape[1]))
        Lp = (labels.shape[1] + 7) // 8 * 8
        if Lp > labels.shape[1]:
            labels = torch.nn.functional.pad(labels, (0, Lp - labels.shape[1]))
        return [video, vlen, audio, alen], (labels, llen)

    def graphed_train_step(self, inputs, targets, precision=torch.bfloat16, cache_size=8, bucket_frames=None):
        """train_step through a cache of captured steps keyed by the batch shape (LRU, `cache_size` graphs: each holds its own activation pool).  Ragged LRS batches
        (nnet/collate_fn.py pads to the batch maximum) repeat a small set of shapes once they are bucketed (`bucket_frames`, see pad_av_batch) or come from a
        length-bucketed sampler.  Falls back to train_step when a step cannot be captured as is (scheduled loss weights, gradient clipping, data parallel without the
        peer exchange)."""
        from .schedulers import ConstantScheduler
        from .. import peer
        lw = self.compiled_loss_weights
        const_w = isinstance(lw, ConstantScheduler) or (isinstance(lw, dict) and all(isinstance(v, ConstantScheduler) for v in lw.values())) \
            or (isinstance(lw, list) and all(isinstance(v, ConstantScheduler) for v in lw))
        capturable = (not self.is_distributed) or peer.active() is not None or \
            (torch.distributed.get_backend() == "nccl" and os.environ.get("AVEC_GRAPH_ALLREDUCE", "1") != "0")
        if not const_w or self.grad_max_norm is not None or not capturable:
            return self.train_step(inputs, targets, precision=precision)[0]
        if bucket_frames and len(inputs) == 4 and inputs[0].dim() == 5 and isinstance(targets, (tuple, list)) and len(targets) == 2:
            inputs, targets = self.pad_av_batch(inputs, targets, bucket_frames)
        key = tuple((tuple(t.shape), str(t.dtype)) for t in list(inputs) + list(targets)) + (str(precision),)
        cache = self.__dict__.setdefault("_graph_cache", {})
        step = cache.pop(key, None)
        if step is None:
            while len(cache) >= cache_size:
                cache.pop(next(iter(cache)))                 # least recently used (dict order = recency)
                ev = self.__dict__["_graph_evictions"] = self.__dict__.get("_graph_evictions", 0) + 1
                if ev == 2 * cache_size:                     # every new shape costs an eager warm-up step, a capture and a private activation pool
                    import warnings
                    warnings.warn("graphed_train_step: %d captured steps evicted from a cache of %d -- the batch shapes do not repeat; bucket the batches "
                                  "(graph_bucket_frames / a length-bucketed sampler, nnet/samplers.py) or raise graph_cache_size" % (ev, cache_size))
            step = self.make_graphed_train_step(inputs, targets, precision=precision, warmup=1)      # (the warm-up pass is a real optimisation step on this batch)
            cache[key] = step
            return step.warm_losses
        cache[key] = step
        return step(inputs, targets)

    @staticmethod
    def _own_losses(batch_losses):
        """the fused CTC launch leaves its means and the weighted total in the step's pre-zeroed scratch pool, which the next step (or graph replay) overwrites: hand
        the caller values it can keep (ONE stacked copy; a captured step returns the pool views themselves -- static outputs refilled by every replay)"""
        keys = list(batch_losses)
        vals = torch.stack([batch_losses[k].detach().reshape(()).float() for k in keys]).unbind(0)
        return dict(zip(keys, vals))

    def eval_step(self, inputs, targets, verbose=0):
        with torch.no_grad():
            rt.reset_zero_pool(self.device, create=False)      # (the scratch pool hands out PRE-ZEROED accumulators: whatever earlier steps or replays of other shapes left there must go; an evaluation-only process never allocates it)
            batch_losses, batch_metrics, batch_truths, batch_preds = self.forward_model(inputs, targets, verbose=verbose)
            return self._own_losses(batch_losses), batch_metrics, batch_truths, batch_preds

    # -- checkpoints (nnet/model.py:499-544) -------------------------------------------------------
    def save(self, path, save_optimizer=True):
        if self.is_distributed:
            from .. import peer
            if peer.active() is not None:
                peer.active().check()            # (synchronises) a SyncBatchNorm exchange that lost a rank since the last check: raise instead of writing a checkpoint of skipped steps
        torch.save({"model_state_dict": self.state_dict(), "optimizer_state_dict": self.optimizer.state_dict() if save_optimizer else None,
                    "model_step": self.model_step, "is_distributed": self.is_distributed or self.is_parallel,
                    "ema_model_state_dict": None, "grad_scaler_state_dict": None}, path)

    def load(self, path, load_optimizer=True, verbose=True, strict=True):
        ckpt = torch.load(path, map_location=self.device, weights_only=False)
        sd = ckpt["model_state_dict"]
        if ckpt.get("is_distributed", False):
            sd = {k.replace(".module.", ".").replace("module.", "", 1) if k.startswith("module.") else k.replace(".module.", "."): v for k, v in sd.items()}
        self.load_state_dict({k: v for k, v in sd.items()}, strict=strict)
        if load_optimizer and ckpt.get("optimizer_state_dict") is not None:
            # the step counter (Noam schedule, loss-weight schedules, Adam bias correction) travels with the optimizer state (nnet/model.py:527-536):
            # fine-tuning from a checkpoint without its optimizer restarts the schedule at step 0 on zero moments
            self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
            self.model_step.fill_(ckpt["model_step"])

    # -- loops (compact counterparts of nnet/model.py:668-942, 1047-1077) ----------------------------
    def fit(self, dataset_train, epochs, dataset_eval=None, eval_steps=None, verbose_eval=0, initial_epoch=0, callback_path=None, steps_per_epoch=None,
            precision=torch.float32, accumulated_steps=1, eval_period_step=None, eval_period_epoch=1, saving_period_epoch=1, log_figure_period_step=None,
            log_figure_period_epoch=1, step_log_period=100, eval_training=True, grad_init_scale=65536.0, detect_anomaly=False, recompute_metrics=False,
            wandb_logging=False, verbose_progress_bar=1, keep_last_k=None, use_graphs=False, graph_bucket_frames=None, graph_cache_size=8):
        """use_graphs: replay captured steps per batch shape (graphed_train_step) instead of eager launches; graph_bucket_frames: additionally zero-pad AV batches to
        multiples of that many video frames so that few shapes occur (opt-in: changes the padding the BatchNorm statistics see)."""
        if callback_path is not None and self.rank == 0:
            os.makedirs(callback_path, exist_ok=True)
        for epoch in range(initial_epoch, epochs):
            self.train()
            sampler = getattr(dataset_train, "sampler", None)
            if self.is_distributed and hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)          # reshuffle + re-shard per epoch (nnet/model.py:709-710)
            bsamp = getattr(dataset_train, "batch_sampler", None)
            if hasattr(bsamp, "set_epoch"):
                bsamp.set_epoch(epoch)            # length-bucketed batches: new windows every epoch (nnet/samplers.py)
            acc_step, t0, n = 0, time.time(), 0
            for step, batch in enumerate(dataset_train):
                inputs = self.transfer_to_device(batch["inputs"])
                targets = self.transfer_to_device(batch["targets"])
                if use_graphs and accumulated_steps == 1 and self.device.type == "cuda":
                    losses = self.graphed_train_step(inputs, targets, precision=precision, cache_size=graph_cache_size, bucket_frames=graph_bucket_frames)
                else:
                    losses, _, acc_step = self.train_step(inputs, targets, precision, None, accumulated_steps, acc_step, eval_training)
                n += 1
                if self.is_distributed and step % step_log_period == 0:
                    from .. import peer
                    if peer.active() is not None:
                        peer.active().check()            # a SyncBatchNorm peer exchange that lost a rank raises here (its sums were NaN from that step on)
                if self.rank == 0 and step % step_log_period == 0:
                    print("epoch %d step %d model_step %d loss %.4f" % (epoch + 1, step, int(self.model_step), float(losses["loss"].detach())))
                if steps_per_epoch is not None and step + 1 >= steps_per_epoch:
                    break
            if self.rank == 0:
                print("epoch %d: %d steps in %.1fs" % (epoch + 1, n, time.time() - t0))
                if callback_path is not None and (epoch + 1) % saving_period_epoch == 0:
                    self.save(os.path.join(callback_path, "checkpoints_epoch_{}_step_{}.ckpt".format(epoch + 1, int(self.model_step))))
            if dataset_eval is not None and (epoch + 1) % eval_period_epoch == 0:
                self.evaluate(dataset_eval, eval_steps)
            if self.is_distributed:
                torch.distributed.barrier()       # ranks leave the epoch together (rank 0 wrote the checkpoint)

    def evaluate(self, dataset_eval, eval_steps=None, verbose=0, eval_loss=True, recompute_metrics=False):
        if isinstance(dataset_eval, (list, tuple)):          # several evaluation sets (the reference configs list LRS2 and LRS3 test sets)
            res = [self.evaluate(d, eval_steps, verbose, eval_loss, recompute_metrics) for d in dataset_eval]
            return res[0] if len(res) == 1 else res
        self.eval()
        sums, count = {}, 0
        metric_keys, truths, preds = [], {}, {}
        for step, batch in enumerate(dataset_eval):
            inputs = self.transfer_to_device(batch["inputs"])
            targets = self.transfer_to_device(batch["targets"])
            losses, metrics, batch_truths, batch_preds = self.eval_step(inputs, targets, verbose)
            for k, v in list(losses.items()) + list(metrics.items()):
                sums[k] = sums.get(k, 0.0) + float(v)
            metric_keys = list(metrics.keys())
            if recompute_metrics:                           # keep the decoded references / hypotheses: corpus-level metrics (nnet/model.py:899-903, 928-931)
                for k in metric_keys:
                    if isinstance(batch_truths.get(k), (list, tuple)) and isinstance(batch_preds.get(k), (list, tuple)):
                        truths.setdefault(k, []).extend(batch_truths[k])
                        preds.setdefault(k, []).extend(batch_preds[k])
            count += 1
            if eval_steps is not None and step + 1 >= eval_steps:
                break
        if self.is_distributed:                             # sum over ranks (nnet/model.py:912-918: reduce_losses_metrics / gather_truths_preds)
            import torch.distributed as dist
            # every collective below must be issued by every rank with the same shapes, whatever this rank saw (zero batches, no list-type hypotheses):
            # agree on the key sets first
            all_keys = [None] * dist.get_world_size()
            dist.all_gather_object(all_keys, (sorted(sums), bool(truths)))
            keys = sorted(set(k for ks, _ in all_keys for k in ks))
            for k in keys:
                sums.setdefault(k, 0.0)
            any_truths = any(t for _, t in all_keys)
            vec = torch.tensor([sums[k] for k in keys] + [float(count)], dtype=torch.float64, device=self.device if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(vec)
            sums, count = {k: float(v) for k, v in zip(keys, vec[:-1].tolist())}, int(vec[-1].item())
            if any_truths:
                gathered = [None] * dist.get_world_size()
                dist.all_gather_object(gathered, (truths, preds))
                truths, preds = {}, {}
                for t, p in gathered:
                    for k in t:
                        truths.setdefault(k, []).extend(t[k])
                        preds.setdefault(k, []).extend(p[k])
        res = {k: v / max(count, 1) for k, v in sums.items()}
        for k in truths:                                    # metric of the whole evaluation set, not the mean of per-batch values
            cands = [mt for mt in (self.metrics.values() if isinstance(self.metrics, dict) else [self.metrics]) if mt is not None]
            metric = next((mt for mt in cands if mt.name == k), None) or next((mt for mt in cands if k.startswith(mt.name + "_")), None)
            if metric is not None:
                res[k] = float(metric(truths[k], preds[k]))
        return res

    def eval_time(self, dataset_eval, eval_steps=None, **kwargs):
        if isinstance(dataset_eval, (list, tuple)):
            return sum(self.eval_time(d, eval_steps, **kwargs) for d in dataset_eval)
        self.eval()
        torch.cuda.synchronize()
        t0 = time.time()
        for step, batch in enumerate(dataset_eval):
            with torch.no_grad():
                self.forward(self.transfer_to_device(batch["inputs"]))
            if eval_steps is not None and step + 1 >= eval_steps:
                break
        torch.cuda.synchronize()
        return time.time() - t0

    def summary(self, show_dict=False):
        print(self.name, "Parameters:", sum(p.numel() for p in self.parameters()))
