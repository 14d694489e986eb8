"""Optimizers (nnet/optimizers.py).  Adam keeps torch.optim.Adam's state_dict format (+ "model_step") but steps the whole model with ONE
HIP launch over the flat fp32 arenas (the reference's single-tensor path issues ~15k ATen ops per step)."""
import torch
import torch.optim as optim

from .. import runtime as rt
from ..lib import lib
from . import schedulers


class Adam(optim.Adam):
    """nnet/optimizers.py:61-93: lr read from the scheduler before every update (model_step incremented first), coupled L2 weight decay."""

    def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-08, weight_decay=0, amsgrad=False):
        assert not amsgrad
        super().__init__(params=params, lr=0.0, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False)
        self.scheduler = lr if isinstance(lr, schedulers.Scheduler) else schedulers.ConstantScheduler(val=lr)
        self.arena = None
        self.grad_scale = 1.0
        self._flat = None

    # -- flat state -------------------------------------------------------------------------
    def attach_arena(self, arena):
        """Bind to the model's ParamArena: moments become two flat fp32 buffers laid out like the master arena."""
        self.arena = arena
        dev = arena.master.device
        self._flat = {"exp_avg": torch.zeros_like(arena.master), "exp_avg_sq": torch.zeros_like(arena.master),
                      "state": torch.zeros(2, dtype=torch.float32, device=dev)}
        self._step_t = torch.tensor(0.0)
        for p, o in zip(arena.params, arena.offsets):
            st = self.state[p]
            n = p.numel()
            st["step"] = self._step_t
            st["exp_avg"] = self._flat["exp_avg"][o:o + n].as_strided(p.shape, p.stride())
            st["exp_avg_sq"] = self._flat["exp_avg_sq"][o:o + n].as_strided(p.shape, p.stride())

    def prepare_step(self):
        """host half of step(): advance the scheduler, publish {step, lr} to the device scalar pair the kernel reads"""
        lr = self.scheduler.step()
        for group in self.param_groups:
            group["lr"] = lr
        if self.arena is None:
            raise RuntimeError("nnet.Adam steps a model's flat parameter arena on the GPU: move the model with Model.to('cuda') first "
                               "(there is no per-tensor / CPU optimizer path)")
        # ring of pinned {step, lr} slots: the H2D copy of step N is asynchronous, so the host may already be preparing step N+1 (graph replays,
        # eager steps never sync) -- it then writes a DIFFERENT slot; a slot is reused only after its copy's event completed
        if "host_ring" not in self._flat:
            self._flat["host_ring"] = [(torch.zeros(2, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(8)]
            self._flat["host_pos"] = 0
        slot, ev = self._flat["host_ring"][self._flat["host_pos"] % 8]
        if self._flat["host_pos"] >= 8:
            ev.synchronize()
        self._flat["host_pos"] += 1
        slot[0] = float(self.scheduler.model_step)
        slot[1] = float(lr)
        self._flat["state"].copy_(slot, non_blocking=True)
        ev.record()

    def launch_step(self):
        """device half: ONE kernel over the flat arenas (graph-capturable)"""
        g = self.param_groups[0]
        self._launch(g)
        self.arena.mark_dirty()

    @torch.no_grad()
    def step(self, closure=None):
        self.prepare_step()
        self.launch_step()
        self._step_t.fill_(float(self.scheduler.model_step))     # one shared CPU scalar stands for every per-parameter "step"
        return None

    def _launch(self, g):
        from .. import peer
        px = peer.active()            # data parallel with the peer exchange: its device error flag guards the update (a lost rank -> NaN sums -> skipped step)
        lib.adam_step_guarded(self.arena.master.data_ptr(), self.arena.grad.data_ptr(), self._flat["exp_avg"].data_ptr(), self._flat["exp_avg_sq"].data_ptr(),
                              self._flat["state"].data_ptr(), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.grad_scale, 1,
                              self.arena.numel, px.err.data_ptr() if px is not None else None, rt.stream())

    def zero_grad(self, set_to_none=False):
        if self.arena is not None:
            return            # the Adam kernel clears the gradient arena in the same pass
        super().zero_grad(set_to_none=False)

    def state_dict(self):
        sd = super().state_dict()
        sd["model_step"] = self.scheduler.model_step
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        self.scheduler.model_step.fill_(state_dict.pop("model_step"))
        super().load_state_dict(state_dict)
        if self.arena is not None:      # re-home the loaded moments into the flat buffers
            loaded = {p: dict(self.state[p]) for p in self.arena.params if p in self.state}
            self.attach_arena(self.arena)
            for p, st in loaded.items():
                if "exp_avg" in st:
                    self.state[p]["exp_avg"].copy_(st["exp_avg"])
                    self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])


optim_dict = {"Adam": Adam}
