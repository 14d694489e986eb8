"""Step-indexed scalar schedules (learning rate, loss weights).  API of nnet/schedulers.py:24-137."""
import torch
import torch.nn as nn


class Scheduler(nn.Module):
    def __init__(self):
        super().__init__()
        self.model_step = torch.tensor(0)     # CPU scalar shared with Model / optimizer

    def step(self):
        self.model_step += 1
        return self.get_val()

    def get_val(self):
        return self.get_val_step(self.model_step)

    def get_val_step(self, step):
        return None


class ConstantScheduler(Scheduler):
    def __init__(self, val):
        super().__init__()
        self.val = val

    def get_val_step(self, step):
        return self.val


class NoamDecayScheduler(Scheduler):
    """val = factor * dim^-0.5 * min(step * warmup^-1.5, step^-0.5)"""

    def __init__(self, warmup_steps, dim_decay, val_factor):
        super().__init__()
        self.warmup_steps, self.dim_decay, self.val_factor = warmup_steps, dim_decay, val_factor

    def get_val_step(self, step):
        s = float(step)
        return self.val_factor * self.dim_decay ** -0.5 * min(s * self.warmup_steps ** -1.5, s ** -0.5)


scheduler_dict = {"ConstantScheduler": ConstantScheduler, "NoamDecayScheduler": NoamDecayScheduler}
