"""Adam over ONE flat fp32 buffer.

All parameters are re-pointed (as views) into a single allocation, and so are their
gradients and both moments: the optimiser step is one kernel launch, zero_grad is one
memset, and the flat gradient buffer is exactly what the data-parallel all-reduce
moves over RCCL (qarig.parallel).  Semantics and state_dict layout follow
torch.optim.Adam as the reference uses it (betas=(0.5, 0.999), eps 1e-8, no weight
decay; train_quantized_transformer.py:317-334), so "model_optimizer" entries of
reference checkpoints load and save unchanged.
"""
import math

import torch

from . import ops


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs the parameters on the GPU (call model.to(device) first)")
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False)
        self.param_groups = [dict(self.defaults, params=list(range(len(self.params))))]
        sizes = [p.numel() for p in self.params]
        # 16-B aligned slices so every view can be read with float4 loads
        offs, total = [], 0
        for s in sizes:
            offs.append(total)
            total += (s + 3) // 4 * 4
        self.offsets, self.total = offs, total
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_count = 0
        with torch.no_grad():
            for p, o, s in zip(self.params, offs, sizes):
                view = self.flat_param[o:o + s].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + s].view(p.shape)

    # -- torch.optim API subset the reference's training loops use ------------------
    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        self.step_count += 1
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, b1, b2,
                      g["eps"], g["lr"] / bc1, math.sqrt(bc2), grad_scale)

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        groups = [{k: v for k, v in g.items()} for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        for k in ("lr", "betas", "eps"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]
        for i, st in sd["state"].items():
            i = int(i)
            o, n = self.offsets[i], self.params[i].numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(float(st["step"]))
