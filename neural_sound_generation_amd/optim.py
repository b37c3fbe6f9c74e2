"""Flat-bucket Adam: every parameter of the model lives in ONE contiguous fp32 buffer (each tensor
256-byte aligned inside it), gradients in a second one, so the optimiser is a single kernel launch
(nsg_adam_step) and data-parallel training needs a single all-reduce per step.

Arithmetic = torch.optim.Adam defaults as constructed at src/main.py:124 (no weight decay, no
amsgrad), applied element-wise, so the update is identical to the per-tensor optimiser's.
"""
from __future__ import annotations

import torch

from . import ops

_ALIGN = 64  # floats (256 bytes): keeps every view 16-byte aligned for the float4 kernels


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params if p.requires_grad]   # e.g. an EMA-trained codebook takes no gradient step
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._params = [p for g in self.param_groups for p in g["params"]]
        if not self._params:
            raise ValueError("FlatAdam got no parameters")
        dev = self._params[0].device
        offs, total = [], 0
        for p in self._params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatAdam needs float32 parameters on one device")
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offsets, self.total = offs, total
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad_views = []
        with torch.no_grad():
            for p, off in zip(self._params, offs):
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view                      # parameters now alias the flat buffer
                gview = self.flat_grad[off:off + p.numel()].view_as(p)
                p.grad = gview                     # autograd accumulates in place into the bucket
                self.grad_views.append(gview)
        self.flat_comm = self.flat_grad          # what a data-parallel step all-reduces: [gradients | reserved tail]
        self.step_count = 0

    @torch.no_grad()
    def reserve_tail(self, n_floats: int) -> torch.Tensor:
        """Room for `n_floats` fp32 values BEHIND the gradients in the same allocation: per-step statistics that must be
        summed over ranks (the EMA codebook's per-code counts and sums) then ride in the gradients' one all-reduce
        (SURVEY.md section 8e).  Re-binds every p.grad to the new storage; returns the tail view."""
        pad = (int(n_floats) + _ALIGN - 1) // _ALIGN * _ALIGN
        comm = torch.zeros(self.total + pad, dtype=torch.float32, device=self.flat_grad.device)
        comm[:self.total].copy_(self.flat_grad)
        self.flat_comm = comm
        self.flat_grad = comm[:self.total]
        self.grad_views = []
        for p, off in zip(self._params, self.offsets):
            gview = self.flat_grad[off:off + p.numel()].view_as(p)
            p.grad = gview
            self.grad_views.append(gview)
        return comm[self.total:self.total + int(n_floats)]

    def zero_grad(self, set_to_none: bool = False):
        # the views must survive: never set to None
        self.flat_grad.zero_()
        for p, g in zip(self._params, self.grad_views):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def grads_for(self, tensors):
        """Bucket views for the given parameter tensors (same order)."""
        index = {id(p): i for i, p in enumerate(self._params)}
        return [self.grad_views[index[id(t)]] for t in tensors]

    # ---- checkpoint interchange (src/main.py:216-220 saves optimizer.state_dict()) -------------------
    # Same layout as torch.optim.Adam's: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]},
    # so a checkpoint written with either optimiser loads into the other.
    def state_dict(self):
        g = self.param_groups[0]
        defaults = dict(torch.optim.Adam([torch.zeros(1)]).defaults)   # every key torch's Adam expects in a group
        group = {**defaults, "lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "params": list(range(len(self._params)))}
        state = {}
        if self.step_count > 0:
            for i, (p, off) in enumerate(zip(self._params, self.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._params):
            raise ValueError("FlatAdam.load_state_dict: expected one parameter group with %d parameters" % len(self._params))
        g = self.param_groups[0]
        g["lr"], g["betas"], g["eps"] = groups[0]["lr"], tuple(groups[0]["betas"]), groups[0]["eps"]
        if groups[0].get("weight_decay", 0) or groups[0].get("amsgrad", False):
            raise ValueError("FlatAdam.load_state_dict: weight decay / amsgrad are not implemented")
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for i, (p, off) in enumerate(zip(self._params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: parameters carry different step counts")
        self.step_count = steps.pop() if steps else 0

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.step_count += 1
        if self.flat_param.is_cuda:
            ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_count, lr=g["lr"],
                          beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], grad_scale=grad_scale)
            # the kernel updated the parameters through raw pointers: bump autograd's version counters (each Parameter keeps its
            # own, `p.data = view` does not share the bucket's), so a backward through a graph built BEFORE this step raises
            # instead of differentiating against new values
            torch.autograd.graph.increment_version(self._params)
        else:
            raise RuntimeError("FlatAdam.step: parameters are not on a GPU; this path has no CPU fallback")
        return loss
