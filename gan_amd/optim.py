"""Fused clip_grad_norm_ + AdamW over flat fp32 buffers (reference: torch.nn.utils.clip_grad_norm_ and
torch.optim.AdamW as used at /root/reference/lab 3/src/lab3_codec_train.py:302-311,446-448,589-591).

All parameters of a model are re-homed into ONE contiguous fp32 buffer (each nn.Parameter becomes a view),
with matching flat gradient / exp_avg / exp_avg_sq buffers.  A step is then: gather the gradients into the
flat buffer, one square-norm reduction, one HBM-bound update kernel (28 B/param).  Under data parallelism the
flat gradient buffer is what gets all-reduced, in contiguous buckets (gan_amd/dp.py).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch

from . import _lib as L


class FusedAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, max_grad_norm: float = 0.0):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FusedAdamW got no trainable parameters")
        dev = self.params[0].device
        L.require_cuda(*self.params)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.max_grad_norm = float(max_grad_norm)
        self.step_count = 0
        self.offsets: List[int] = []
        n = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("parameters must be float32 (master weights)")
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        self.numel = n
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self._sq = torch.zeros(1, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
        self.grad_views = [self.flat_g[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self.offsets)]
        from . import ops
        lo, hi = self.flat_p.data_ptr(), self.flat_p.data_ptr() + 4 * n
        self._epoch = ops.register_param_range(lo, hi)  # packed-weight cache
        import weakref
        weakref.finalize(self, ops.unregister_param_range, lo, hi)  # packs and epoch go when this optimizer does
        self.last_grad_norm: Optional[torch.Tensor] = None
        self.grad_scale = 1.0  # 1/world_size when flat_g holds a SUM over ranks
        self.sqnorm_hook = None  # called with the device scalar sum(g^2) between the norm kernel and the AdamW kernel
        self.device_step: Optional[torch.Tensor] = None  # int32 [1] once use_device_step() was called (HIP-graph replays)

    def use_device_step(self) -> None:
        """Keep the step number in device memory as well: step() then increments it with a device op and the AdamW kernel reads it, so
        an iteration recorded as a HIP graph applies the right bias correction on every replay.  The caller of a replay adds the replayed
        steps to `step_count` (the host copy, which checkpoints save) itself."""
        if self.device_step is None:
            self.device_step = torch.full((1,), int(self.step_count), device=self.flat_p.device, dtype=torch.int32)
        else:
            self.device_step.fill_(int(self.step_count))

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def gather_grads(self) -> torch.Tensor:
        """Copy every p.grad into the flat gradient buffer (missing gradients count as zero)."""
        self.flat_g.zero_()
        src = [p.grad for p in self.params if p.grad is not None]
        dst = [v for p, v in zip(self.params, self.grad_views) if p.grad is not None]
        if src:
            torch._foreach_copy_(dst, src)
        return self.flat_g

    @torch.no_grad()
    def step(self, grads_gathered: bool = False) -> None:
        if not grads_gathered:
            self.gather_grads()
        self.step_count += 1
        self._epoch[0] += 1  # the parameters change under a raw pointer: invalidate cached packs
        lib = L.lib()
        self._sq.zero_()
        L.check(lib.ga_grad_sqnorm(self.flat_g.data_ptr(), self.numel, self._sq.data_ptr(), L.stream()), "ga_grad_sqnorm")
        if self.sqnorm_hook is not None:
            self.sqnorm_hook(self._sq)  # data parallel: every rank clips by the same number (gan_amd/dp.py::DataParallel.sync_sqnorm)
        if self.device_step is not None:
            self.device_step += 1
            L.check(lib.ga_adamw_fused_devstep(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                               self.exp_avg_sq.data_ptr(), self.numel, self.lr, self.betas[0], self.betas[1], self.eps,
                                               self.weight_decay, self.device_step.data_ptr(), self.max_grad_norm, float(self.grad_scale),
                                               self._sq.data_ptr(), 0, L.stream()), "ga_adamw_fused_devstep")
        else:
            L.check(lib.ga_adamw_fused(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), self.numel, self.lr, self.betas[0], self.betas[1], self.eps,
                                       self.weight_decay, self.step_count, self.max_grad_norm, float(self.grad_scale),
                                       self._sq.data_ptr(), 0, L.stream()), "ga_adamw_fused")
        self.last_grad_norm = self._sq.sqrt() * float(self.grad_scale)  # device tensor, no sync

    # torch.optim.AdamW-compatible checkpoint format (reference _save_stage_checkpoint, lab3_codec_train.py:225-243)
    def state_dict(self) -> Dict:
        state = {}
        if self.step_count > 0:
            for i, (p, off) in enumerate(zip(self.params, self.offsets)):
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + p.numel()].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + p.numel()].view_as(p).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd: Dict) -> None:
        group = sd["param_groups"][0]
        self.lr, self.betas = float(group["lr"]), (float(group["betas"][0]), float(group["betas"][1]))
        self.eps, self.weight_decay = float(group["eps"]), float(group["weight_decay"])
        steps = set()
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.exp_avg[off:off + p.numel()].view_as(p).copy_(st["exp_avg"])
            self.exp_avg_sq[off:off + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdamW keeps one step counter; the checkpoint has per-parameter steps that differ")
        self.step_count = steps.pop() if steps else 0
        if self.device_step is not None:
            self.device_step.fill_(int(self.step_count))
