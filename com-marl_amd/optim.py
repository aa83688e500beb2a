"""Adam for the policy / critic parameter sets as TWO launches per step (cm_multi_adam_step: one for the gradient norm, one
for clip + update of every tensor) instead of ~6 framework kernels and a host sync per tensor.

Same update rule, defaults and state layout as the reference's vendored torch-1.9 Adam
(com_marl/torch/algos/my_optimizer/adam.py:56-120, _functional.py:72-98; state keys ``step`` / ``exp_avg`` /
``exp_avg_sq``), so ``state_dict()`` interchanges with ``torch.optim.Adam``.  ``step(max_norm=...)`` folds
``torch.nn.utils.clip_grad_norm_`` (centralized_ma_ppo.py:253-255) into the same launches and returns the pre-clip norm
as a DEVICE tensor - nothing in an optimiser step waits for the host."""
import ctypes as C

import torch

from . import _lib as L


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("the runners use plain Adam (no weight decay, no amsgrad)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._norm = None
        self.norm_sq = None
        self._dev_steps = None

    # ------------------------------------------------------------------------------------------
    # device-side step counter: lets ONE captured hipGraph of an optimiser step be replayed for successive steps
    # ------------------------------------------------------------------------------------------
    DEV_STEP_CAPACITY = 256      # optimiser steps one begin_device_steps() can cover (the reference schedule takes 30 per epoch)

    def begin_device_steps(self, n_steps):
        """From now until end_device_steps(), step() reads the bias corrections of step `step + 1 + cursor` from a device table
        and advances the device cursor itself (both captured with the step); the host-side ``state[p]['step']`` stands still.
        Table and cursor are allocated once per optimiser and refilled in place: a hipGraph captured in one epoch can be
        replayed in a later one."""
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None or self.state[p]]
        steps = {int(self.state[p]["step"]) for p in ps if self.state[p]}
        assert len(steps) == 1 and len(self.param_groups) == 1, "device steps: one parameter group that has stepped together"
        assert 1 <= n_steps <= self.DEV_STEP_CAPACITY, "device steps: at most DEV_STEP_CAPACITY steps per begin_device_steps()"
        first = steps.pop() + 1
        b1, b2 = self.param_groups[0]["betas"]
        host = torch.zeros(self.DEV_STEP_CAPACITY, 2, dtype=torch.float32)
        L.lib().cm_adam_bias_corrections(float(b1), float(b2), first, n_steps, host.data_ptr())
        dev = ps[0].device
        buf = getattr(self, "_dev_step_buf", None)
        if buf is None or buf[0].device != dev:
            buf = self._dev_step_buf = (torch.zeros(self.DEV_STEP_CAPACITY, 2, dtype=torch.float32, device=dev),
                                        torch.zeros(1, dtype=torch.int32, device=dev))
        buf[0].copy_(host)
        buf[1].zero_()
        self._dev_steps = dict(table=buf[0], cursor=buf[1], n=n_steps, params=ps)

    def end_device_steps(self, n_done):
        """Book the `n_done` replayed steps into the host-side state (and the parameters' version counters)."""
        ds, self._dev_steps = self._dev_steps, None
        if ds is None:
            return
        assert 0 <= n_done <= ds["n"]
        for p in ds["params"]:
            self.state[p]["step"] = int(self.state[p]["step"]) + n_done
            if n_done:
                torch.autograd.graph.increment_version(p)

    @torch.no_grad()
    def step(self, closure=None, max_norm=None, return_norm=True):
        """-> pre-clip gradient norm (device scalar tensor) when max_norm is given, else None.  return_norm=False skips the
        square-root launch: the caller reads |g|^2 from ``norm_sq`` (a view of the kernels' workspace, overwritten by the next
        step)."""
        assert closure is None
        out = None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise L.CommarlError("com_marl_amd.optim.Adam updates CUDA parameters (there is no CPU path)")
            b1, b2 = group["betas"]
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if self._dev_steps is None:
                    st["step"] = int(st["step"]) + 1
            steps = {int(self.state[p]["step"]) for p in ps}
            assert len(steps) == 1, "parameters of one group step together"
            step = steps.pop()
            ds = self._dev_steps
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            for p, g in zip(ps, grads):
                if g is not p.grad:
                    p.grad = g
            n = len(ps)
            arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])          # noqa: E731
            sizes = (C.c_int64 * n)(*[p.numel() for p in ps])
            norm_ptr = None
            if max_norm is not None:
                if self._norm is None or self._norm.device != dev:
                    self._norm = torch.zeros(65, dtype=torch.float32, device=dev)   # CM_ADAM_NORM_FLOATS: [0] = |g|^2
                norm_ptr = self._norm.data_ptr()
                out = self._norm[0]
            if n > 40:
                raise L.CommarlError("cm_multi_adam_step takes at most 40 tensors per parameter group")
            with torch.cuda.device(dev):
                if ds is not None:
                    L.check(L.lib().cm_multi_adam_step_dev(
                        n, arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]),
                        arr([self.state[p]["exp_avg_sq"] for p in ps]), sizes, norm_ptr,
                        float(max_norm if max_norm is not None else 0.0), float(group["lr"]), float(b1), float(b2),
                        float(group["eps"]), ds["table"].data_ptr(), ds["cursor"].data_ptr(), self.DEV_STEP_CAPACITY, L.current_stream()),
                        "cm_multi_adam_step_dev")
                    ds["cursor"].add_(1)
                else:
                    L.check(L.lib().cm_multi_adam_step(
                        n, arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]),
                        arr([self.state[p]["exp_avg_sq"] for p in ps]), sizes, norm_ptr,
                        float(max_norm if max_norm is not None else 0.0), float(group["lr"]), float(b1), float(b2),
                        float(group["eps"]), step, L.current_stream()), "cm_multi_adam_step")
            # the kernel wrote through raw pointers: tell torch (and the nets' weight-pack cache, which keys on the version
            # counters) that the parameters changed
            for p in ps:
                torch.autograd.graph.increment_version(p)
                if max_norm is not None:
                    torch.autograd.graph.increment_version(p.grad)
        self.norm_sq = out
        return None if (out is None or not return_norm) else out.sqrt()
