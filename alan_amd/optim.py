"""
``alan_amd.Adam``: the optimiser of the reference's training loop (basic_runner.py:108-110) as ONE launch of
libalan_mi355.so per step (alan_adam_step) with its step count on the device.  A ``torch.optim.Optimizer``: it drops in
where ``torch.optim.Adam(params, lr, betas, eps, maximize=..., capturable=True)`` stands, with the same arithmetic
(tests/test_e2e_host.py compares 100 steps against torch's fused capturable Adam).  What it buys: with it a training
iteration captured by ``GraphedStep`` holds library launches only, so the iteration is re-issued from its recorded launch
list instead of replayed as a HIP graph (no idle time between graph launches).  fp32 parameters on the GPU, no weight
decay, no amsgrad (the reference's runner uses neither).
"""
import ctypes as C

import torch as t

from . import native as N


class Adam(t.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, maximize=False):
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid beta parameters: {betas}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, maximize=maximize))

    def _group_state(self, group, device):
        st = group.get("_alan")
        if st is None or st["step"].device != device:
            st = group["_alan"] = {"step": t.zeros((), dtype=t.float32, device=device),
                                   "ticket": t.zeros((), dtype=t.int32, device=device)}
        return st

    @t.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with t.enable_grad():
                loss = closure()
        L = N.lib()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not (p.is_cuda and p.dtype == t.float32 and p.is_contiguous() and p.grad.dtype == t.float32):
                    raise N.NativeError("alan_amd.Adam takes contiguous fp32 parameters on the GPU (use torch.optim.Adam otherwise)")
                s = self.state[p]
                if not s:
                    s["exp_avg"], s["exp_avg_sq"] = t.zeros_like(p), t.zeros_like(p)
            gs = self._group_state(group, ps[0].device)
            stream = N.current_stream(ps[0].device)
            # (the launches of one step share its step count: the first ones read it, the LAST one advances it)
            chunks = [ps[i:i + N.ADAM_MAX_TENSORS] for i in range(0, len(ps), N.ADAM_MAX_TENSORS)]
            for ci, chunk in enumerate(chunks):
                d = N.AdamDesc()
                d.n_tensors, d.maximize = len(chunk), int(bool(group["maximize"]))
                keep = []
                for i, p in enumerate(chunk):
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    keep.append(g)
                    d.param[i], d.grad[i] = p.data_ptr(), g.data_ptr()
                    d.exp_avg[i], d.exp_avg_sq[i] = self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()
                    d.numel[i] = p.numel()
                d.lr, d.beta1, d.beta2, d.eps = float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"])
                last = ci == len(chunks) - 1
                d.step = gs["step"].data_ptr() if last else self._frozen_step(gs).data_ptr()
                d.ticket = gs["ticket"].data_ptr() if last else self._scratch_ticket(gs).data_ptr()
                N.check(L.alan_adam_step(C.byref(d), stream), "alan_adam_step")
                if N._REC[0] is not None:
                    N._REC[0].keep.append(keep)
                    N._REC[0].record(L.alan_adam_step, C.byref(d), None)
        return loss

    @staticmethod
    def _frozen_step(gs):
        """More tensors than one launch takes: the earlier launches of a step work from a COPY of the count (which their own
        last workgroup advances instead), refreshed from the real one at the start of every step."""
        if "step_copy" not in gs:
            gs["step_copy"] = t.zeros_like(gs["step"])
        gs["step_copy"].copy_(gs["step"])
        return gs["step_copy"]

    @staticmethod
    def _scratch_ticket(gs):
        if "ticket_copy" not in gs:
            gs["ticket_copy"] = t.zeros_like(gs["ticket"])
        return gs["ticket_copy"]
