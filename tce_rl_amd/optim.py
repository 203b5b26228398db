"""Flat Adam on the HIP optimizer kernel (csrc/optim.hip).

Same update rule, hyper-parameters and ``state_dict`` layout as the
``torch.optim.Adam(params, lr, weight_decay)`` the reference builds
(mprl/rl/agent/abstract_agent.py:62-82), so optimizer checkpoints
(``policy_optimizer_state_<epoch>``) stay interchangeable and ``LinearLR``
drives ``param_groups[0]["lr"]`` unchanged.  Differences in mechanism only:

* parameters, gradients and both moments live in ONE flat buffer each (the
  parameters / ``.grad`` of the modules become views), so a step is two kernel
  launches whatever the number of layers, and a multi-GPU step all-reduces one
  buffer;
* the global gradient norm, the clip factor of ``grad_norm_clip``
  (mprl/util/util_numerical.py:244-275) and the step count stay on the device.
"""
import torch

from ._lib import call, ptr, sfx, stream


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999),
                 eps=1e-8):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps,
                                      weight_decay=weight_decay))
        p0 = params[0]
        if not p0.is_cuda:
            raise RuntimeError("FlatAdam: parameters must live on the GPU "
                               "(there is no CPU path)")
        assert all(p.dtype == p0.dtype and p.device == p0.device
                   for p in params)
        self._params = params
        n = sum(p.numel() for p in params)
        mk = lambda: torch.zeros(n, dtype=p0.dtype, device=p0.device)
        self.flat_param, self.flat_grad, self.m, self.v = mk(), mk(), mk(), mk()
        # {step, |g| before clip, |g| after, clip factor}
        self.dev_state = torch.zeros(4, dtype=p0.dtype, device=p0.device)
        self.host_step = 0        # host mirror of dev_state[0]
        self._views = []
        off = 0
        for p in params:
            k = p.numel()
            sl = slice(off, off + k)
            self.flat_param[sl].copy_(p.data.reshape(-1))
            p.data = self.flat_param[sl].view(p.shape)
            gv = self.flat_grad[sl].view(p.shape)
            p.grad = gv
            self._views.append(gv)
            self.state[p] = {"step": self.dev_state[0],
                             "exp_avg": self.m[sl].view(p.shape),
                             "exp_avg_sq": self.v[sl].view(p.shape)}
            off += k

    def zero_grad(self, set_to_none=True):
        self.flat_grad.zero_()
        self.bind_grads()

    def bind_grads(self):
        for p, gv in zip(self._params, self._views):
            if p.grad is not gv:
                p.grad = gv

    @torch.no_grad()
    def sync_grads(self):
        """Gradients that autograd put into fresh tensors -> the flat buffer."""
        for p, gv in zip(self._params, self._views):
            if p.grad is not gv:
                if p.grad is None:
                    gv.zero_()
                else:
                    gv.copy_(p.grad)
                p.grad = gv

    @torch.no_grad()
    def step(self, clip=0.0, sumsq=None, grad_scale=1.0):
        """One Adam step on the flat buffers; returns the gradient norm before
        and after clipping as 0-dim device tensors (valid until the next
        step).  grad_scale: factor on the stored gradient (1 / world after a
        summing all-reduce)."""
        g = self.param_groups[0]
        self.sync_grads()
        self.host_step += 1
        call("tce_adam_flat_" + sfx(self.flat_param.dtype), ptr(self.flat_param),
             ptr(self.flat_grad), ptr(self.m), ptr(self.v),
             self.flat_param.numel(), ptr(self.dev_state), ptr(sumsq),
             float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
             float(g["eps"]), float(g["weight_decay"]), float(clip),
             float(grad_scale), stream())
        return self.dev_state[1], self.dev_state[2]

    @torch.no_grad()
    def step_once(self, clip=0.0, grad_scale=1.0, norms_out=None):
        """``step`` as ONE launch (tce_adam_once_*: buffers of <= 2^17 elements;
        larger ones take ``step``): the step count comes from the host mirror.
        norms_out: [2] device tensor for {|g|, |g| clipped}."""
        n = self.flat_param.numel()
        if n > (1 << 17):
            self.step(clip, grad_scale=grad_scale)
            if norms_out is not None:
                norms_out.copy_(self.dev_state[1:3])
            return
        g = self.param_groups[0]
        self.sync_grads()
        self.host_step += 1
        self._opt_called = True               # for LinearLR's order check
        call("tce_adam_once_" + sfx(self.flat_param.dtype), ptr(self.flat_param),
             ptr(self.flat_grad), ptr(self.m), ptr(self.v), n,
             ptr(self.dev_state), ptr(norms_out), float(self.host_step),
             float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
             float(g["eps"]), float(g["weight_decay"]), float(clip),
             float(grad_scale), stream())

    @torch.no_grad()
    def step_exchange(self, xchg, clip=0.0, grad_scale=1.0, norms_out=None):
        """Sum of the flat gradient over the ranks of `xchg` (a dist.Exchange;
        the sum stays in ``flat_grad``) + ``step_once`` as ONE C call and --
        without clipping -- ONE launch (tce_xchg_adam_*)."""
        g = self.param_groups[0]
        self.sync_grads()
        self.host_step += 1
        self._opt_called = True
        call("tce_xchg_adam_" + sfx(self.flat_param.dtype), xchg.handle,
             ptr(self.flat_param), ptr(self.flat_grad), ptr(self.m), ptr(self.v),
             self.flat_param.numel(), ptr(self.dev_state), ptr(norms_out),
             float(self.host_step), float(g["lr"]), float(g["betas"][0]),
             float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
             float(clip), float(grad_scale), stream())

    def state_dict(self):
        sd = super().state_dict()
        for st in sd["state"].values():          # detach from the flat buffers
            for k, v in list(st.items()):
                if torch.is_tensor(v):
                    st[k] = v.detach().clone()
        return sd

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        for g, sg in zip(self.param_groups, groups):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
        idx = [i for sg in groups for i in sg["params"]]
        for p, i in zip(self._params, idx):
            st = state_dict["state"].get(i)
            if st is None:
                continue
            mine = self.state[p]
            mine["exp_avg"].copy_(st["exp_avg"])
            mine["exp_avg_sq"].copy_(st["exp_avg_sq"])
            self.dev_state[0] = float(st["step"])
            self.host_step = int(float(st["step"]))
