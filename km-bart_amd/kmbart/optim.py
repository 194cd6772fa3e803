"""transformers-3.0.2-style AdamW (reference vcg_train.py:13,100: lr given, betas (0.9, 0.999), eps 1e-6,
weight_decay 0.0, correct_bias True) whose step() is ONE fused HIP kernel over the engine's flat arena
(28 B/param of HBM traffic + the bf16 mirror), instead of ~260 per-tensor launches."""
import torch


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: {} - should be in [0.0, 1.0[".format(betas))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      correct_bias=correct_bias))
        self.grad_scale = 1.0  # data-parallel wrapper may fold 1/world here
        import os
        self.overlap = os.environ.get("KMB_ADAMW_OVERLAP", "1") != "0"

    def _engine_ranges(self, group):
        """[(engine, offset, count)] with adjacent parameters coalesced."""
        items = []
        for p in group["params"]:
            eng = getattr(p, "_kmb_engine", None)
            if eng is None:
                raise RuntimeError("kmbart.optim.AdamW only steps parameters that live in a kmbart engine arena "
                                   "(move the model to the GPU before building the optimizer)")
            off, cnt = p._kmb_range
            items.append((eng, off, cnt))
        items.sort(key=lambda x: (id(x[0]), x[1]))
        out = []
        for eng, off, cnt in items:
            if out and out[-1][0] is eng and out[-1][1] + out[-1][2] == off:
                out[-1] = (eng, out[-1][1], out[-1][2] + cnt)
            else:
                out.append((eng, off, cnt))
        return out

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            ranges = self._engine_ranges(group)
            engines = {id(e): e for e, _, _ in ranges}
            for e in engines.values():
                e.step_count += 1
            for eng, off, cnt in ranges:
                args = (group["lr"], group["betas"], group["eps"], group["weight_decay"], group["correct_bias"],
                        self.grad_scale)
                # per gradient bucket, beside the backward pass still running on the GPU (Engine.adamw_step_overlapped)
                if not (self.overlap and eng.adamw_step_overlapped(*args, offset=off, count=cnt)):
                    eng.adamw_step(*args, offset=off, count=cnt, bump=False)
        return loss

    def zero_grad(self, set_to_none=False):
        pass  # gradients are views into the arena and are overwritten by the next backward

    def state_dict(self):
        engines = {}
        for g in self.param_groups:
            for e, _, _ in self._engine_ranges(g):
                engines[id(e)] = e
        eng = list(engines.values())
        return {
            "kmbart_adamw": True,
            "step": [e.step_count for e in eng],
            "exp_avg": [e.exp_avg.detach().cpu() for e in eng],
            "exp_avg_sq": [e.exp_avg_sq.detach().cpu() for e in eng],
            "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
        }

    def load_state_dict(self, state):
        engines = {}
        for g in self.param_groups:
            for e, _, _ in self._engine_ranges(g):
                engines[id(e)] = e
        for i, e in enumerate(engines.values()):
            e.step_count = int(state["step"][i])
            e.exp_avg.copy_(state["exp_avg"][i])
            e.exp_avg_sq.copy_(state["exp_avg_sq"][i])
        for g, s in zip(self.param_groups, state["param_groups"]):
            g.update(s)
