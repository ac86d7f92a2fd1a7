"""`torch.optim.Optimizer`-shaped front of the fused Adafactor step (round 5).

The reference trains through HF `Trainer` with `optim="adafactor"` (ref:train.py:298; TF:trainer.py builds
`Adafactor(params, lr, scale_parameter=False, relative_step=False)`), i.e. one Python loop over ~460 parameter tensors with a
dozen small kernels each, after `clip_grad_norm_` has walked the same tensors twice.  On this framework's flat parameter /
gradient buffers the same update is ONE multi-tensor launch sequence (csrc/adafactor.hip through `ops.AdafactorPlan`: factored
second moments over the last two dims, per-tensor RMS clip, the bf16 compute copies refreshed in the same pass), and the
global-norm clip can ride in its statistics pass.  This class lets `Trainer(optimizers=(FusedAdafactor(model, ...), scheduler))`
take it:

    opt = FusedAdafactor(model, lr=5e-4, max_grad_norm=1.0)          # and TrainingArguments(max_grad_norm=0): no second clip
    trainer = Trainer(model=model, args=args, ..., optimizers=(opt, torch.optim.lr_scheduler.LambdaLR(opt, ...)))

Semantics follow HF's Adafactor with the Trainer's settings (eps = (1e-30, 1e-3), clip_threshold 1.0, decay_rate -0.8, no
momentum, no weight decay, external learning rate): `tests/golden/adafactor.npz` pins the kernel to HF's class
(tests/test_gpu_kernels.py), `tests/test_gpu_r5b.py` pins this wrapper to `transformers.Adafactor` stepped on the same
gradients.  Parameters whose `.grad` is None (frozen by the reference's FreezingCallback, or a LayerDrop-dropped layer's) are
skipped with their state untouched, like HF's loop does.
"""
from __future__ import annotations

import torch

from . import ops


class FusedAdafactor(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, clip_threshold=1.0, decay_rate=-0.8, eps1=1e-30, max_grad_norm=0.0):
        """model: a speechmix_amd model on the GPU (its FlatStore holds every parameter as a view of one buffer).
        max_grad_norm > 0: clip by the global gradient norm inside the step (what `Trainer` does with `clip_grad_norm_` before
        `optimizer.step()`: pass TrainingArguments(max_grad_norm=0) then, or the gradient is clipped twice)."""
        model._need_engine()
        self.model, self.store = model, model.store
        st = self.store
        if st.device.type != "cuda":
            raise RuntimeError("FusedAdafactor needs the model on the GPU (no CPU fallback)")
        self.names = [nm for nm, _ in sorted(st.offsets.items(), key=lambda kv: kv[1][0])]
        params = [st.params[nm] for nm in self.names]
        super().__init__(params, dict(lr=lr, clip_threshold=clip_threshold, decay_rate=decay_rate, eps1=eps1, max_grad_norm=max_grad_norm))
        self.plan = ops.AdafactorPlan([(st.offsets[nm][0], st.offsets[nm][2]) for nm in self.names], st.device)
        pre = model.engine.ep + "encoder.layers."
        self._layer = [(int(nm[len(pre):].split(".", 1)[0]) if nm.startswith(pre) else -1) for nm in self.names]
        self._params = params
        self._gptr = [st.g(nm).data_ptr() for nm in self.names]
        st.external_updates = False            # this optimizer refreshes the bf16 compute copies itself
        st.refresh_shadow(force=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st, g = self.store, self.param_groups[0]
        # HF skips parameters without a gradient: frozen ones (requires_grad flipped by the FreezingCallback), tensors whose .grad
        # is None, and - on one GPU - the layers LayerDrop skipped in this step's forward (their flat slices hold zeros)
        # (under DistributedDataParallel the all-reduce gives every tensor a gradient - other ranks kept the layer - so nothing is
        # skipped for LayerDrop there, exactly as StepRunner does)
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        # (dropped in EVERY micro-batch since zero_grad: under gradient accumulation a layer kept once holds a gradient - ADVICE r5)
        dropped = set() if multi else set(getattr(self.model.engine, "dropped_since_zero", ()) or ())
        active = []
        for nm, l, p, gp in zip(self.names, self._layer, self._params, self._gptr):
            on = bool(p.requires_grad and p.grad is not None and l not in dropped)
            active.append(on)
            # `.grad` normally IS the parameter's slice of the flat gradient (FlatStore.publish_grads).  A wrapper that installs its
            # own gradient tensors (DistributedDataParallel's bucket views) leaves the reduced values there: bring them over
            if on and p.grad.data_ptr() != gp:
                dst = st.g(nm)
                dst.copy_(p.grad.reshape(dst.shape))
        sh = None if st.shadow is st.master else st.shadow
        self.plan.step(st.master, st.grad, sh, float(g["lr"]), active=active, decay_rate=g["decay_rate"], eps1=g["eps1"],
                       clip_threshold=g["clip_threshold"], grad_scale=1.0, max_grad_norm=float(g["max_grad_norm"] or 0.0))
        st.mark_shadow_fresh()
        return loss

    # ---- checkpointing: the second-moment factors live in the plan's flat buffers
    def state_dict(self):
        sd = super().state_dict()
        p = self.plan
        sd["fused_adafactor"] = dict(row=p.row.detach().cpu(), col=p.col.detach().cpu(), rmean=p.rmean.detach().cpu(),
                                     steps=torch.from_numpy(p.steps.copy()), names=list(self.names))
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fused = state_dict.pop("fused_adafactor", None)
        super().load_state_dict(state_dict)
        if fused is not None:
            if list(fused["names"]) != list(self.names):
                raise ValueError("FusedAdafactor: the checkpoint's parameter list differs from this model's")
            p = self.plan
            p.row.copy_(fused["row"])
            p.col.copy_(fused["col"])
            p.rmean.copy_(fused["rmean"])
            p.steps[:] = fused["steps"].numpy()
