"""Diffusion sampler with the reference's call surface, driving the gfx950 kernels.

Surface kept (all under /root/reference/diffusion/):
  create_gaussian_diffusion(num_diffusion_timesteps, timestep_respacing, body_rep_mean, body_rep_std)  model_util.py:4-22
  SpacedDiffusion / GaussianDiffusion attributes (betas, alphas_cumprod, posterior_*, timestep_map, num_timesteps ...)
      gaussian_diffusion.py:122-169, respace.py:64-87
  .val_losses(model, batch, shape, clip_denoised, progress, cond_fn_with_grad, cond_grad_weight, cur_epoch,
              timestep_respacing, compute_loss)                                   gaussian_diffusion.py:749-780
  .p_sample_loop / .p_sample_loop_progressive / .p_sample / .p_sample_with_grad   gaussian_diffusion.py:298-508
  .ddim_sample_loop / .ddim_sample_loop_progressive / .ddim_sample                gaussian_diffusion.py:511-718
  .q_sample / .q_posterior_mean_variance / .p_mean_variance                        gaussian_diffusion.py:189-276

Two execution routes, both on the HIP library (no CPU route):
  * fused   - the model exposes ``fused_sampler`` (egohmr_amd.model.EgoHMR does): the whole T-step loop is
              one C-ABI call (ehm_sample_loop) with the step-invariant conditioning hoisted;
  * generic - any callable ``model(batch, t) -> {'pred_x_start': ...}``: Python drives the loop, each
              update is ehm_ddpm_step / ehm_ddim_step.
Extension over the reference: ``noise_stack=[T+1,B,144]`` feeds explicit N(0,1) draws (row 0 = x_T, row
1+k = k-th executed step) for cross-device parity; without it draws come from torch's generator in the
reference's order (randn(*shape) once, then randn_like once per step, t == 0 included).
"""
from __future__ import annotations

import math

import numpy as np
import torch as th

from . import _lib

__all__ = ["create_gaussian_diffusion", "GaussianDiffusion", "SpacedDiffusion", "space_timesteps", "get_named_beta_schedule"]


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.0):
    """gaussian_diffusion.py:22-66."""
    n = num_diffusion_timesteps
    if schedule_name == "linear":
        scale = scale_betas * 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if schedule_name == "cosine":
        abar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - abar((i + 1) / n) / abar(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def space_timesteps(num_timesteps, section_counts):
    """respace.py:8-61: which original timesteps a respaced process keeps."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                kept = range(0, num_timesteps, stride)
                if len(kept) == want:
                    return set(kept)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(s) for s in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = base + (i < extra)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            kept.append(start + round(cur))
            cur += stride
        start += size
    return set(kept)


def _extract(arr, timesteps, broadcast_shape):
    """gaussian_diffusion.py:784-797 (float64 table -> float32 at the gather)."""
    res = th.from_numpy(arr).to(device=timesteps.device)[timesteps].float()
    while res.dim() < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)


def _f32(v) -> float:
    return float(np.float32(v))


class GaussianDiffusion:
    def __init__(self, *, betas, rescale_timesteps=False, body_rep_mean=None, body_rep_std=None):
        self.rescale_timesteps = rescale_timesteps
        self.body_rep_mean, self.body_rep_std = body_rep_mean, body_rep_std
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        self.alphas_cumprod = ac
        self.alphas_cumprod_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod_next = np.append(ac[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - ac)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - ac)
        if not hasattr(self, "timestep_map"):
            self.timestep_map = list(range(self.num_timesteps))

    # ------------------------------------------------------------------ per-step coefficient rows
    def step_coefs(self, i: int, ddim: bool, eta: float = 0.0, grad_weight: float = 0.0, guided: bool = False):
        """ehm_step_coefs for respaced index i; float32 roundings as torch would make them."""
        c = _lib.StepCoefs()
        c.coef1, c.coef2 = _f32(self.posterior_mean_coef1[i]), _f32(self.posterior_mean_coef2[i])
        c.log_variance, c.variance = _f32(self.posterior_log_variance_clipped[i]), _f32(self.posterior_variance[i])
        c.sqrt_recip_ac, c.sqrt_recipm1_ac = _f32(self.sqrt_recip_alphas_cumprod[i]), _f32(self.sqrt_recipm1_alphas_cumprod[i])
        ab, abp = th.tensor(_f32(self.alphas_cumprod[i])), th.tensor(_f32(self.alphas_cumprod_prev[i]))
        sigma = eta * th.sqrt((1 - abp) / (1 - ab)) * th.sqrt(1 - ab / abp)           # :541-545, float32 tensor ops
        c.sqrt_ac_prev = float(th.sqrt(abp))
        c.dir_coef = float(th.sqrt(1 - abp - sigma ** 2))
        c.sigma = float(sigma)
        c.nonzero = 0.0 if i == 0 else 1.0
        c.grad_scale = 0.0
        if guided and not ddim and i <= 10:                                           # :378 (respaced index)
            # :381 float32(w) * variance (tensor op)  /  :385 float32(w * 0.01) (python floats first)
            c.grad_scale = _f32(np.float32(grad_weight) * np.float32(c.variance)) if i >= 5 else _f32(float(grad_weight) * 0.01)
        if guided and ddim and i <= 3:                                                # ddim_sample_with_grad :580-586: eps -= (1 - alpha_bar).sqrt() * grad * 1.0
            c.grad_scale = float(th.sqrt(1 - ab))                                     # (float32 tensor ops, as the reference's _extract_into_tensor broadcast)
        return c

    # ------------------------------------------------------------------ forward process (used for init_data)
    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = th.randn_like(x_start)
        assert noise.shape == x_start.shape
        return (_extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + _extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def q_mean_variance(self, x_start, t):
        return (_extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start,
                _extract(1.0 - self.alphas_cumprod, t, x_start.shape),
                _extract(self.log_one_minus_alphas_cumprod, t, x_start.shape))

    def q_posterior_mean_variance(self, x_start, x_t, t):
        assert x_start.shape == x_t.shape
        mean = (_extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + _extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return mean, _extract(self.posterior_variance, t, x_t.shape), _extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def _scale_timesteps(self, t):
        return t.float() * (1000.0 / self.num_timesteps) if self.rescale_timesteps else t

    def _model_timesteps(self, t):
        """respace.py:124-129: respaced index -> original timestep the model is conditioned on."""
        return t

    def p_mean_variance(self, model, batch, x, t, clip_denoised=True, denoised_fn=None):
        """gaussian_diffusion.py:233-276 (START_X parameterisation; clip_denoised is ignored there too)."""
        B = x.shape[0]
        assert t.shape == (B,)
        batch["x_t"] = x
        out = model(batch, self._model_timesteps(t))
        x0 = out["pred_x_start"]
        mean, var, logvar = self.q_posterior_mean_variance(x_start=x0, x_t=x, t=t)
        return {"mean": mean, "variance": var, "log_variance": logvar, "pred_xstart": x0, "other_outputs": out}

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        return (_extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart) / \
            _extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape)

    def _predict_xstart_from_eps(self, x_t, t, eps):
        return _extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - _extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * eps

    # ------------------------------------------------------------------ single steps (generic route)
    def _uniform_index(self, t) -> int:
        """The reference's p_sample / ddim_sample take one timestep per sample; its loops (and its guidance gate, `t[0] <= 10`,
        gaussian_diffusion.py:378) always pass a uniform vector.  The native steps take schedule coefficients by value, so a
        non-uniform ``t`` is refused instead of being silently read as t[0]."""
        if t.numel() > 1 and not bool((t == t[0]).all()):
            raise ValueError("egohmr_amd samplers need the same timestep for every item of the batch (got a non-uniform t)")
        return int(t[0])

    def _step(self, model, batch, x, t, ddim, guided, cond_grad_weight, eta, noise, index=None):
        i = self._uniform_index(t) if index is None else index      # the loops pass their python index: no device read-back
        batch["x_t"] = x
        mo = model(batch, self._model_timesteps(t))
        x0 = _lib.f32(mo["pred_x_start"], x.device)
        xc = _lib.f32(x)
        if noise is None:
            noise = th.randn_like(xc)                                                 # :331 / :547 (drawn even at t == 0)
        noise = _lib.f32(noise, x.device)
        c = self.step_coefs(i, ddim, eta, cond_grad_weight, guided and not ddim)   # (the generic route spells ddim_sample_with_grad out below)
        grad = None
        if c.grad_scale != 0.0:
            grad = _lib.f32(model.guide_coll(batch, mo, t, compute_grad="x_t"), x.device)   # :379
        if ddim and guided and i <= 3:
            # ddim_sample_with_grad, gaussian_diffusion.py:580-592: on the last four respaced steps the collision gradient is
            # subtracted from eps (scale 1.0) and x0 re-derived from it; the plain DDIM update then runs on that x0.  The
            # reference's float32 scalar broadcasts (`_extract_into_tensor`) are mirrored with float32 tensors.
            g = _lib.f32(model.guide_coll(batch, mo, t, compute_grad="x_t"), x.device)       # :584
            f = lambda v: th.tensor(v, dtype=th.float64).float().to(x.device)
            ab, sr, srm1 = f(self.alphas_cumprod[i]), f(self.sqrt_recip_alphas_cumprod[i]), f(self.sqrt_recipm1_alphas_cumprod[i])
            eps = (sr * xc - x0) / srm1                                                      # :582
            eps = eps - (1 - ab).sqrt() * g * 1.0                                            # :585-586
            x0 = (sr * xc - srm1 * eps).contiguous()                                         # :587
            mo = dict(mo)
            mo["pred_x_start_guided"] = x0
        out = th.empty_like(xc)
        L, n, st = _lib.lib(), xc.numel(), _lib.stream_ptr()
        if ddim:
            _lib.check(L.ehm_ddim_step(_lib.ptr(xc), _lib.ptr(x0), _lib.ptr(noise), _lib.ptr(out), c.sqrt_recip_ac, c.sqrt_recipm1_ac,
                                       c.sqrt_ac_prev, c.dir_coef, c.sigma, c.nonzero, n, st), "ehm_ddim_step")
        else:
            _lib.check(L.ehm_ddpm_step(_lib.ptr(xc), _lib.ptr(x0), _lib.ptr(noise), _lib.ptr(grad), _lib.ptr(out), c.coef1, c.coef2,
                                       c.log_variance, c.nonzero, c.grad_scale, n, st), "ehm_ddpm_step")
        return {"sample": out, "pred_xstart": mo.get("pred_x_start_guided", mo["pred_x_start"]), "other_outputs": mo}

    def p_sample(self, model, batch, x, t, clip_denoised=True, denoised_fn=None, cond_grad_weight=0.0, noise=None):
        return self._step(model, batch, x, t, False, False, cond_grad_weight, 0.0, noise)

    def p_sample_with_grad(self, model, batch, x, t, clip_denoised=True, denoised_fn=None, cond_grad_weight=1.0, noise=None):
        return self._step(model, batch, x, t, False, True, cond_grad_weight, 0.0, noise)

    def ddim_sample(self, model, batch, x, t, clip_denoised=True, denoised_fn=None, eta=0.0, noise=None):
        return self._step(model, batch, x, t, True, False, 0.0, eta, noise)

    def ddim_sample_with_grad(self, model, batch, x, t, clip_denoised=True, denoised_fn=None, eta=0.0, noise=None):
        """gaussian_diffusion.py:559-614 (the reference notes at :579 that DDIM "does not work well" with the collision guidance; built for
        completeness of the call surface).  This is the single step of the generic route; ddim_sample_loop(cond_fn_with_grad=True) runs the same
        update inside the one-call loop (step_dev.h: step_body_one, ehm_step_coefs.grad_scale = sqrt(1 - alpha_bar) on the last four steps)."""
        return self._step(model, batch, x, t, True, True, 1.0, eta, noise)

    # ------------------------------------------------------------------ loops
    def _device_of(self, model, device):
        if device is not None:
            return device
        return next(model.parameters()).device                                        # :472-473

    def _loop(self, model, batch, shape, ddim, noise, device, progress, eta, skip_timesteps, init_data, cond_fn_with_grad,
              cond_grad_weight, noise_stack):
        device = self._device_of(model, device)
        assert isinstance(shape, (tuple, list))
        indices = list(range(self.num_timesteps - skip_timesteps))[::-1]              # :483
        if noise_stack is not None:
            noise_stack = _lib.f32(noise_stack, device)
            assert noise_stack.shape[0] >= len(indices) + 1
            data = noise_stack[0]
        else:
            data = noise if noise is not None else th.randn(*shape, device=device)    # :475-478
        if skip_timesteps and init_data is None:
            init_data = th.zeros_like(data)
        if init_data is not None:
            my_t = th.ones([shape[0]], device=device, dtype=th.long) * indices[0]
            data = self.q_sample(init_data, my_t, data)
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for k, i in enumerate(indices):
            t = th.tensor([i] * shape[0], device=device)                              # :495
            with th.no_grad():
                eps = None if noise_stack is None else noise_stack[1 + k]
                out = self._step(model, batch, data, t, ddim, cond_fn_with_grad, cond_grad_weight, eta, eps, index=i)
                yield out
                data = out["sample"]

    def p_sample_loop_progressive(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None, device=None,
                                  progress=False, skip_timesteps=0, init_data=None, cond_fn_with_grad=False, cond_grad_weight=1.0,
                                  noise_stack=None):
        yield from self._loop(model, batch, shape, False, noise, device, progress, 0.0, skip_timesteps, init_data,
                              cond_fn_with_grad, cond_grad_weight, noise_stack)

    def ddim_sample_loop_progressive(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None, device=None,
                                     progress=False, eta=0.0, skip_timesteps=0, init_data=None, cond_fn_with_grad=False,
                                     noise_stack=None):
        yield from self._loop(model, batch, shape, True, noise, device, progress, eta, skip_timesteps, init_data,
                              cond_fn_with_grad, 1.0, noise_stack)

    def _fused_ok(self, model, skip_timesteps, init_data, dump_steps, progress, eta):
        return (getattr(self, "allow_fused", True) and getattr(model, "fused_sampler", None) is not None
                and not skip_timesteps and init_data is None
                and dump_steps is None and not progress and eta == 0.0)

    def _draw_stack(self, shape, device, noise):
        """Same generator consumption as the reference: randn(*shape) then randn_like per step."""
        rows = [noise if noise is not None else th.randn(*shape, device=device)]
        for _ in range(self.num_timesteps):
            rows.append(th.randn_like(rows[0]))
        return th.stack(rows)

    def p_sample_loop(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None, device=None, progress=False,
                      skip_timesteps=0, init_data=None, cond_fn_with_grad=False, cond_grad_weight=1.0, dump_steps=None,
                      noise_stack=None):
        """gaussian_diffusion.py:391-446 -> dict(sample, pred_xstart, other_outputs) of the last step."""
        if self._fused_ok(model, skip_timesteps, init_data, dump_steps, progress, 0.0):
            device = self._device_of(model, device)
            stack = noise_stack if noise_stack is not None else self._draw_stack(shape, device, noise)
            return model.fused_sampler.run(self, batch, stack, ddim=False, guided=cond_fn_with_grad, cond_grad_weight=cond_grad_weight)
        final, dump = None, []
        for i, sample in enumerate(self.p_sample_loop_progressive(
                model, batch, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, device=device,
                progress=progress, skip_timesteps=skip_timesteps, init_data=init_data, cond_fn_with_grad=cond_fn_with_grad,
                cond_grad_weight=cond_grad_weight, noise_stack=noise_stack)):
            if dump_steps is not None and i in dump_steps:
                dump.append(sample["sample"].clone())
            final = sample
        return dump if dump_steps is not None else final

    def ddim_sample_loop(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None, device=None, progress=False,
                         eta=0.0, skip_timesteps=0, init_data=None, cond_fn_with_grad=False, noise_stack=None):
        """gaussian_diffusion.py:618-658."""
        if self._fused_ok(model, skip_timesteps, init_data, None, progress, eta):
            # (cond_fn_with_grad: ddim_sample_with_grad inside the one-call loop - the collision gradient enters eps on the last four respaced steps)
            device = self._device_of(model, device)
            stack = noise_stack if noise_stack is not None else self._draw_stack(shape, device, noise)
            return model.fused_sampler.run(self, batch, stack, ddim=True, guided=bool(cond_fn_with_grad), cond_grad_weight=1.0 if cond_fn_with_grad else 0.0)
        final = None
        for sample in self.ddim_sample_loop_progressive(
                model, batch, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, device=device,
                progress=progress, eta=eta, skip_timesteps=skip_timesteps, init_data=init_data, cond_fn_with_grad=cond_fn_with_grad,
                noise_stack=noise_stack):
            final = sample
        return final

    def training_losses(self, model, batch, t, cur_epoch=0, noise=None):
        raise NotImplementedError("training (gaussian_diffusion.py:721-746) is outside the sampling hot path this package implements")

    def val_losses(self, model, batch, shape, clip_denoised=True, progress=False, cond_fn_with_grad=False, cond_grad_weight=1.0,
                   cur_epoch=0, timestep_respacing="", compute_loss=True, noise_stack=None):
        """gaussian_diffusion.py:749-780: eval-mode sampling, returns the final step's model output dict."""
        model.validation_setup()
        if timestep_respacing == "":
            out = self.p_sample_loop(model=model, batch=batch, shape=shape, progress=progress, clip_denoised=clip_denoised,
                                     cond_fn_with_grad=cond_fn_with_grad, cond_grad_weight=cond_grad_weight, noise_stack=noise_stack)
        elif timestep_respacing[0:4] == "ddim":
            out = self.ddim_sample_loop(model=model, batch=batch, shape=shape, progress=progress, clip_denoised=clip_denoised,
                                        eta=0.0, cond_fn_with_grad=cond_fn_with_grad, noise_stack=noise_stack)
        else:
            print("timestep_respacing_eval not setup correctly")                       # :774-775
            raise SystemExit()
        if compute_loss:
            model.compute_loss(batch, out["other_outputs"], cur_epoch=cur_epoch)
        return out["other_outputs"]


class SpacedDiffusion(GaussianDiffusion):
    """respace.py:64-114: keep ``use_timesteps`` of a base process; betas re-derived from its alpha-bar."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        self.timestep_map, new_betas, last = [], [], 1.0
        for i, ac in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _model_timesteps(self, t):
        new_ts = th.tensor(self.timestep_map, device=t.device, dtype=t.dtype)[t]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return new_ts

    def _scale_timesteps(self, t):
        return t


def create_gaussian_diffusion(num_diffusion_timesteps=1000, timestep_respacing="ddim5", body_rep_mean=None, body_rep_std=None):
    """model_util.py:4-22: cosine schedule, no timestep rescaling."""
    steps = num_diffusion_timesteps
    betas = get_named_beta_schedule("cosine", steps, 1.0)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return SpacedDiffusion(use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas, rescale_timesteps=False,
                           body_rep_mean=body_rep_mean, body_rep_std=body_rep_std)
