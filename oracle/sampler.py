"""Oracle: DDPM / DDIM sampling loops on the CPU (eager torch).  Test infrastructure only.

Follows diffusion/gaussian_diffusion.py: p_mean_variance :233-276 (START_X parameterisation,
fixed-small variance, ``clip_denoised`` ignored), q_posterior_mean_variance :209-231, p_sample
:298-337, p_sample_with_grad :340-388, p_sample_loop_progressive :449-508, ddim_sample :511-556
(eta = 0 as val_losses passes, :770-772), _extract_into_tensor :784-797 (float64 table -> float32
at the gather), and diffusion/respace.py:124-129 (respaced index -> original timestep for the model).

Noise is explicit: ``noise[0]`` is x_T, ``noise[1+k]`` the draw of the k-th executed step - the
reference's own draw order (:478 then one randn_like per step, including t == 0 where it is masked).
"""
from __future__ import annotations

import torch

from .schedule import Tables


def _coef(table, i, dtype):
    # _extract_into_tensor: from_numpy(float64)[t].float()  -> one float32 scalar broadcast over [B,144]
    return torch.tensor(table[i], dtype=torch.float64).float().to(dtype)


def p_sample_loop(model, batch, tables: Tables, noise: torch.Tensor, cond_fn_with_grad=False,
                  cond_grad_weight=1.0, guide_reduction="mean", trace=None, guide_all_points=False):
    """Returns the last step's dict {'sample','pred_xstart','other_outputs'} (gaussian_diffusion.py:391-446)."""
    dt = noise.dtype
    x = noise[0]
    B = x.shape[0]
    out = None
    T = tables.num_timesteps
    for k, i in enumerate(range(T - 1, -1, -1)):
        t_model = torch.full((B,), tables.timestep_map[i], dtype=torch.long)           # respace.py:125-126
        batch["x_t"] = x                                                               # :256
        with torch.no_grad():
            mo = model(batch, t_model)
        x0 = mo["pred_x_start"]
        mean = _coef(tables.posterior_mean_coef1, i, dt) * x0 + _coef(tables.posterior_mean_coef2, i, dt) * x   # :217-220
        var = _coef(tables.posterior_variance, i, dt)
        logvar = _coef(tables.posterior_log_variance_clipped, i, dt)
        eps = noise[1 + k]
        if cond_fn_with_grad and i <= 10:                                              # :378 (respaced index)
            g, _ = model.guide_coll(batch, mo, t_model, compute_grad="x_t", reduction=guide_reduction, all_points=guide_all_points)
            if i >= 5:
                mean = mean.float() + cond_grad_weight * var * g.float()               # :381
            else:
                mean = mean.float() + cond_grad_weight * 0.01 * g.float()              # :385
            mean = mean.to(dt)
        nz = 0.0 if i == 0 else 1.0
        x = mean + nz * torch.exp(0.5 * logvar) * eps                                  # :336
        out = {"sample": x, "pred_xstart": x0, "other_outputs": mo}
        if trace is not None:
            trace.append((x.clone(), x0.clone()))
    return out


def ddim_sample_loop(model, batch, tables: Tables, noise: torch.Tensor, eta: float = 0.0, trace=None, cond_fn_with_grad=False,
                     guide_reduction="mean"):
    """gaussian_diffusion.py:618-718 with ddim_sample :511-556, or ddim_sample_with_grad :559-614 when cond_fn_with_grad
    (the loop picks it at :700-703): on the last four respaced steps (t <= 3) the collision gradient is subtracted from eps,
    x0 is re-derived from that eps, and the plain DDIM update continues from there."""
    dt = noise.dtype
    x = noise[0]
    B = x.shape[0]
    out = None
    T = tables.num_timesteps
    for k, i in enumerate(range(T - 1, -1, -1)):
        t_model = torch.full((B,), tables.timestep_map[i], dtype=torch.long)
        batch["x_t"] = x
        with torch.no_grad():
            mo = model(batch, t_model)
        x0 = mo["pred_x_start"]
        if cond_fn_with_grad and i <= 3:                                               # :580 (respaced index)
            ab_ = _coef(tables.alphas_cumprod, i, dt)
            eps_ = (_coef(tables.sqrt_recip_alphas_cumprod, i, dt) * x - x0) / _coef(tables.sqrt_recipm1_alphas_cumprod, i, dt)
            g, _ = model.guide_coll(batch, mo, t_model, compute_grad="x_t", reduction=guide_reduction)   # :584
            eps_ = eps_ - (1 - ab_).sqrt() * g.to(dt) * 1.0                            # :585-586, scale = 1.0
            x0 = _coef(tables.sqrt_recip_alphas_cumprod, i, dt) * x - _coef(tables.sqrt_recipm1_alphas_cumprod, i, dt) * eps_   # :587
        eps = (_coef(tables.sqrt_recip_alphas_cumprod, i, dt) * x - x0) / _coef(tables.sqrt_recipm1_alphas_cumprod, i, dt)  # :286-290
        ab = _coef(tables.alphas_cumprod, i, dt)
        abp = _coef(tables.alphas_cumprod_prev, i, dt)
        sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)      # :541-545
        mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps           # :548-551
        nz = 0.0 if i == 0 else 1.0
        x = mean + nz * sigma * noise[1 + k]                                           # :555
        out = {"sample": x, "pred_xstart": x0, "other_outputs": mo}
        if trace is not None:
            trace.append((x.clone(), x0.clone()))
    return out


def val_losses(model, batch, tables: Tables, noise, timestep_respacing="", cond_fn_with_grad=False,
               cond_grad_weight=1.0, guide_reduction="mean", trace=None, guide_all_points=False):
    """gaussian_diffusion.py:749-780 with compute_loss=False: returns the final step's model dict."""
    model.validation_setup()
    if timestep_respacing == "":
        o = p_sample_loop(model, batch, tables, noise, cond_fn_with_grad, cond_grad_weight, guide_reduction, trace, guide_all_points)
    elif timestep_respacing[0:4] == "ddim":
        o = ddim_sample_loop(model, batch, tables, noise, 0.0, trace, cond_fn_with_grad, guide_reduction)
    else:
        raise SystemExit("timestep_respacing_eval not setup correctly")                # :774-775
    return o["other_outputs"]
