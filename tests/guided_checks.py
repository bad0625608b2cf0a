"""Teacher-forced checks of one guided denoising step, shared by the GPU test modules.

Adam's first step delta = -lr g / (|g| + 1e-8) is sign-like: two correct fp32 implementations of the gradient agree to a
tolerance, not to the bit, and where |g| is within that tolerance of zero the step may legitimately land anywhere in
[-lr, +lr].  Instead of allowing a fraction of elements to be arbitrarily wrong at the end of a chain (where one such flip
has been amplified by the remaining steps and has changed later gradients), every step is checked on IDENTICAL inputs and
EVERY element gets a bound: rounding of the mean + what the asserted gradient tolerance allows Adam to do at that element
(`oracle.adam_step_budget`: ~0 wherever |g| >> tolerance, at most 2 lr).  No element is exempt."""
import numpy as np
import torch

GRAD_RTOL = 2e-5          # of max|g| of the step: the bar test_guidance_step_golden holds the gradient to


def check_guided_step(engine, O, w, wd, x_t, cond, non_cond, cfg_w, gd_gpu, gd_ref, i, z, tag=""):
    """One loop iteration at timestep i on x_t, three comparisons, all elements:
      (1) cld_sample_step's posterior mean vs the oracle's                          <= 1e-4 * max(1, max|mean|)
      (2) cld_guidance_step on the ORACLE's mean: gradient                         <= GRAD_RTOL * max|g|
          guided mean vs the oracle's: <= 4 ulp of the mean + adam_step_budget(g_ref, lr, gradient tolerance)  (SGD: lr * tolerance)
      (3) cld_sample_step's own guided mean / x_next == cld_guidance_step on ITS mean, bit for bit (one kernel, same input)
    -> the step's outputs (GPU tensors) and the share of elements whose Adam budget exceeds 1e-3."""
    got = engine.sample_step(x_t, cond, i, z=z, non_cond=non_cond, guidance_w=cfg_w, guidance=gd_gpu, want_grad=True)
    ref = O.sample_step(w, wd, O.schedule(engine.n_timesteps), x_t.cpu(), cond.cpu(), i, None if z is None else z.cpu(),
                        None if non_cond is None else non_cond.cpu(), cfg_w, gd_ref)
    mscale = max(1.0, float(ref["mean"].abs().max()))
    err = float((got["mean"].cpu() - ref["mean"]).abs().max())
    assert err <= 1e-4 * mscale, (tag, i, "posterior mean", err, mscale)
    assert abs(got["sigma"] - ref["sigma"]) <= 2e-6 * ref["sigma"]
    share = 0.0
    if i > 0 and gd_gpu is not None:
        lr = gd_ref.get("lr") or got["sigma"]
        opt = gd_ref.get("optimizer", "adam")
        mg, xn, gr = engine.guidance_step(ref["mean"], cond, dict(gd_gpu, lr=lr), sigma=got["sigma"], z=z, want_grad=True)
        gtol = GRAD_RTOL * float(ref["grad"].abs().max())
        gerr = float((gr.cpu() - ref["grad"]).abs().max())
        assert gerr <= gtol, (tag, i, "gradient", gerr, gtol)
        if opt == "adam":
            budget = O.adam_step_budget(ref["grad"], lr, gtol)
        else:
            budget = torch.full_like(ref["grad"], lr * gtol)
        ulp = 4 * 1.2e-7 * mscale
        over = (mg.cpu() - ref["mean_guided"]).abs() - budget - ulp
        assert float(over.max()) <= 0.0, (tag, i, "guided mean beyond rounding + Adam budget", float(over.max()))
        assert float((mg.cpu() - ref["mean_guided"]).abs().max()) <= 2.0 * lr + ulp if opt == "adam" else True
        over = (xn.cpu() - ref["x_next"]).abs() - budget - 2 * ulp
        assert float(over.max()) <= 0.0, (tag, i, "x_next", float(over.max()))
        share = float((budget > 1e-3).float().mean())
        # (3) the loop iteration is exactly head -> guidance kernel: same kernel on its own mean reproduces it
        mg2, xn2 = engine.guidance_step(got["mean"], cond, dict(gd_gpu, lr=lr), sigma=got["sigma"], z=z)
        assert torch.equal(mg2, got["mean_guided"]) and torch.equal(xn2, got["x_next"]), (tag, i, "sample_step != head + guidance_step")
    else:
        xerr = float((got["x_next"].cpu() - ref["x_next"]).abs().max())
        assert xerr <= 1e-4 * mscale, (tag, i, "x_next", xerr)
    return got, share
