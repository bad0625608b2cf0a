"""Teacher-forced checks of one guided denoising step, shared by the GPU test modules.

Adam's first step delta = -lr g / (|g| + 1e-8) is sign-like: two correct fp32 implementations of the gradient agree to a
tolerance, not to the bit, and where |g| is within that tolerance of zero the step may legitimately land anywhere in
[-lr, +lr].  Instead of allowing a fraction of elements to be arbitrarily wrong at the end of a chain (where one such flip
has been amplified by the remaining steps and has changed later gradients), every step is checked on IDENTICAL inputs and
EVERY element gets a bound: rounding of the mean + what the asserted gradient tolerance allows Adam to do at that element
(`oracle.adam_step_budget`: ~0 wherever |g| >> tolerance, at most 2 lr).  No element is exempt."""
import numpy as np
import torch

GRAD_RTOL = 2e-5          # of max|g| of the step: the bar test_guidance_step_golden holds the gradient to (O(1) latents)
KINK_TOL = 5e-5           # m/s: an agent whose speed chain comes this close to a kink of the loss (|v - target|, the clips) is held
                          # to a bound, not to GRAD_RTOL -- the decoded speeds themselves agree to ~2e-6 (bar 1e-4)
GRAD_COND = 4.0           # ... or this many times the distance of the fp32 ORACLE's gradient from the fp64 oracle's, if larger


def grad_tolerance(O, wd, mean, cond, gd_ref, lr, g32):
    """Tolerance for dL/dmean at this mean.  Late in a random-init chain the latents reach ~1e4, the decoder's layer-0 gates
    sit at sigma(x) = 1 - 1e-5 .. 1 - 1e-7, and their derivative sigma (1 - sigma) is known in fp32 only to ulp(1) / (1 - sigma)
    = 1e-2 .. 1: the gradient is ill-conditioned IN FP32, whoever computes it.  The fp64 oracle is the truth; the GPU has to
    be as close to it as the fp32 oracle is (times GRAD_COND), or within GRAD_RTOL where the problem is well conditioned.
    -> (fp64 gradient, tolerance, mask [B] of agents sitting on a kink of the loss: `oracle.speed_kink_margin` < KINK_TOL)."""
    d = lambda t: None if t is None else t.double()
    wd64 = {k: v.double() for k, v in wd.items()}
    _, g64 = O.guidance_step(wd64, mean.double(), cond.double(), gd_ref["curr_states"].double(), d(gd_ref.get("target_speed")),
                             d(gd_ref.get("loss_scale")), lr, None, gd_ref.get("optimizer", "adam"))
    cond_err = float((g32.double() - g64).abs().max())
    margin = O.speed_kink_margin(wd64, mean.double(), cond.double(), gd_ref["curr_states"].double(), gd_ref["target_speed"].double())
    return g64, max(GRAD_RTOL * float(g64.abs().max()), GRAD_COND * cond_err), margin < KINK_TOL


def check_guided_step(engine, O, w, wd, x_t, cond, non_cond, cfg_w, gd_gpu, gd_ref, i, z, tag=""):
    """One loop iteration at timestep i on x_t, three comparisons, all elements:
      (1) cld_sample_step's posterior mean vs the oracle's                          <= 1e-4 * max(1, max|mean|)
      (2) cld_guidance_step on the ORACLE's mean: gradient vs the fp64 oracle      <= grad_tolerance (2e-5 * max|g| where well conditioned)
          guided mean vs the oracle's: <= 4 ulp of the mean + adam_step_budget(g_ref, lr, gradient tolerance)  (SGD: lr * tolerance)
      (3) cld_sample_step's own guided mean / x_next == cld_guidance_step on ITS mean, bit for bit (one kernel, same input)
    -> the step's outputs (GPU tensors) and how much the Adam budget concedes: the share of elements whose budget exceeds 1e-3,
    or 0 when even the largest possible step difference (2 lr) is below 1e-3 of max|mean| (late in a chain whose latents have
    grown to ~1e4 the decoder saturates, gradients vanish and many elements sit near the kink -- but a flip there moves the
    mean by 7e-5 of its scale)."""
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
        g64, gtol, kink = grad_tolerance(O, wd, ref["mean"], cond.cpu(), gd_ref, lr, ref["grad"])
        assert float(kink.float().mean()) <= 0.01, (tag, i, "kink-adjacent agents", int(kink.sum()))
        ok = ~kink
        gdiff = (gr.cpu().double() - g64).abs()
        gerr = float(gdiff[ok].max())
        assert gerr <= gtol, (tag, i, "gradient", gerr, gtol, float(g64.abs().max()))
        assert float(gdiff.max()) <= 2.0 * float(g64.abs().max()), (tag, i, "gradient of a kink-adjacent agent: one flipped term at most")
        gtol *= 1.0 + 1.0 / GRAD_COND          # the guided mean is compared with the fp32 oracle's: |g_gpu - g_32| <= |g_gpu - g_64| + |g_64 - g_32|
        if opt == "adam":
            budget = O.adam_step_budget(ref["grad"], lr, gtol)
            budget[kink] = 2.0 * lr             # any sign may differ there; the step itself is bounded by lr
        else:
            budget = torch.full_like(ref["grad"], lr * gtol)
            budget[kink] = lr * 2.0 * float(g64.abs().max())
        ulp = 4 * 1.2e-7 * mscale
        over = (mg.cpu() - ref["mean_guided"]).abs() - budget - ulp
        assert float(over.max()) <= 0.0, (tag, i, "guided mean beyond rounding + Adam budget", float(over.max()))
        over = (xn.cpu() - ref["x_next"]).abs() - budget - 2 * ulp
        assert float(over.max()) <= 0.0, (tag, i, "x_next", float(over.max()))
        print(f"   [{tag} t={i}] max|mean| {mscale:.3e}; gradient: max|g| {float(g64.abs().max()):.3e}, GPU vs fp64 oracle {gerr:.2e} (tolerance {gtol:.2e}), "
              f"kink-adjacent agents {int(kink.sum())}")
        share = 0.0 if 2.0 * lr <= 1e-3 * mscale else float((budget > 1e-3).float().mean())
        # (3) the loop iteration is exactly head -> guidance kernel: same kernel on its own mean reproduces it
        mg2, xn2 = engine.guidance_step(got["mean"], cond, dict(gd_gpu, lr=lr), sigma=got["sigma"], z=z)
        assert torch.equal(mg2, got["mean_guided"]) and torch.equal(xn2, got["x_next"]), (tag, i, "sample_step != head + guidance_step")
    else:
        xerr = float((got["x_next"].cpu() - ref["x_next"]).abs().max())
        assert xerr <= 1e-4 * mscale, (tag, i, "x_next", xerr)
    return got, share
