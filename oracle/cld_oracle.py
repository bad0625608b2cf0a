"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the CLD
latent-diffusion sampling path in plain fp32 PyTorch ops.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; the product package never does (it fails loudly when the HIP
library is missing instead of falling back to anything here).

Parity status: PINNED.  Every function below is checked in
`tests/test_oracle_golden.py` against golden vectors under `tests/golden/`,
which `oracle/make_golden.py` produced by importing the reference's own,
unmodified files from /root/reference in the build container (torch 2.10 CPU,
1 thread) with weights from `synth.py` pushed in through `load_state_dict`.

Each function cites the reference file:line it restates.  Nothing here is
copied from the reference: the reference is an `nn.Module` tree driven by
einops layers; this is a flat functional walk over a `state_dict`.

All tensors are torch CPU tensors; `dtype` may be float32 (the parity dtype) or
float64 (used only to report the fp32 rounding floor next to parity numbers).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# a-1  schedule buffers
# --------------------------------------------------------------------------- #
def cosine_betas(n: int, s: float = 0.008) -> Tensor:
    """src/tbsim/models/diffuser_helpers.py:451-462 -- float64 NumPy cosine
    schedule, clipped to [0, 0.999], then cast to fp32."""
    steps = n + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.tensor(np.clip(betas, a_min=0, a_max=0.999), dtype=torch.float32)


def schedule(n: int = 100) -> Dict[str, Tensor]:
    """models/dm/dm_model.py:29-56 -- the buffers the sampler reads
    (x_t_cof, noise_cof, posterior_log_variance_clipped) plus the ones they
    derive from, all computed in fp32 exactly in the reference's op order."""
    betas = cosine_betas(n)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, 0)
    ac_prev = torch.cat([torch.ones(1), ac[:-1]])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(torch.clamp(post_var, min=1e-20)),
        "x_t_cof": torch.sqrt(1.0 / alphas),
        "noise_cof": betas / torch.sqrt(alphas - ac * alphas),
    }


# --------------------------------------------------------------------------- #
# a-4  U-Net
# --------------------------------------------------------------------------- #
def sinusoidal_emb(t: Tensor, dim: int = 32, dtype=torch.float32) -> Tensor:
    """diffuser_helpers.py:25-32 -- [sin(t*w_k), cos(t*w_k)], w_k = exp(-k ln(1e4)/(dim/2-1))."""
    half = dim // 2
    w = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = t[:, None] * w[None, :]          # int64 * fp32 -> fp32, as in the reference
    return torch.cat((e.sin(), e.cos()), dim=-1).to(dtype)


def conv_block(x: Tensor, w: Dict[str, Tensor], p: str) -> Tensor:
    """diffuser_helpers.py:50-67 -- Conv1d(k, pad=k//2) -> GroupNorm(8, eps 1e-5) -> Mish."""
    cw = w[p + ".block.0.weight"]
    y = F.conv1d(x, cw, w[p + ".block.0.bias"], padding=cw.shape[-1] // 2)
    y = F.group_norm(y, 8, w[p + ".block.2.weight"], w[p + ".block.2.bias"], eps=1e-5)
    return F.mish(y)


def res_block(x: Tensor, tc: Tensor, w: Dict[str, Tensor], p: str) -> Tensor:
    """temporal.py:16-45 -- out = CB1(CB0(x) + Linear(Mish(tc))[:, :, None]) + res(x)."""
    bias = F.linear(F.mish(tc), w[p + ".time_mlp.1.weight"], w[p + ".time_mlp.1.bias"])
    h = conv_block(x, w, p + ".blocks.0") + bias[:, :, None]
    h = conv_block(h, w, p + ".blocks.1")
    rw = w.get(p + ".residual_conv.weight")
    r = x if rw is None else F.conv1d(x, rw, w[p + ".residual_conv.bias"])
    return h + r


def unet_forward(w: Dict[str, Tensor], x: Tensor, cond: Tensor, t: Tensor,
                 taps: Optional[dict] = None) -> Tensor:
    """temporal.py:122-180 -- x [B,52,4], cond [B,256], t [B] int64 -> eps [B,52,4].
    `taps`, when given, collects named intermediate activations ([B,C,L] layout)."""
    dt = x.dtype
    h = x.transpose(1, 2)                                   # temporal.py:139
    te = sinusoidal_emb(t, w["model.time_mlp.1.weight"].shape[1], dt)
    te = F.linear(te, w["model.time_mlp.1.weight"], w["model.time_mlp.1.bias"])
    te = F.linear(F.mish(te), w["model.time_mlp.3.weight"], w["model.time_mlp.3.bias"])
    tc = torch.cat([te, cond], dim=-1)                      # temporal.py:146
    skips = []
    n_down = 0
    while f"model.downs.{n_down}.0.time_mlp.1.weight" in w:
        n_down += 1
    for i in range(n_down):
        h = res_block(h, tc, w, f"model.downs.{i}.0")
        h = res_block(h, tc, w, f"model.downs.{i}.1")
        if taps is not None:
            taps[f"downs.{i}.1"] = h
        skips.append(h)
        dw = w.get(f"model.downs.{i}.2.conv.weight")
        if dw is not None:                                  # Downsample1d, diffuser_helpers.py:34-40
            h = F.conv1d(h, dw, w[f"model.downs.{i}.2.conv.bias"], stride=2, padding=1)
            if taps is not None:
                taps[f"downs.{i}.2"] = h
    h = res_block(h, tc, w, "model.mid_block1")
    h = res_block(h, tc, w, "model.mid_block2")
    if taps is not None:
        taps["mid_block2"] = h
    i = 0
    while f"model.ups.{i}.0.time_mlp.1.weight" in w:
        h = torch.cat((h, skips.pop()), dim=1)              # temporal.py:164
        h = res_block(h, tc, w, f"model.ups.{i}.0")
        h = res_block(h, tc, w, f"model.ups.{i}.1")
        if taps is not None:
            taps[f"ups.{i}.1"] = h
        # Upsample1d, diffuser_helpers.py:42-48: ConvTranspose1d(k4, s2, p1)
        h = F.conv_transpose1d(h, w[f"model.ups.{i}.2.conv.weight"],
                               w[f"model.ups.{i}.2.conv.bias"], stride=2, padding=1)
        if taps is not None:
            taps[f"ups.{i}.2"] = h
        i += 1
    h = conv_block(h, w, "model.final_conv.0")              # temporal.py:117-120
    if taps is not None:
        taps["final_conv.0"] = h
    h = F.conv1d(h, w["model.final_conv.1.weight"], w["model.final_conv.1.bias"])
    return h.transpose(1, 2)                                # temporal.py:176


# --------------------------------------------------------------------------- #
# a-3 / a-2 / a-8  DDPM update, sampling loop, log-prob
# --------------------------------------------------------------------------- #
def ddpm_step(w, sched, x: Tensor, cond: Tensor, i: int, z: Tensor):
    """dm_model.py:144-163 -- one ancestral step at timestep i with caller noise z.
    Returns (x_{t-1}, mean, sigma scalar tensor)."""
    B = x.shape[0]
    t = torch.full((B,), i, dtype=torch.long)
    eps = unet_forward(w, x, cond, t)
    dt = x.dtype
    mean = sched["x_t_cof"][i].to(dt) * x - sched["noise_cof"][i].to(dt) * eps
    sigma = (0.5 * sched["posterior_log_variance_clipped"][i].to(dt)).exp()
    nz = 0.0 if i == 0 else 1.0                             # nonzero_mask, dm_model.py:151
    return mean + (nz * sigma) * z, mean, sigma


def normal_log_prob_mean(x: Tensor, mean: Tensor, sigma: Tensor) -> Tensor:
    """torch.distributions.Normal(mean, sigma).log_prob(x).mean((1,2)), dm_model.py:130-132,170-173:
    -((x-mean)^2)/(2 sigma^2) - log(sigma) - log(sqrt(2 pi))."""
    var = sigma ** 2
    lp = -((x - mean) ** 2) / (2 * var) - sigma.log() - math.log(math.sqrt(2 * math.pi))
    return lp.mean(dim=(1, 2))


def sample(w, sched, x_T: Tensor, noise: Tensor, cond: Tensor, n_steps: Optional[int] = None, stride: int = 1) -> dict:
    """dm_model.py:103-142 -- loop over i in reversed(range(0, n, stride)) (:119; stride = 1 in the reference's ctor, :25);
    noise[s] feeds iteration s.  Returns pred_traj (x0), x1 (None unless step 1 is visited), log_prob_final."""
    n = int(sched["x_t_cof"].shape[0]) if n_steps is None else n_steps
    x = x_T
    x1 = None
    out = {}
    for s, i in enumerate(reversed(range(0, n, stride))):
        x, mean, sigma = ddpm_step(w, sched, x, cond, i, noise[s])
        if i == 1:
            x1 = x.clone()
        if i == 0:
            out["pred_traj"] = x.clone()
            out["log_prob_final"] = normal_log_prob_mean(x, mean, sigma)
    out["x1"] = x1
    return out


def sample_cfg(w, sched, x_T: Tensor, noise: Tensor, cond: Tensor, non_cond: Tensor, guidance_w: float) -> dict:
    """Classifier-free-guided chain.  Not in CLD's DmModel; defined by the vendored upstream
    DiffuserModel.p_mean_variance, src/tbsim/models/diffuser.py:766-789: a second U-Net pass on
    aux_info['non_cond_feat'] and eps = (1 + w) eps_cond - w eps_uncond, then the DDPM update of
    dm_model.py:144-163 in the loop of dm_model.py:119-132."""
    n = int(sched["x_t_cof"].shape[0])
    x, x1 = x_T, None
    dt = x.dtype
    for s, i in enumerate(reversed(range(n))):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        eps = (1 + guidance_w) * unet_forward(w, x, cond, t) - guidance_w * unet_forward(w, x, non_cond, t)
        mean = sched["x_t_cof"][i].to(dt) * x - sched["noise_cof"][i].to(dt) * eps
        sigma = (0.5 * sched["posterior_log_variance_clipped"][i].to(dt)).exp()
        x = mean + ((0.0 if i == 0 else 1.0) * sigma) * noise[s]
        if i == 1:
            x1 = x.clone()
    return {"pred_traj": x, "x1": x1}


def q_sample(sched, x0: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """dm_model.py:91-96 -- sqrt(acp[t]) x0 + sqrt(1 - acp[t]) noise, per-row t."""
    return sched["sqrt_alphas_cumprod"][t].view(-1, 1, 1) * x0 + sched["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1) * noise


def compute_losses(w, sched, z0: Tensor, cond: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """dm_model.py:82-89 with the draws (t, noise) supplied: F.mse_loss(noise, U-Net(q_sample(z0, t, noise), cond, t))."""
    return F.mse_loss(noise, unet_forward(w, q_sample(sched, z0, t, noise), cond, t))


def log_prob(w, sched, x_t: Tensor, x_tm1: Tensor, cond: Tensor, i: int) -> Tensor:
    """dm_model.py:165-174 -- log N(x_{t-1}; mean(x_t, eps), sigma_t) averaged over (T, D)."""
    B = x_t.shape[0]
    t = torch.full((B,), i, dtype=torch.long)
    eps = unet_forward(w, x_t, cond, t)
    dt = x_t.dtype
    mean = sched["x_t_cof"][i].to(dt) * x_t - sched["noise_cof"][i].to(dt) * eps
    sigma = (0.5 * sched["posterior_log_variance_clipped"][i].to(dt)).exp()
    return normal_log_prob_mean(x_tm1, mean, sigma)


# --------------------------------------------------------------------------- #
# a-5  LSTM-VAE decoder
# --------------------------------------------------------------------------- #
def lstm_decode(w: Dict[str, Tensor], z: Tensor, cond: Tensor) -> Tensor:
    """models/vae/lstm_vae.py:44-52 -- h0 = cond2hidden(cond) for both layers, c0 = 0,
    2-layer LSTM(4->64) (gate rows i,f,g,o), hid2act 64->2.  z [B,52,4] -> [B,52,2]."""
    B, T, _ = z.shape
    H = w["lstm_dec.lstm.weight_hh_l0"].shape[1]
    h0 = F.linear(cond, w["lstm_dec.cond2hidden.weight"], w["lstm_dec.cond2hidden.bias"])
    h = [h0.clone(), h0.clone()]
    c = [torch.zeros(B, H, dtype=z.dtype), torch.zeros(B, H, dtype=z.dtype)]
    outs = []
    for t in range(T):
        inp = z[:, t]
        for l in range(2):
            g = (F.linear(inp, w[f"lstm_dec.lstm.weight_ih_l{l}"], w[f"lstm_dec.lstm.bias_ih_l{l}"])
                 + F.linear(h[l], w[f"lstm_dec.lstm.weight_hh_l{l}"], w[f"lstm_dec.lstm.bias_hh_l{l}"]))
            gi, gf, gg, go = g.chunk(4, dim=1)
            c[l] = torch.sigmoid(gf) * c[l] + torch.sigmoid(gi) * torch.tanh(gg)
            h[l] = torch.sigmoid(go) * torch.tanh(c[l])
            inp = h[l]
        outs.append(inp)
    y = torch.stack(outs, dim=1)
    return F.linear(y, w["lstm_dec.hid2act.weight"], w["lstm_dec.hid2act.bias"])


# --------------------------------------------------------------------------- #
# a-6 / a-7  descale + unicycle roll-out
# --------------------------------------------------------------------------- #
NORM_MEAN = (13.162, -0.13891, 5.0223, -0.0046415, -0.0080072, -0.0013546)   # config.yaml:162
NORM_STD = (13.0717, 2.2462, 3.6187, 0.2210, 2.5770, 0.0840)                 # config.yaml:163
DYN = dict(acce_lo=-10.0, acce_hi=8.0, v_lo=-10.0, v_hi=30.0,                 # config.yaml:134-141,
           max_steer=0.5, max_yawvel=2 * math.pi, dt=0.1)                     # unicycle.py:8-19


def unicycle_parallel(cs: Tensor, act: Tensor, dyn: dict = DYN) -> Tensor:
    """diffuser_helpers.py:541-639, mode='parallel' -- cs [B,4]=(x,y,v,yaw), act [B,T,2]
    (acc, yaw-rate; descaled) -> [B,T,4].  The reference builds tril matrices and bmm's;
    the sums below are the same inclusive prefix sums (v is clipped AFTER the sum)."""
    dt = dyn["dt"]
    acc = act[..., 0].clamp(dyn["acce_lo"], dyn["acce_hi"])
    v_raw = torch.cumsum(torch.cat((cs[:, 2:3], acc * dt), dim=1), dim=1)      # [B,T+1]
    v = v_raw.clamp(dyn["v_lo"], dyn["v_hi"])
    v_avg = 0.5 * (v[:, :-1] + v[:, 1:])
    v_prev = v[:, :-1]
    yb = torch.minimum(dyn["max_steer"] * v_prev.abs(),
                       dyn["max_yawvel"] / v_prev.abs().clamp(min=0.1)).clamp(min=0.1)
    yr = torch.maximum(torch.minimum(act[..., 1], yb), -yb)
    yaw_full = torch.cumsum(torch.cat((cs[:, 3:4], yr * dt), dim=1), dim=1)
    yaw_prev = yaw_full[:, :-1]
    vx = v_avg * torch.cos(yaw_prev)
    vy = v_avg * torch.sin(yaw_prev)
    xs = torch.cumsum(torch.cat((cs[:, 0:1], vx * dt), dim=1), dim=1)[:, 1:]
    ys = torch.cumsum(torch.cat((cs[:, 1:2], vy * dt), dim=1), dim=1)[:, 1:]
    return torch.stack((xs, ys, v[:, 1:], yaw_full[:, 1:]), dim=-1)


def action_to_state_and_action(act_scaled: Tensor, cs: Tensor, scaled_input: bool = True,
                               descaled_output: bool = False) -> Tensor:
    """models/vae/vae_model.py:100-129 with the (x-mean)/std convention of :131-173
    (restated: the reference's scale/descale raise on CPU, `get_device()` = -1)."""
    mean = torch.tensor(NORM_MEAN, dtype=act_scaled.dtype)
    std = torch.tensor(NORM_STD, dtype=act_scaled.dtype)
    a = act_scaled * std[4:6] + mean[4:6] if scaled_input else act_scaled
    st = unicycle_parallel(cs, a)
    out = torch.cat((st, a), dim=-1)
    if scaled_input and not descaled_output:
        out = (out - mean) / std
    return out


def decode(wdec, z: Tensor, cond: Tensor, cs: Tensor, descaled_output: bool = True) -> Tensor:
    """guide_dm_trainer.py:97-98 -- lstm_dec then convert_action_to_state_and_action -> [B,52,6]."""
    return action_to_state_and_action(lstm_decode(wdec, z, cond), cs, True, descaled_output)


# --------------------------------------------------------------------------- #
# f-4  VAE encoder: state -> state+action, LSTM encoder, reparametrisation
# --------------------------------------------------------------------------- #
def state_to_state_and_action(pos: Tensor, yaw: Tensor, speed: Tensor, dt: float = 0.1, scaled: bool = False) -> Tensor:
    """src/tbsim/models/diffuser_helpers.py:685-749 as called at models/context_utils.py:64-70:
    pos [B,T,2], yaw [B,T,1], speed [B] -> [B,T,6] = (x, y, v, yaw, acc, yaw-rate); positions and yaw are
    pre-padded with zeros, speed with curr_speed; yaw differences wrap through a floored modulo."""
    B = pos.shape[0]
    p = torch.cat((torch.zeros(B, 1, 2, dtype=pos.dtype), pos), dim=1)
    y = torch.cat((torch.zeros(B, 1, 1, dtype=pos.dtype), yaw), dim=1)
    vel = ((p[:, 1:, 0:1] - p[:, :-1, 0:1]) / dt * torch.cos(y[:, 1:]) +
           (p[:, 1:, 1:2] - p[:, :-1, 1:2]) / dt * torch.sin(y[:, 1:]))
    vel = torch.cat((speed[:, None, None], vel), dim=1)
    acc = (vel[:, 1:] - vel[:, :-1]) / dt
    period = 2 * np.pi
    d = (y[:, 1:] - y[:, :-1] + period / 2) % period - period / 2
    d = torch.where(d > np.pi, d - 2 * np.pi, d)
    out = torch.cat((p[:, 1:], vel[:, 1:], y[:, 1:], acc, d / dt), dim=-1)
    if scaled:                                              # VaeModel.scale_traj, vae_model.py:152
        out = (out - torch.tensor(NORM_MEAN, dtype=out.dtype)) / torch.tensor(NORM_STD, dtype=out.dtype)
    return out


def traj2z(w: Dict[str, Tensor], x6: Tensor, cond: Tensor, noise: Optional[Tensor]):
    """models/vae/lstm_vae.py:87-99 (Encoder :6-26): 2-layer LSTM(6->64), h0 = cond2hidden(cond), c0 = 0,
    mu / logvar = Linear(64->4)(outputs), z = mu + noise * exp(0.5 logvar)."""
    B, T, _ = x6.shape
    H = w["lstm_enc.lstm.weight_hh_l0"].shape[1]
    h0 = F.linear(cond, w["lstm_enc.cond2hidden.weight"], w["lstm_enc.cond2hidden.bias"])
    h = [h0.clone(), h0.clone()]
    c = [torch.zeros(B, H, dtype=x6.dtype), torch.zeros(B, H, dtype=x6.dtype)]
    outs = []
    for t in range(T):
        inp = x6[:, t]
        for l in range(2):
            g = (F.linear(inp, w[f"lstm_enc.lstm.weight_ih_l{l}"], w[f"lstm_enc.lstm.bias_ih_l{l}"])
                 + F.linear(h[l], w[f"lstm_enc.lstm.weight_hh_l{l}"], w[f"lstm_enc.lstm.bias_hh_l{l}"]))
            gi, gf, gg, go = g.chunk(4, dim=1)
            c[l] = torch.sigmoid(gf) * c[l] + torch.sigmoid(gi) * torch.tanh(gg)
            h[l] = torch.sigmoid(go) * torch.tanh(c[l])
            inp = h[l]
        outs.append(inp)
    y = torch.stack(outs, dim=1)
    mu = F.linear(y, w["mu.weight"], w["mu.bias"])
    lv = F.linear(y, w["logvar.weight"], w["logvar.bias"])
    z = mu if noise is None else mu + noise * torch.exp(0.5 * lv)
    return z, mu, lv


# --------------------------------------------------------------------------- #
# helpers for tests / bench
# --------------------------------------------------------------------------- #
# --------------------------------------------------------------------------- #
# f-1  ContextEncoder (models/context_utils.py:8-61)
# --------------------------------------------------------------------------- #
CTX = "context_encoder."
RESNET = CTX + "map_encoder.encoder_heads.map_model."


def mlp_ln(w: Dict[str, Tensor], p: str, x: Tensor, n_hidden: int) -> Tensor:
    """src/tbsim/models/base_models.py:21-96 with normalization=True: n_hidden x (Linear -> LayerNorm -> ReLU), then a
    last Linear; `_model` is the nn.Sequential, so the state_dict indices step by 3."""
    i = 0
    for _ in range(n_hidden):
        x = F.linear(x, w[f"{p}._model.{i}.weight"], w[f"{p}._model.{i}.bias"])
        x = F.layer_norm(x, (x.shape[-1],), w[f"{p}._model.{i + 1}.weight"], w[f"{p}._model.{i + 1}.bias"], 1e-5)
        x = F.relu(x)
        i += 3
    return F.linear(x, w[f"{p}._model.{i}.weight"], w[f"{p}._model.{i}.bias"])


def _bn(w, p, x):
    # eval-mode BatchNorm2d (running statistics, eps 1e-5): torchvision resnet.py BasicBlock / ResNet._forward_impl
    return F.batch_norm(x, w[p + ".running_mean"], w[p + ".running_var"], w[p + ".weight"], w[p + ".bias"], False, 0.0, 1e-5)


def resnet18_features(w: Dict[str, Tensor], image: Tensor, taps: Optional[dict] = None) -> Tensor:
    """The map_model of RasterizedMapEncoder (src/tbsim/models/base_models.py:559-614): torchvision 0.20 `resnet18()`
    (third-party, absent here: restated from its published definition, torchvision/models/resnet.py -- conv1 7x7/2 ->
    BN -> ReLU -> maxpool 3x3/2 -> 4 stages of 2 BasicBlocks [conv3x3 -> BN -> ReLU -> conv3x3 -> BN, + identity or
    1x1/2 conv + BN, ReLU] -> adaptive avg-pool -> fc) with conv1 replaced by Conv2d(34, 64, 7, 2, 3, bias=False) and
    fc by Linear(512, 256).  MapEncoder reads the 'map_model.fc' node of the feature extractor
    (diffuser_helpers.py:313-341), i.e. the fc output BEFORE RasterizedMapEncoder's output ReLU.  Parity of this
    function is UNPINNED (no torchvision to run the reference's own module)."""
    r = RESNET
    x = F.conv2d(image, w[r + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(w, r + "bn1", x))
    if taps is not None:
        taps["stem"] = x
    x = F.max_pool2d(x, 3, 2, 1)
    for li in range(1, 5):
        for b in range(2):
            p = f"{r}layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            idt = x
            y = F.relu(_bn(w, p + ".bn1", F.conv2d(x, w[p + ".conv1.weight"], None, stride=stride, padding=1)))
            y = _bn(w, p + ".bn2", F.conv2d(y, w[p + ".conv2.weight"], None, stride=1, padding=1))
            if (p + ".downsample.0.weight") in w:
                idt = _bn(w, p + ".downsample.1", F.conv2d(x, w[p + ".downsample.0.weight"], None, stride=stride))
            x = F.relu(y + idt)
        if taps is not None:
            taps[f"layer{li}"] = x
    x = x.mean(dim=(2, 3))
    return F.linear(x, w[r + "fc.weight"], w[r + "fc.bias"])


def context_encode(w: Dict[str, Tensor], image: Tensor, curr_states: Tensor, taps: Optional[dict] = None) -> Tensor:
    """ContextEncoder.forward (models/context_utils.py:40-61): cond_feat = process_cond_mlp(
    [agent_state_encoder(curr_states) | map_encoder(image)]) -> [B,256].  curr_states [B,4] = (x, y, v, yaw)
    as batch_utils.get_current_states builds them (src/tbsim/utils/batch_utils.py:46-65)."""
    sf = mlp_ln(w, CTX + "agent_state_encoder", curr_states, 2)
    mf = resnet18_features(w, image, taps)
    if taps is not None:
        taps["state_feat"], taps["map_feat"] = sf, mf
    return mlp_ln(w, CTX + "process_cond_mlp", torch.cat([sf, mf], dim=-1), 4)


# --------------------------------------------------------------------------- #
# f-3  sampling-time guidance (upstream diffuser.py:844-929, guidance_loss.py:219-254,2221-2282)
# --------------------------------------------------------------------------- #
def guidance_step(wdec, mean: Tensor, cond: Tensor, cs: Tensor, target_speed: Optional[Tensor], loss_scale: Optional[Tensor],
                  lr: float, perturb_th: Optional[float], optimizer: str = "adam", speed_limit=None, acc_limit=None, target_pos=None,
                  collision: Optional[dict] = None, grad_steps: int = 1, num_samp: int = 1, map_collision: Optional[dict] = None,
                  trace: Optional[list] = None):
    """One PerturbationGuidance.perturb call (guidance_loss.py:2221-2282) with decoder = `decode` and
    TargetSpeedLoss (:219-254): L = sum_b loss_scale[b] * sum_t |v_t - target| (+ optional SpeedLimitLoss / AccLimitLoss
    terms, each a (limit, per-agent scale) pair; + `collision`: the agent_collision configs, see scene_collision_total).
    grad_steps optimiser steps with the optimiser's state carried across them (:2247-2252: one torch.optim.Adam / SGD per
    call): Adam with torch's defaults (betas 0.9 / 0.999, eps 1e-8, bias-corrected: m_k / (1 - 0.9^k) over
    sqrt(v_k / (1 - 0.999^k)) + eps), whose FIRST step is -lr * g / (|g| + 1e-8); SGD's -lr * g.  perturb_th None = the
    reference's actual behaviour: its clip (:2275-2278) acts on x_guidance - x_initial, two names of one tensor
    (:2239), so it never changes anything (the golden vectors confirm); a number clips the accumulated step to +- that
    around the initial mean, as the code intends.  Returns (guided mean, gradient of the first step); `trace` (a list) receives every
    step's gradient."""
    def total(x):
        traj = decode(wdec, x, cond, cs, True)
        loss = traj.sum() * 0.0
        if target_speed is not None:
            dev = (traj[..., 2] - target_speed).abs()
            sc = loss_scale if loss_scale is not None else torch.full((mean.shape[0],), 1.0 / mean.shape[1], dtype=mean.dtype)
            loss = loss + (dev.sum(dim=1) * sc).sum()
        if speed_limit is not None:        # (limit, scale [B]): SpeedLimitLoss, guidance_loss.py:1509-1538
            loss = loss + ((traj[..., 2].abs() - speed_limit[0]).clamp(min=0).sum(dim=1) * speed_limit[1]).sum()
        if acc_limit is not None:          # (limit, scale [B]): AccLimitLoss, guidance_loss.py:1444-1467
            loss = loss + ((traj[..., 4].abs() - acc_limit[0]).clamp(min=0).sum(dim=1) * acc_limit[1]).sum()
        if target_pos is not None:         # (pos [B,2], time index [B], scale [B]): TargetPosAtTimeLoss, guidance_loss.py:632-670
            p_, t_, s_ = target_pos
            for bb in range(traj.shape[0]):        # t >= 0: hit at that step; t < 0: TargetPosLoss (:672-716), any step >= -(t + 1)
                tb = int(t_[bb])
                if tb >= 0:
                    loss = loss + (traj[bb, tb, :2] - p_[bb]).norm() * s_[bb]
                else:
                    e = traj[bb, -tb - 1:, :2] - p_[bb]
                    dist = e.norm(dim=-1)
                    loss = loss + (F.softmin(dist, dim=-1) * (e ** 2).sum(dim=-1)).mean() * s_[bb]
        if collision is not None:
            loss = loss + scene_collision_total(traj, collision, num_samp)
        if map_collision is not None:
            loss = loss + scene_map_collision_total(traj, map_collision, num_samp)
        return loss

    x = mean.clone()
    m, v = torch.zeros_like(x), torch.zeros_like(x)
    g_first = None
    for k in range(1, grad_steps + 1):
        xk = x.clone().requires_grad_(True)
        with torch.enable_grad():
            (g,) = torch.autograd.grad(total(xk), xk)
        if g_first is None:
            g_first = g
        if trace is not None:
            trace.append(g.detach().clone())
        if optimizer == "adam":
            if grad_steps == 1:
                delta = -lr * g / (g.abs() + 1e-8)
            else:
                m = 0.9 * m + 0.1 * g
                v = 0.999 * v + 0.001 * g * g
                delta = -(lr / (1.0 - 0.9 ** k)) * m / (v.sqrt() / math.sqrt(1.0 - 0.999 ** k) + 1e-8)
        else:
            delta = -lr * g
        x = x + delta
        if perturb_th is not None:
            x = mean + (x - mean).clamp(-perturb_th, perturb_th)
    return x, g_first


def sample_guided(w, wdec, sched, x_T: Tensor, noise: Tensor, cond: Tensor, cs: Tensor, target_speed: Tensor,
                  loss_scale: Optional[Tensor] = None, lr: Optional[float] = 0.3, optimizer: str = "adam",
                  non_cond: Optional[Tensor] = None, guidance_w: float = 0.0, clip_to_sigma: bool = False,
                  output: Optional[dict] = None, taps: Optional[dict] = None) -> dict:
    """The ancestral loop of dm_model.py:119-132 with upstream's p_sample guidance (diffuser.py:844-929): steps t > 0
    perturb the posterior mean before the noise is added (lr None -> sigma_t); t = 0 is unguided unless `output` =
    dict(lr, optimizer) is given: upstream's apply_guidance_output with its final_step_opt_params (diffuser.py:877-880;
    the perturb_th of those never acts, see guidance_step).  `taps` (a dict) receives `grads`: the list of (timestep, dL/dmean)
    of every guided step -- the tests bound what Adam's sign-like step may do with them (`adam_step_budget`)."""
    n = sched["betas"].shape[0]
    x = x_T
    x1 = None
    if taps is not None:
        taps["grads"] = []
    for s_, i in enumerate(reversed(range(n))):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        eps = unet_forward(w, x, cond, t)
        if non_cond is not None:
            eps = (1 + guidance_w) * eps - guidance_w * unet_forward(w, x, non_cond, t)
        mean = sched["x_t_cof"][i] * x - sched["noise_cof"][i] * eps
        sigma = float((0.5 * sched["posterior_log_variance_clipped"][i]).exp())
        if i > 0:
            mean, g_ = guidance_step(wdec, mean, cond, cs, target_speed, loss_scale, sigma if lr is None else lr,
                                     sigma if clip_to_sigma else None, optimizer)
            if taps is not None:
                taps["grads"].append((i, g_))
        elif output is not None:
            mean, _ = guidance_step(wdec, mean, cond, cs, target_speed, loss_scale, output.get("lr", 0.3), None, output.get("optimizer", "adam"))
        x = mean + (0.0 if i == 0 else sigma) * noise[s_]
        if i == 1:
            x1 = x.clone()
    return {"pred_traj": x, "x1": x1}


def sample_step(w, wdec, sched, x_t: Tensor, cond: Tensor, i: int, z: Optional[Tensor], non_cond: Optional[Tensor] = None,
                guidance_w: float = 0.0, guidance: Optional[dict] = None) -> dict:
    """ONE iteration of the loops above at timestep i on a given x_t (upstream p_sample, diffuser.py:844-929; with CFG and
    guidance off it is DmModel.x_Tminus1, dm_model.py:144-156).  guidance = dict(curr_states, target_speed, loss_scale | None,
    lr | None, optimizer, grad_steps = 1, collision | None, guide_clean = False).  guide_clean: upstream's `guide_clean=True`
    (diffuser.py:866-873): the steps act on the model's clean prediction x0_hat = sqrt(1 / acp) x_t - sqrt(1 / acp - 1) eps
    (predict_start_from_noise, :710-719) and x_next is the guided x0_hat + sigma z; `mean` then reports x0_hat.
    -> dict(mean [before guidance], sigma, x_next, and on a guided step mean_guided, grad)."""
    t = torch.full((x_t.shape[0],), i, dtype=torch.long)
    eps = unet_forward(w, x_t, cond, t)
    if non_cond is not None:
        eps = (1 + guidance_w) * eps - guidance_w * unet_forward(w, x_t, non_cond, t)
    mean = sched["x_t_cof"][i] * x_t - sched["noise_cof"][i] * eps
    sigma = float((0.5 * sched["posterior_log_variance_clipped"][i]).exp())
    if guidance is not None and i > 0 and guidance.get("guide_clean"):
        acp = sched["alphas_cumprod"][i]
        mean = torch.sqrt(1.0 / acp) * x_t - torch.sqrt(1.0 / acp - 1) * eps
    out = {"mean": mean, "sigma": sigma}
    m = mean
    if guidance is not None and i > 0:
        lr = guidance.get("lr")
        m, g = guidance_step(wdec, mean, cond, guidance["curr_states"], guidance.get("target_speed"), guidance.get("loss_scale"),
                             sigma if lr is None else lr, None, guidance.get("optimizer", "adam"),
                             collision=guidance.get("collision"), grad_steps=guidance.get("grad_steps", 1))
        out["mean_guided"], out["grad"] = m, g
    out["x_next"] = m if (i == 0 or z is None) else m + sigma * z
    return out


def speed_kink_margin(wdec, mean: Tensor, cond: Tensor, cs: Tensor, target_speed: Tensor, dyn: dict = DYN) -> Tensor:
    """Per agent [B]: how close the decoded speed chain of `mean` comes to a point where the target-speed guidance loss is not
    differentiable -- |v_t - target_t| (the loss's own kink, guidance_loss.py:219-254), the speed clip and the acceleration clip
    of the roll-out (diffuser_helpers.py:557-559,573-576).  Two correct implementations that disagree in the last bits of v
    may land on different sides of such a point and then differ by a whole term of the gradient; the tests hold agents
    with a margin below their trajectory tolerance to a bound instead of to 2e-5 (test infrastructure)."""
    std = torch.tensor(NORM_STD, dtype=mean.dtype)
    mu = torch.tensor(NORM_MEAN, dtype=mean.dtype)
    acc = lstm_decode(wdec, mean, cond)[..., 0] * std[4] + mu[4]
    v_raw = torch.cumsum(torch.cat((cs[:, 2:3], acc.clamp(dyn["acce_lo"], dyn["acce_hi"]) * dyn["dt"]), dim=1), dim=1)[:, 1:]
    v = v_raw.clamp(dyn["v_lo"], dyn["v_hi"])
    m = torch.minimum((v - target_speed).abs(), torch.minimum((v_raw - dyn["v_lo"]).abs(), (v_raw - dyn["v_hi"]).abs()))
    m = torch.minimum(m, torch.minimum((acc - dyn["acce_lo"]).abs(), (acc - dyn["acce_hi"]).abs()) * dyn["dt"])
    return m.amin(dim=1)


def adam_step_budget(g: Tensor, lr: float, grad_tol: float) -> Tensor:
    """How far Adam's first step delta(g) = -lr g / (|g| + 1e-8) (guidance_step) can move when the gradient is only known to
    +-grad_tol: delta is monotone in g, so the worst case over [g - tol, g + tol] sits at an end point.  ~0 wherever
    |g| >> grad_tol (the step is -lr sign(g) whatever the rounding), up to 2 lr where |g| <~ grad_tol (a sign flip).  Test
    infrastructure: two correct fp32 implementations of the gradient agree to grad_tol, not to the bit."""
    f = lambda v: v / (v.abs() + 1e-8)
    return lr * torch.maximum(f(g + grad_tol) - f(g), f(g) - f(g - grad_tol))


# --------------------------------------------------------------------------- #
# f-2  policy surface: guide-loss values, sample selection, world update, closed loop
# --------------------------------------------------------------------------- #
def transform_agents_to_world(pos: Tensor, yaw: Tensor, world_from_agent: Tensor):
    """src/tbsim/utils/geometry_utils.py:458-483 -- [B,N,T,2] / [B,N,T,1] in the agent frames -> world positions and headings
    (the heading through atan2 of the transformed unit vector, as upstream writes it)."""
    R, t = world_from_agent[:, None, None, :2, :2], world_from_agent[:, None, None, :2, 2]
    pos_w = (R @ pos.unsqueeze(-1)).squeeze(-1) + t
    hvec = torch.cat([torch.cos(yaw), torch.sin(yaw)], dim=-1)
    hw = (R @ hvec.unsqueeze(-1)).squeeze(-1)                      # (W h + origin) - origin
    return pos_w, torch.atan2(hw[..., 1], hw[..., 0]).unsqueeze(-1)


def agent_collision_loss(x: Tensor, extent: Tensor, world_from_agent: Tensor, curr_speed: Tensor, scene_index: Tensor,
                         agt_mask: Optional[Tensor] = None, num_disks: int = 5, buffer_dist: float = 0.2, decay_rate: float = 0.9,
                         moving_speed_th: float = 0.5, excluded_agents=None) -> Tensor:
    """AgentCollisionLoss.forward (src/tbsim/utils/guidance_loss.py:506-630) on x [B,N,T,6] = (x, y, v, yaw, acc, yaw-rate) in
    the agent frames -> [B,N] (rows of agt_mask when given).  Every agent is `num_disks` disks of radius width / 2 along its
    axis (:481-492); two agents of one scene collide at a step when their closest disk centres are within r_i + r_j +
    buffer_dist, the penalty is 1 - dist / that bound, weighted by decay_rate ** t (normalised), summed over the steps and
    AVERAGED over all B columns (:621), zero for agents slower than moving_speed_th (which also receive no gradient, :512-516;
    nor do agents outside agt_mask, :523-534).  Sample n of every agent lives in scene copy n.  `excluded_agents` (batch indices,
    :447,586-593): a pair whose agents are both listed is not penalised."""
    B, N, T_, _ = x.shape
    moving = curr_speed.abs() > moving_speed_th
    x = torch.where(moving.view(B, 1, 1, 1), x, x.detach())
    pos_w, yaw_w = transform_agents_to_world(x[..., :2], x[..., 3:4], world_from_agent)
    if agt_mask is not None:
        m = agt_mask.view(B, 1, 1, 1)
        pos_w, yaw_w = torch.where(m, pos_w, pos_w.detach()), torch.where(m, yaw_w, yaw_w.detach())
    rad = extent[:, 1] / 2.0
    cmin, cmax = -(extent[:, 0] / 2.0) + rad, (extent[:, 0] / 2.0) - rad
    cx = torch.stack([torch.linspace(float(cmin[b]), float(cmax[b]), num_disks) for b in range(B)]).to(x.dtype)   # as init_disks builds them
    cent = pos_w.unsqueeze(-2) + cx.view(B, 1, 1, num_disks, 1) * torch.cat([torch.cos(yaw_w), torch.sin(yaw_w)], dim=-1).unsqueeze(-2)   # [B,N,T,D,2]
    pen_d = rad.view(B, 1) + rad.view(1, B) + buffer_dist
    same = (scene_index.view(B, 1) == scene_index.view(1, B)) & ~torch.eye(B, dtype=torch.bool)
    c = cent.permute(2, 1, 0, 3, 4)                                                      # [T,N,B,D,2]
    d = (c[:, :, :, None, :, None, :] - c[:, :, None, :, None, :, :]).norm(dim=-1)        # [T,N,B,B,D,D]
    pair = d.reshape(T_, N, B, B, num_disks * num_disks).min(dim=-1)[0]
    if excluded_agents is not None:
        ex = torch.zeros(B, dtype=torch.bool)
        ex[torch.as_tensor(list(excluded_agents), dtype=torch.long)] = True
        same = same & ~(ex.view(B, 1) & ex.view(1, B))
    hit = (pair <= pen_d) & same
    pen = torch.where(hit, 1.0 - pair / pen_d, torch.zeros_like(pair))
    wts = torch.tensor([decay_rate ** t for t in range(T_)], dtype=x.dtype)
    wts = wts / wts.sum()
    out = (pen * wts.view(T_, 1, 1, 1)).sum(0).mean(-1).transpose(0, 1)                  # [B,N]
    out = torch.where(moving.view(B, 1), out, torch.zeros_like(out))
    return out if agt_mask is None else out[agt_mask]


def map_collision_loss(x: Tensor, extent: Tensor, raster_from_agent: Tensor, drivable_map: Tensor, curr_speed: Tensor,
                       num_points_lw=(10, 10), decay_rate: float = 0.9, moving_speed_th: float = 0.5) -> Tensor:
    """MapCollisionLoss.forward (src/tbsim/utils/guidance_loss.py:772-875) on x [B,N,T,6] -> [B,N].  Every agent box is sampled
    on a num_points_lw grid (:731-735, scaled by length / width, rotated and shifted by the plan's pose, :745-753); a point is
    off road where the drivable map is 0 at its raster pixel (coordinates truncated toward zero and clamped to the map,
    :797-805; metrics.py:451-481).  Only steps where SOME but not all points are off road count (:807-809); there every off-road
    point contributes 1 - (distance to the nearest on-road point of the box) / (box diagonal), the on-road points carrying the
    gradient and the off-road ones detached (:833-848).  Steps are weighted by decay_rate ** t (normalised) and summed; agents
    slower than moving_speed_th contribute nothing."""
    B, N, T_, _ = x.shape
    lw = extent[:, :2]
    lwise, wwise = torch.linspace(-0.5, 0.5, num_points_lw[0]), torch.linspace(-0.5, 0.5, num_points_lw[1])
    loc = torch.cartesian_prod(lwise, wwise).to(x.dtype)                     # [P,2]
    P = loc.shape[0]
    locs = loc[None] * lw[:, None, :]                                        # [B,P,2]
    yaw, pos = x[..., 3], x[..., :2]
    c, s_ = torch.cos(yaw)[..., None], torch.sin(yaw)[..., None]             # [B,N,T,1]
    lx, wy = locs[:, None, None, :, 0], locs[:, None, None, :, 1]            # [B,1,1,P]
    pts = torch.stack([lx * c - wy * s_, lx * s_ + wy * c], dim=-1) + pos[..., None, :]     # [B,N,T,P,2] agent frame
    Rm, tv = raster_from_agent[:, None, None, None, :2, :2], raster_from_agent[:, None, None, None, :2, 2]
    pix = ((Rm @ pts.unsqueeze(-1)).squeeze(-1) + tv).long()                 # truncation, as .long() does
    H, W = drivable_map.shape[-2:]
    px, py = pix[..., 0].clamp(0, W - 1), pix[..., 1].clamp(0, H - 1)
    bi = torch.arange(B).view(B, 1, 1, 1).expand(B, N, T_, P)
    off = ~(drivable_map[bi, py, px] != 0)                                   # [B,N,T,P]
    cnt = off.sum(dim=-1)
    overlap = (cnt != 0) & (cnt != P)
    diag = (lw * lw).sum(dim=-1).sqrt()
    d = (pts[..., :, None, :] - pts.detach()[..., None, :, :]).norm(dim=-1)   # rows carry the gradient, columns detached
    d = torch.where(off[..., :, None], torch.full_like(d, float("inf")), d)  # rows of off-road points masked out
    dmin = d.amin(dim=-2)                                                    # over the rows: nearest on-road point of every column
    per_pt = torch.where(off & overlap[..., None], 1.0 - dmin / diag.view(B, 1, 1, 1), torch.zeros_like(dmin))
    per_step = per_pt.sum(dim=-1)
    moving = curr_speed.abs() > moving_speed_th
    per_step = torch.where(moving.view(B, 1, 1), per_step, torch.zeros_like(per_step))
    wts = torch.tensor([decay_rate ** t for t in range(T_)], dtype=x.dtype)
    wts = wts / wts.sum()
    return (per_step * wts).sum(dim=-1)


def map_collision_grad_bounds(x: Tensor, extent: Tensor, raster_from_agent: Tensor, drivable_map: Tensor, curr_speed: Tensor,
                              row_coef: Tensor, num_points_lw=(10, 10), decay_rate: float = 0.9, moving_speed_th: float = 0.5,
                              tie_tol: float = 1e-4):
    """Test aid for the gradient of `map_collision_loss`: which of its elements are DEFINED by the loss and which depend on a tie.
    An off-road sample pulls on its nearest on-road sample; on a regular sample grid two or more on-road samples can be equidistant in
    exact arithmetic (mirror images), and torch.amin's backward then follows whichever rounding made smaller (or shares the gradient
    among bit-equal minima; the reference's torch.cdist adds ~1e-4 m of noise of its own).  For every (plan, step) this returns the
    interval [lo, hi] that d total / d (x, y, yaw) can take when every off-road sample may pull on ANY candidate within `tie_tol` metres
    of its minimum, or on any mix of them -- lo == hi where no sample of the step has such a tie -- and `tied` [B,N,T], the steps that
    have one.  Computed in float64 from the definition (:833-848), not by autograd.  x [B,N,T,6]; row_coef [B,N]: scene weight /
    (agents of the scene x samples).  -> (lo [B,N,T,3], hi [B,N,T,3], tied [B,N,T])."""
    x = x.double()
    B, N, T_, _ = x.shape
    lw = extent[:, :2].double()
    # the sample grid and the off-road flags exactly as map_collision_loss forms them (float32: a pixel boundary must fall the same way)
    x32 = x.float()
    lwise, wwise = torch.linspace(-0.5, 0.5, num_points_lw[0]), torch.linspace(-0.5, 0.5, num_points_lw[1])
    loc = torch.cartesian_prod(lwise, wwise)
    P = loc.shape[0]
    locs32 = loc[None] * extent[:, None, :2]
    yaw32, pos32 = x32[..., 3], x32[..., :2]
    c32, s32 = torch.cos(yaw32)[..., None], torch.sin(yaw32)[..., None]
    lx32, wy32 = locs32[:, None, None, :, 0], locs32[:, None, None, :, 1]
    pts32 = torch.stack([lx32 * c32 - wy32 * s32, lx32 * s32 + wy32 * c32], dim=-1) + pos32[..., None, :]
    Rm, tv = raster_from_agent[:, None, None, None, :2, :2], raster_from_agent[:, None, None, None, :2, 2]
    pix = ((Rm @ pts32.unsqueeze(-1)).squeeze(-1) + tv).long()
    H, W = drivable_map.shape[-2:]
    px, py = pix[..., 0].clamp(0, W - 1), pix[..., 1].clamp(0, H - 1)
    bi = torch.arange(B).view(B, 1, 1, 1).expand(B, N, T_, P)
    off = ~(drivable_map[bi, py, px] != 0)
    cnt = off.sum(dim=-1)
    overlap = (cnt != 0) & (cnt != P) & (curr_speed.abs() > moving_speed_th).view(B, 1, 1)
    locs = loc.double()[None] * lw[:, None, :]
    yaw, pos = x[..., 3], x[..., :2]
    c, s_ = torch.cos(yaw)[..., None], torch.sin(yaw)[..., None]
    lx, wy = locs[:, None, None, :, 0], locs[:, None, None, :, 1]
    pts = torch.stack([lx * c - wy * s_, lx * s_ + wy * c], dim=-1) + pos[..., None, :]          # [B,N,T,P,2]
    diag = (lw * lw).sum(dim=-1).sqrt().view(B, 1, 1, 1, 1)
    e = pts[..., :, None, :] - pts[..., None, :, :]                                             # [.., i, j, 2] = p_i - p_j
    d = e.norm(dim=-1)
    d = torch.where(off[..., :, None], torch.full_like(d, float("inf")), d)                     # candidates i must be on road
    dmin = d.amin(dim=-2, keepdim=True)
    cand = d <= dmin + tie_tol                                                                   # [.., i, j]
    use = off[..., None, :] & overlap[..., None, None]                                           # columns j: off-road samples of counted steps
    k = -1.0 / (d.clamp(min=1e-30) * diag)
    rel = pts - pos[..., None, :]                                                                # p_i - pos
    gx, gy = k * e[..., 0], k * e[..., 1]
    gw = k * (-e[..., 0] * rel[..., :, None, 1] + e[..., 1] * rel[..., :, None, 0])              # d p_i / d yaw = perp(p_i - pos)
    wts = torch.tensor([decay_rate ** t for t in range(T_)], dtype=torch.float64)
    wts = wts / wts.sum()
    sc = row_coef.double().view(B, N, 1) * wts.view(1, 1, T_)
    lo, hi = [], []
    inf = torch.full_like(d, float("inf"))
    for gcomp in (gx, gy, gw):
        cmin = torch.where(cand, gcomp, inf).amin(dim=-2)                                       # per column j
        cmax = torch.where(cand, gcomp, -inf).amax(dim=-2)
        z = torch.zeros_like(cmin)
        a_, b_ = torch.where(use.squeeze(-2), cmin, z).sum(dim=-1) * sc, torch.where(use.squeeze(-2), cmax, z).sum(dim=-1) * sc
        lo.append(torch.minimum(a_, b_)); hi.append(torch.maximum(a_, b_))
    tied = ((cand.sum(dim=-2) > 1) & use.squeeze(-2)).any(dim=-1)
    return torch.stack(lo, dim=-1), torch.stack(hi, dim=-1), tied


def scene_map_collision_total(traj: Tensor, mp: dict, num_samp: int = 1) -> Tensor:
    """What DiffuserGuidance.compute_guidance_loss (guidance_loss.py:2143-2172) adds for `map_collision` configs: sum over scenes
    of weight * mean over the scene's agents (and samples); mp: extent, raster_from_agent, drivable_map, curr_speed, scene_index,
    scene_weight [S] and the loss parameters."""
    BN = traj.shape[0]
    B = BN // num_samp
    x = traj.reshape(B, num_samp, traj.shape[1], 6)
    _, local = torch.unique_consecutive(mp["scene_index"], return_inverse=True)
    kw = {k: mp[k] for k in ("num_points_lw", "decay_rate", "moving_speed_th") if k in mp}
    vals = map_collision_loss(x, mp["extent"], mp["raster_from_agent"], mp["drivable_map"], mp["curr_speed"], **kw)
    tot = x.sum() * 0.0
    for si, wgt in enumerate(mp["scene_weight"]):
        if float(wgt) != 0.0:
            tot = tot + vals[local == si].mean() * float(wgt)
    return tot


def scene_collision_total(traj: Tensor, col: dict, num_samp: int = 1) -> Tensor:
    """What DiffuserGuidance.compute_guidance_loss (guidance_loss.py:2143-2172) adds to the total for `agent_collision` configs:
    sum over scenes of weight * mean over the scene's guided agents (and samples).  traj [B * num_samp, T, 6] sample-minor;
    col: extent, world_from_agent, curr_speed, scene_index, scene_weight [S] (0 = scene not guided), agents (optional dict
    scene -> local indices), and the loss parameters (incl. excluded_agents, batch indices)."""
    BN = traj.shape[0]
    B = BN // num_samp
    x = traj.reshape(B, num_samp, traj.shape[1], 6)
    _, local = torch.unique_consecutive(col["scene_index"], return_inverse=True)
    tot = x.sum() * 0.0
    kw = {k: col[k] for k in ("num_disks", "buffer_dist", "decay_rate", "moving_speed_th", "excluded_agents") if k in col}
    for si, wgt in enumerate(col["scene_weight"]):
        if float(wgt) == 0.0:
            continue
        mask = local == si
        sub = (col.get("agents") or {}).get(si)
        if sub is not None:
            idx = torch.nonzero(mask, as_tuple=True)[0][torch.as_tensor(sub)]
            mask = torch.zeros_like(mask)
            mask[idx] = True
        tot = tot + agent_collision_loss(x, col["extent"], col["world_from_agent"], col["curr_speed"], col["scene_index"], mask, **kw).mean() * float(wgt)
    return tot


def guidance_losses(traj: Tensor, target_speed: Optional[Tensor] = None, loss_scale: Optional[Tensor] = None, speed_limit=None,
                    acc_limit=None, target_pos=None) -> Tensor:
    """The unweighted per-agent values upstream's guidance losses return (what DiffuserGuidance.compute_guidance_loss stores in
    `guide_losses`, guidance_loss.py:2143-2172) on decoded trajectories [B,52,6] -> [B,4] = (TargetSpeedLoss :219-254,
    SpeedLimitLoss :1509-1538, AccLimitLoss :1444-1467, TargetPosAtTimeLoss :632-670 | TargetPosLoss :672-716); NaN where a
    term is off for the agent (upstream: agents outside the loss's mask, :2167-2169).  Argument conventions as `guidance_step`."""
    B = traj.shape[0]
    out = torch.full((B, 4), float("nan"), dtype=traj.dtype)
    if target_speed is not None:
        on = torch.ones(B, dtype=torch.bool) if loss_scale is None else loss_scale != 0
        dev = torch.nan_to_num((traj[..., 2] - target_speed).abs(), nan=0.0)
        out[on, 0] = dev.mean(dim=-1)[on]
    if speed_limit is not None:
        on = speed_limit[1] != 0
        out[on, 1] = (traj[..., 2].abs() - speed_limit[0]).clamp(min=0).mean(dim=-1)[on]
    if acc_limit is not None:
        on = acc_limit[1] != 0
        out[on, 2] = (traj[..., 4].abs() - acc_limit[0]).clamp(min=0).mean(dim=-1)[on]
    if target_pos is not None:
        p_, t_, s_ = target_pos
        for b in range(B):
            if float(s_[b]) == 0.0:
                continue
            tb = int(t_[b])
            if tb >= 0:
                out[b, 3] = (traj[b, min(tb, 51), :2] - p_[b]).norm()
            else:
                e = traj[b, min(-tb - 1, 51):, :2] - p_[b]
                out[b, 3] = (F.softmin(e.norm(dim=-1), dim=-1) * (e ** 2).sum(dim=-1)).mean()
    return out


SCENE_LEVEL_LOSSES = ("agent_collision", "social_group", "gptcollision", "gptkeepdistance")


def choose_action_from_guidance(guide_losses: Dict[str, Tensor], guide_config_names) -> Tensor:
    """guidance_loss.py:22-66 for the agent-centric layout (preds [M,N,T,2], B = 1): `guide_losses` maps
    '<name>_scene_%03d_%02d' -> [M,N] (NaN outside the loss's agents), stacked in dict order; `guide_config_names` is the
    per-scene list of loss names (`guide_configs[si][g].name`).  Restated as written, INCLUDING its quirk: every scene's
    result overwrites `act_idx` for the whole batch (the scene mask is commented out, :49-50,62-63), so the returned
    indices are the LAST scene's -- agents outside it see an all-NaN row, nansum 0, argmin 0."""
    accum = torch.stack([v for v in guide_losses.values()], dim=2)                 # [M, N, total number of losses]
    M, N = accum.shape[:2]
    act_idx = torch.zeros(M, dtype=torch.long)
    scount = 0
    for names in guide_config_names:
        ends = scount + len(names)
        scene_loss = torch.nansum(accum[..., scount:ends], dim=-1)                   # [M, N]
        scount = ends
        if any(nm in SCENE_LEVEL_LOSSES for nm in names):                            # one sample index for the whole scene (B = 1)
            act_idx = torch.argmin(scene_loss.reshape(1, M, N).sum(dim=1), dim=1).unsqueeze(-1).expand(1, M).reshape(M)
        else:
            act_idx = torch.argmin(scene_loss, dim=-1)
    return act_idx


def choose_action_from_gt(positions: Tensor, target_positions: Tensor, target_availabilities: Tensor) -> Tensor:
    """guidance_loss.py:67-99: the sample with the smallest average displacement from the ground-truth future over its valid
    steps; rows whose every step is invalid keep sample 0.  positions [M,N,T,2]."""
    M, N, T_ = positions.shape[:3]
    endT = min(T_, target_positions.shape[1])
    err = torch.norm(positions[:, :, :endT] - target_positions[:, :endT].unsqueeze(1), dim=-1)
    valid = target_availabilities[:, :endT].unsqueeze(1).expand(M, N, endT).bool()
    err = torch.where(valid, err, torch.full_like(err, float("nan")))
    ade = torch.nanmean(err, dim=-1)
    ok = torch.isnan(ade).sum(dim=-1) == 0
    act_idx = torch.zeros(M, dtype=torch.long)
    if bool(ok.any()):
        act_idx[ok] = torch.argmin(ade, dim=-1)[ok]
    return act_idx


def world_step(traj: Tensor, centroid: Tensor, yaw: Tensor, k: int):
    """EnvUnifiedSimulation._step, src/tbsim/envs/env_trajdata.py:452-468, for action index k: the planned state k (agent
    frame at planning time) placed in the world: xy' = p_k @ [[c, s], [-s, c]] + centroid, h' = yaw + yaw_k.  Returns
    (world [B,3] = (x, y, h), next curr_states [B,4] = (0, 0, v_k, 0): the agent-centric state the next planning call
    conditions on, batch_utils.py:46-65)."""
    c, s_ = torch.cos(yaw), torch.sin(yaw)
    px, py = traj[:, k, 0], traj[:, k, 1]
    wx = px * c - py * s_ + centroid[:, 0]
    wy = px * s_ + py * c + centroid[:, 1]
    world = torch.stack([wx, wy, yaw + traj[:, k, 3]], dim=1)
    cs = torch.zeros(traj.shape[0], 4, dtype=traj.dtype)
    cs[:, 2] = traj[:, k, 2]
    return world, cs


def closed_loop(w, wdec, sched, cond_fn, centroid: Tensor, yaw: Tensor, cs: Tensor, x_T: Tensor, noise: Tensor, n_sim_steps: int,
                n_step_action: int = 5) -> Tensor:
    """The loop of rollout_episodes (src/tbsim/utils/env_utils.py:255-304): observation -> plan (sample + decode, sample 0
    executed) -> the world takes `n_step_action` steps of the plan (env_trajdata.py:452-468) -> re-plan.  `cond_fn(step, world,
    curr_states) -> cond_feat` stands in for observation + ContextEncoder; the same x_T / noise feed every planning call.
    Returns the world poses after each sim step [n_sim_steps, B, 3]."""
    world = torch.cat([centroid, yaw[:, None]], dim=1)
    poses = []
    for step in range(n_sim_steps):
        cond = cond_fn(step, world, cs)
        out = sample(w, sched, x_T, noise, cond)
        traj = decode(wdec, out["pred_traj"], cond, cs, True)
        world, cs = world_step(traj, world[:, :2], world[:, 2], n_step_action - 1)
        poses.append(world)
    return torch.stack(poses)


def get_action(w, wdec, sched, cond: Tensor, cs: Tensor, x_T: Tensor, noise: Tensor, num_samp: int = 1, guidance: Optional[dict] = None,
               guide_config_names=None, filter_only: bool = False, stationary_th: Optional[float] = None):
    """The get_action contract of upstream's DiffuserTrafficModel (src/tbsim/algos/algos.py:2024-2099) over this path:
    repeat every agent num_samp times (dm_model.py:116), sample (guided on the steps t > 0 unless `filter_only`,
    algos.py:1815), decode, evaluate the guidance losses on the final output (diffuser.py:924-926) and execute the sample
    `choose_action_from_guidance` picks (sample 0 without guidance, algos.py:2053-2054); stationary agents are zeroed
    (algos.py:2076-2083).  `guidance` = dict(target_speed [B,52], loss_scale [B] | None) (per-agent tensors are repeated
    like the batch).  Returns (positions [B,52,2], yaws [B,52,1], act_idx [B], trajectories [B,N,52,6], guide losses [B,N,4] | None)."""
    B, N = cond.shape[0], num_samp
    rep = lambda v: v if v is None else v.repeat_interleave(N, dim=0)
    cond_r, cs_r = rep(cond), rep(cs)
    if guidance is not None and not filter_only:
        out = sample_guided(w, wdec, sched, x_T, noise, cond_r, cs_r, rep(guidance["target_speed"]), rep(guidance.get("loss_scale")),
                            lr=guidance.get("lr", 0.3), optimizer=guidance.get("optimizer", "adam"))
    else:
        out = sample(w, sched, x_T, noise, cond_r)
    traj = decode(wdec, out["pred_traj"], cond_r, cs_r, True).reshape(B, N, 52, 6)
    act_idx = torch.zeros(B, dtype=torch.long)
    gl = None
    if guidance is not None:
        gl = guidance_losses(traj.reshape(B * N, 52, 6), rep(guidance["target_speed"]), rep(guidance.get("loss_scale"))).reshape(B, N, 4)
        names = guide_config_names if guide_config_names is not None else [["target_speed"]]
        act_idx = choose_action_from_guidance({"target_speed_scene_000_00": gl[..., 0]}, names)
    pos, yaw_ = traj[..., :2].clone(), traj[..., 3:4].clone()
    if stationary_th is not None:
        still = cs[:, 2].abs() < stationary_th
        pos[still] = 0
        yaw_[still] = 0
    ar = torch.arange(B)
    return pos[ar, act_idx], yaw_[ar, act_idx], act_idx, traj, gl


# --------------------------------------------------------------------------- #
# f-3  PPO reward (models/rl/criticmodel.py:7-64,88-145)
# --------------------------------------------------------------------------- #
def transform_points(traj_xy: Tensor, raster_from_agent: Tensor) -> Tensor:
    """criticmodel.py:101-112 -- [B,T,2] points times the transposed 3x3: p' = R[:2,:2] p + R[:2,2]."""
    Tm = raster_from_agent.transpose(1, 2)
    return torch.bmm(traj_xy, Tm[:, :2, :2]) + Tm[:, -1:, :2]


def offroad_flags(traj: Tensor, raster_from_agent: Tensor, drivable_map: Tensor) -> Tensor:
    """criticmodel.py:13-23 / :121-127 -- [B,T] True where the rounded, clamped raster pixel is drivable."""
    B, T = traj.shape[:2]
    ti = transform_points(traj[..., :2], raster_from_agent).round().long()
    cols = ti[..., 0].clamp(0, drivable_map.shape[-1] - 1)
    rows = ti[..., 1].clamp(0, drivable_map.shape[-2] - 1)
    bi = torch.arange(B).view(B, 1).expand(B, T)
    return drivable_map[bi, rows, cols] != 0


def collision_reward(traj_xy: Tensor, other_pos: Tensor, other_avail: Tensor, thresh: float = 0.8) -> Tensor:
    """criticmodel.py:42-64 (3-D branch) -- minus the number of (other agent, timestep) pairs closer than `thresh`."""
    To = other_pos.shape[2]
    d = (traj_xy[:, None, :To] - other_pos).norm(dim=-1)
    return -((d < thresh) & (other_avail != 0)).float().sum(dim=(1, 2))


def compute_reward(traj: Tensor, traj_scaled: Tensor, raster_from_agent: Tensor, drivable_map: Tensor, other_pos: Tensor,
                   other_avail: Tensor, thresh: float = 0.8):
    """criticmodel.py:7-40 per agent (num_samp = 1; the reference function itself cannot run: it unpacks 4-D and calls the
    3-D helpers) -> (reward, offroad, collision)."""
    off = -(~offroad_flags(traj, raster_from_agent, drivable_map)).float().sum(dim=-1)
    col = collision_reward(traj[..., :2], other_pos, other_avail, thresh)
    acc = traj_scaled[..., 4]
    jerk = ((acc[:, 1:] - acc[:, :-1]) / 0.1).abs().mean(dim=-1)
    return off + col - jerk * 0.1, off, col


def vae_loss(x6_scaled: Tensor, act_out: Tensor, mu: Tensor, logvar: Tensor, beta: float):
    """models/vae/vae_model.py:89-99 -> (loss, recon, kld)."""
    recon = F.mse_loss(x6_scaled[..., -2:], act_out, reduction="mean")
    B, T, _ = mu.shape
    kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / (B * T)
    return recon + beta * kld, recon, kld


def to_torch(d: dict, dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in d.items()}
