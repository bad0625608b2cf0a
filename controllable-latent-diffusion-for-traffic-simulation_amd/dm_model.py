"""Host-side mirror of the reference `DmModel` call surface over the HIP engine.

Same method names, argument meaning and returned dict keys as
`/root/reference/models/dm/dm_model.py:15-174` (forward / sample_traj /
x_Tminus1 / x_tminus1_mean_var / log_prob, and `.model(x, aux_info, t)` for
`TemporalMapUnet.forward`, `src/tbsim/models/temporal.py:122-180`), so the
callers in `src/trainers/guide_dm_trainer.py:87,164,189,207` work unchanged.
Differences, all additive: `noise=` lets the caller supply the Gaussian draws
(the reference draws them with `torch.randn`, which cannot be reproduced across
devices); `compute_losses` / `q_sample` exist in their forward-only form (the validation loss of
`src/trainers/dm_trainer.py:84-90`; training itself is out of scope).
"""
from __future__ import annotations

from typing import Mapping, Optional

import torch

from .engine import Engine
from ._lib import CldError


def cfg_get(cfg, path: str, default=None):
    """Read 'a.b' from a dict- or attribute-style config (reference configs are both)."""
    cur = cfg
    for p in path.split("."):
        if cur is None:
            return default
        if isinstance(cur, Mapping):
            cur = cur.get(p, None)
        else:
            cur = getattr(cur, p, None)
    return default if cur is None else cur


def repeat_by_expand_at(x, repeats: int, dim: int = 0):
    """tbsim.utils.tensor_utils.repeat_by_expand_at (tensor_utils.py:668-681) for tensors / dicts:
    each entry is repeated `repeats` times consecutively along `dim`."""
    if isinstance(x, Mapping):
        return {k: repeat_by_expand_at(v, repeats, dim) for k, v in x.items()}
    if isinstance(x, torch.Tensor):
        return x if repeats == 1 else x.repeat_interleave(repeats, dim=dim)
    return x


def repeat_guidance(guidance: Mapping, num_samp: int, curr_states=None) -> dict:
    """Per-agent tensors of a `guidance=` dict follow the num_samp repeat of the batch (dm_model.py:116); `curr_states`
    (already repeated) is filled in when the dict has none."""
    g = dict(guidance)
    if curr_states is not None:
        g.setdefault("curr_states", curr_states)

    def rep(v):
        return repeat_by_expand_at(v, num_samp, 0) if isinstance(v, torch.Tensor) and v.dim() >= 1 else v
    for k in ("target_speed", "loss_scale"):
        if g.get(k) is not None:
            g[k] = rep(torch.as_tensor(g[k]))
    for k in ("speed_limit", "acc_limit", "target_pos"):
        if g.get(k) is not None:
            g[k] = tuple(rep(v) for v in g[k])
    for k in ("agent_collision", "map_collision"):  # per-AGENT tensors stay as they are: sample n of every agent lives in scene copy n
        if g.get(k) is not None:
            g[k] = dict(g[k], num_samp=num_samp)
    return g


class _UnetSurface:
    """`dm.model(x, aux_info, t)` -- TemporalMapUnet.forward (temporal.py:122-180)."""

    def __init__(self, engine: Engine):
        self.engine = engine

    def __call__(self, x, aux_info, time):
        cond = aux_info["cond_feat"]
        four_d = x.dim() == 4                      # [BN, M, T, D] with cond [BN, M, C]  (temporal.py:127-135)
        if four_d:
            BN, M, T, _ = x.shape
            x = x.reshape(BN * M, T, -1)
            cond = cond.reshape(BN * M, -1)
            time = time.repeat_interleave(M, dim=0)
        if not torch.is_tensor(time) or time.device.type == "cpu":
            # a Python int or host values: whether the batch shares one timestep is decided here, without a device round trip
            tt = torch.as_tensor(time).reshape(-1)
            if tt.numel() == 1 or bool((tt == tt[0]).all()):
                eps = self.engine.unet_forward(x, cond, int(tt[0]))
            else:
                eps = self.engine.unet_forward_rows(x, cond, tt)
        else:                                      # device tensor (0-dim or [B]): per-row path (the time bias is folded per agent,
            time = time.reshape(-1)                    # cld_unet_forward_t); no .to("cpu") / unique, so no device sync per call
            if time.numel() == 1:
                time = time.expand(x.shape[0])
            eps = self.engine.unet_forward_rows(x, cond, time)
        return eps.reshape(BN, M, T, -1) if four_d else eps


class DmModel:
    def __init__(self, algo_config=None, modality_shapes=None, n_timesteps: int = 100, device="cuda:0",
                 engine: Optional[Engine] = None):
        self.n_timesteps = int(n_timesteps)
        self.horizon = cfg_get(algo_config, "horizon", 52)
        self.dt = cfg_get(algo_config, "step_time", 0.1)
        if self.horizon != 52 or cfg_get(algo_config, "vae.latent_size", 4) != 4 or \
                cfg_get(algo_config, "cond_feat_dim", 256) != 256:
            raise CldError("only the reference architecture (horizon 52, latent 4, cond 256) is built")
        self.engine = engine or Engine(n_timesteps=n_timesteps, device=device,
                                       dynamics=cfg_get(algo_config, "dynamics"),
                                       norm_info=cfg_get(algo_config, "nusc_norm_info.diffuser"),
                                       step_time=self.dt)
        self.device = self.engine.device
        self.model = _UnetSurface(self.engine)
        for name in ("x_t_cof", "noise_cof", "posterior_log_variance_clipped"):
            setattr(self, name, torch.from_numpy(getattr(self.engine, name)).to(self.device))

    @property
    def stride(self) -> int:            # dm_model.py:25 -- a plain attribute in the reference; here it also reaches the engine
        return self.engine.stride

    @stride.setter
    def stride(self, value: int):
        self.engine.set_stride(value)

    # nn.Module-compatible conveniences used by the callers
    def load_state_dict(self, sd, strict=True):
        self.engine.load_state_dict(sd, strict=strict)
        self.engine.finalize()
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, data_batch, aux_info, algo_config, **kw):
        return self.forward(data_batch, aux_info, algo_config, **kw)

    # ---- dm_model.py:98-142 ------------------------------------------------------------
    @torch.no_grad()
    def forward(self, data_batch, aux_info, algo_config, noise: Optional[Mapping] = None, seed: int = 0,
                class_free_guide_w: float = 0.0, guidance: Optional[Mapping] = None, guidance_fn=None, **guidance_opt):
        """`guidance_fn(traj [BN,52,6]) -> scalar` is a caller-defined torch loss on the decoded trajectories (e.g. one of
        upstream's guidance losses); it takes the place of the built-in losses (`guidance=`) through `Engine.sample_with_loss`
        (`guidance_opt`: lr / optimizer / perturb_th).  Needs `noise=` and aux_info['curr_states']."""
        if guidance_fn is not None:
            num_samp = int(cfg_get(algo_config, "num_samp", 1))
            aux = repeat_by_expand_at(aux_info, repeats=num_samp, dim=0)
            if noise is None:
                BN = aux["cond_feat"].shape[0]
                noise = {"x_T": torch.randn(BN, 52, 4, device=self.device), "noise": torch.randn(self.engine.loop_steps, BN, 52, 4, device=self.device)}
            x0, x1 = self.engine.sample_with_loss(noise["x_T"], aux["cond_feat"], aux["curr_states"], noise["noise"], guidance_fn, **guidance_opt)
            return {"pred_traj": x0, "x1": x1, "log_prob_final": None, "aux_info": aux}
        return self.sample_traj(data_batch, algo_config, aux_info, noise=noise, seed=seed,
                                class_free_guide_w=class_free_guide_w, guidance=guidance)

    def sample_traj(self, data_batch, algo_config, aux_info, noise: Optional[Mapping] = None, seed: int = 0,
                    class_free_guide_w: float = 0.0, guidance: Optional[Mapping] = None):
        """`class_free_guide_w` != 0 needs aux_info['non_cond_feat'] and follows the upstream CFG definition
        (src/tbsim/models/diffuser.py:766-789; kwarg name as injected by policies/wrappers.py:143-167).
        `guidance` = dict(target_speed [B,52], loss_scale [B] | None, lr, perturb_th, optimizer): sampling-time guidance
        of every step's posterior mean (upstream diffuser.py:844-929; Engine._guidance); curr_states come from aux_info."""
        batch_size = data_batch["history_positions"].size()[0]
        num_samp = int(cfg_get(algo_config, "num_samp", 1))
        BN = batch_size * num_samp
        aux_info = repeat_by_expand_at(aux_info, repeats=num_samp, dim=0)
        if noise is not None:
            x_T, z = noise["x_T"], noise.get("noise")
        else:   # same draw count as the reference (one for x_T, one per step), from torch's device generator
            x_T = torch.randn(BN, 52, 4, device=self.device)
            z = torch.randn(self.engine.loop_steps, BN, 52, 4, device=self.device)
        x_T = torch.as_tensor(x_T).reshape(BN, 52, 4)
        if guidance is not None:
            guidance = repeat_guidance(guidance, num_samp, aux_info["curr_states"])
        if class_free_guide_w and aux_info.get("non_cond_feat") is None:
            # upstream builds non_cond_feat itself (diffuser.py:390-411,459-471); here the ContextEncoder mirror does
            # (`include_class_free_cond=True`, Engine.non_cond_feat).  Running unguided instead would be a silent wrong answer.
            raise CldError("class_free_guide_w != 0 needs aux_info['non_cond_feat'] [B,256] (ContextEncoder(..., include_class_free_cond=True) "
                           "or Engine.non_cond_feat(curr_states) produce it)")
        x0, x1, logp = self.engine.sample(x_T, aux_info["cond_feat"], noise=z, seed=seed,
                                          non_cond=aux_info.get("non_cond_feat") if class_free_guide_w else None,
                                          guidance_w=class_free_guide_w, guidance=guidance)
        return {"pred_traj": x0, "x1": x1, "log_prob_final": logp, "aux_info": aux_info}

    # ---- dm_model.py:144-163 -----------------------------------------------------------
    def _t_scalar(self, t) -> int:
        t = torch.as_tensor(t).reshape(-1)
        v = int(t[0])
        if t.numel() > 1 and not bool((t == v).all()):
            raise CldError("the sampler surface takes one timestep for the whole batch (dm_model.py:122)")
        return v

    def x_Tminus1(self, x, t, aux_info, noise=None):
        i = self._t_scalar(t)
        if noise is None:
            noise = torch.randn(x.shape, device=self.device)
        xn, mean, sigma = self.engine.ddpm_step(x, aux_info["cond_feat"], i, noise)
        return xn, mean, torch.full((x.shape[0], 1, 1), sigma, device=self.device)

    def x_tminus1_mean_var(self, xt, noise, t):
        i = self._t_scalar(t)
        mean = float(self.engine.x_t_cof[i]) * xt - float(self.engine.noise_cof[i]) * noise
        logvar = torch.full((xt.shape[0], 1, 1), float(self.engine.posterior_log_variance_clipped[i]), device=xt.device)
        return mean, logvar

    # ---- dm_model.py:82-96 (forward only: the validation loss of dm_trainer.py:84-90) ---------
    def q_sample(self, x_0, t, noise):
        return self.engine.q_sample(x_0, noise, t)

    def compute_losses(self, aux_info, z0, t=None, noise=None):
        """MSE between the drawn noise and the U-Net's prediction on q_sample(z0, t, noise); `t` / `noise` default to the
        reference's draws (torch.randint / randn_like on the device).  No gradients: training is out of scope."""
        B = len(z0)
        if t is None:
            t = torch.randint(0, self.n_timesteps, (B,), device=self.device)
        if noise is None:
            noise = torch.randn(B, 52, 4, device=self.device)
        return self.engine.denoise_loss(z0, noise, aux_info["cond_feat"], t).mean()

    # ---- dm_model.py:165-174 -----------------------------------------------------------
    def log_prob(self, x_t, x_t_minus_1, aux_info, t):
        return self.engine.log_prob(x_t, x_t_minus_1, aux_info["cond_feat"], self._t_scalar(t))
