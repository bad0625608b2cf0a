"""Host-side mirror of the VAE-decoder half of the reference call surface:
`LSTMVAE.lstm_dec(z, context)` (models/vae/lstm_vae.py:44-52) and
`VaeModel.convert_action_to_state_and_action / scale_traj / descale_traj`
(models/vae/vae_model.py:100-173), as used at
src/trainers/guide_dm_trainer.py:97-98,195-196,210-211.
`traj2z` (the encoder) and `ContextEncoder` / `pre_vae` (f-1, context_utils.py) are built too.
"""
from __future__ import annotations

from typing import Optional

import torch

from .context_utils import ContextEncoder
from .dm_model import cfg_get
from .engine import Engine


class LSTMVAE:
    def __init__(self, engine: Engine):
        self.engine = engine

    def lstm_dec(self, z, context):
        return self.engine.lstm_decode(z, context)

    def forward(self, x, context, noise=None):
        """lstm_vae.py:82-85 -> (act_output [B,52,2], mean, logvar): encode, reparametrise, decode."""
        z, mean, logvar = self.traj2z(x, context, noise)
        return self.lstm_dec(z, context), mean, logvar

    __call__ = forward

    def traj2z(self, x, context, noise=None):
        """lstm_vae.py:87-99 -> (z, mean, logvar).  `noise` replaces the reference's randn_like draw (:97);
        drawn from torch's device generator when omitted."""
        if noise is None:
            noise = torch.randn(x.shape[0], 52, 4, device=self.engine.device)
        return self.engine.traj2z(x, context, noise)


class VaeModel:
    def __init__(self, algo_config=None, train_config=None, modality_shapes=None, device="cuda:0",
                 engine: Optional[Engine] = None):
        self._owns_engine = engine is None       # standalone VaeModel: a handle without U-Net weights (cld_finalize allows it)
        self.engine = engine or Engine(device=device, dynamics=cfg_get(algo_config, "dynamics"),
                                       norm_info=cfg_get(algo_config, "nusc_norm_info.diffuser"))
        self.lstmvae = LSTMVAE(self.engine)
        self.context_encoder = ContextEncoder(self.engine)
        self.default_chosen_inds = [0, 1, 2, 3, 4, 5]
        self.dt = 0.1
        c = self.engine.cfg
        self.add_coeffs = torch.tensor(list(c.norm_mean), dtype=torch.float32)
        self.div_coeffs = torch.tensor(list(c.norm_std), dtype=torch.float32)

    def load_state_dict(self, sd, strict=True):
        """Weights under the reference's `lstmvae.*` / `context_encoder.*` names.  A VaeModel that owns its engine finalizes it
        here (decoder / encoder / ContextEncoder calls work, U-Net calls report the missing weights); with a shared engine
        (`engine=`) the DmModel's load_state_dict finalizes, so load the VAE weights first."""
        self.engine.load_state_dict(sd, strict=strict)
        if self._owns_engine:
            self.engine.finalize()
        return self

    def pre_vae(self, batch):
        """vae_model.py:84-88 -> (aux_info, state_and_action_scaled, state_and_action)."""
        aux_info = self.context_encoder(batch)
        sa = self.get_state_and_action_from_data_batch(batch, scaled=False)
        return aux_info, self.get_state_and_action_from_data_batch(batch, scaled=True), sa

    def forward(self, batch, beta=None, noise=None):
        """VaeModel.forward (vae_model.py:64-82), forward values only (training is out of scope): pre_vae -> lstmvae ->
        convert_action_to_state_and_action -> descale, and the loss terms of compute_vae_loss; the reference's dict keys."""
        aux_info, sa_scaled, _ = self.pre_vae(batch)
        recon_act, mu, logvar = self.lstmvae(sa_scaled, aux_info["cond_feat"], noise)
        recon = self.convert_action_to_state_and_action(recon_act, aux_info["curr_states"], descaled_output=True)
        lt = self.engine.vae_loss(sa_scaled, recon_act, mu, logvar, 0.0 if beta is None else float(beta))
        return {"loss": lt[0], "recon": lt[1], "kld": lt[2], "hist": batch["history_positions"], "input": batch.get("target_positions"), "output": recon[..., :2],
                "raster_from_agent": batch.get("raster_from_agent"), "image": batch["image"], "mu": mu, "logvar": logvar,
                "recon_act": recon_act}

    def compute_vae_loss(self, input, output, mu, logvar, beta):
        """vae_model.py:89-99 -> (loss, recon, kld), forward only."""
        lt = self.engine.vae_loss(input, output, mu, logvar, float(beta))
        return lt[0], lt[1], lt[2]

    def convert_action_to_state_and_action(self, x_out, curr_states, scaled_input=True, descaled_output=False):
        four_d = x_out.dim() == 4          # vae_model.py:108-111
        if four_d:
            B, N, T, _ = x_out.shape
            x_out = x_out.reshape(B * N, T, -1)
        out = self.engine.action_to_state(x_out, curr_states, scaled_input, descaled_output)
        return out.reshape(B, N, T, -1) if four_d else out

    def get_state_and_action_from_data_batch(self, batch, scaled=False):
        """models/context_utils.py:64-70 (+ scale_traj when `scaled`): future (x, y, yaw) + curr_speed ->
        [B,52,6] = (x, y, v, yaw, acc, yaw-rate)."""
        return self.engine.state_to_state_and_action(batch["target_positions"][:, :52], batch["target_yaws"][:, :52],
                                                     batch["curr_speed"], scaled_output=scaled)

    def scale_traj(self, traj, chosen_inds=()):
        inds = list(chosen_inds) or self.default_chosen_inds      # (x - mean) / std, vae_model.py:152
        return (traj - self.add_coeffs[inds].to(traj.device)) / self.div_coeffs[inds].to(traj.device)

    def descale_traj(self, traj, chosen_inds=()):
        inds = list(chosen_inds) or self.default_chosen_inds      # x * std + mean, vae_model.py:170
        return traj * self.div_coeffs[inds].to(traj.device) + self.add_coeffs[inds].to(traj.device)


class DecodeFn(torch.autograd.Function):
    """`traj = DecodeFn.apply(z, cond, curr_states, engine)`: the decode chain (lstm_dec -> descale -> unicycle roll-out,
    descaled [B,52,6]) as a differentiable torch op whose backward is the HIP vector-Jacobian product `Engine.decode_vjp`.
    Upstream's guidance losses (`src/tbsim/utils/guidance_loss.py`) are torch code on exactly this trajectory, so they can be
    evaluated with autograd on the decoded output and pulled back to the latent without any CPU or torch fallback of the
    path itself.  Gradients flow to `z` only."""

    @staticmethod
    def forward(ctx, z, cond, curr_states, engine):
        ctx.engine = engine
        ctx.save_for_backward(z.detach(), cond.detach(), curr_states.detach())
        return engine.decode(z.detach(), cond, curr_states, descaled_output=True)

    @staticmethod
    def backward(ctx, grad_traj):
        z, cond, cs = ctx.saved_tensors
        return ctx.engine.decode_vjp(z, cond, cs, grad_traj.contiguous()), None, None, None
