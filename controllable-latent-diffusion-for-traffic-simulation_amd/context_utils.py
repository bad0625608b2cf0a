"""Host-side mirror of `models/context_utils.py`: `ContextEncoder.forward(data_batch)` ->
aux_info {'cond_feat', 'curr_states', 'image'} and `get_state_and_action_from_data_batch`.

All arithmetic (ResNet-18 on the raster, both MLPs) runs in libcld_hip behind `cld_context_encode`; the only
thing done here is the slicing that assembles `curr_states` from the batch dict.
"""
from __future__ import annotations

import torch

from .engine import Engine


def get_current_states(batch: dict) -> torch.Tensor:
    """batch_utils.get_current_states for the unicycle model (src/tbsim/utils/batch_utils.py:46-65):
    [x, y, vel, yaw] = (history_positions[..., -1, :], curr_speed, history_yaws[..., -1, 0])."""
    spd = batch["curr_speed"]
    cs = torch.zeros(*spd.shape, 4, dtype=torch.float32, device=spd.device)
    cs[..., :2] = batch["history_positions"][..., -1, :]
    cs[..., 2] = spd
    cs[..., 3] = batch["history_yaws"][..., -1, 0]
    return cs


class ContextEncoder:
    """Same call surface as the reference module (context_utils.py:8-61); weights arrive through
    `Engine.load_state_dict` under their `context_encoder.*` state_dict names."""

    def __init__(self, engine: Engine):
        self.engine = engine

    def forward(self, data_batch: dict, include_class_free_cond: bool = False, cond_fill_value: float = -1.0) -> dict:
        """`include_class_free_cond` adds aux_info['non_cond_feat'] as upstream builds it for classifier-free guidance
        (src/tbsim/models/diffuser.py:390-411,459-471): same state features, map features of a raster filled with -1."""
        curr_states = get_current_states(data_batch).to(self.engine.device)
        image = data_batch["image"]
        cond_feat = self.engine.context_encode(image, curr_states)
        aux = {"cond_feat": cond_feat, "curr_states": curr_states, "image": image}
        if include_class_free_cond:
            aux["non_cond_feat"] = self.engine.non_cond_feat(curr_states, cond_fill_value)
        return aux

    __call__ = forward


def get_state_and_action_from_data_batch(engine: Engine, batch: dict, chosen_inds=()):
    """context_utils.py:64-70: future (x, y, yaw) + curr_speed -> [B,52,6] = (x, y, v, yaw, acc, yaw-rate)."""
    inds = list(chosen_inds) or [0, 1, 2, 3, 4, 5]
    out = engine.state_to_state_and_action(batch["target_positions"][:, :52], batch["target_yaws"][:, :52], batch["curr_speed"])
    return out[..., inds]
