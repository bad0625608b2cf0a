"""The policy surface the reference's rollout loop expects but CLD never implemented
(`rollout.py:95-100` calls `policy.get_action(obs)`; `DMLightningModule` has none -- SURVEY section 0).
The contract is upstream's `DiffuserTrafficModel.get_action` (`src/tbsim/algos/algos.py:2024-2099`):
returns `(Action(positions [B,T,2], yaws [B,T,1]), {"action_samples": {...}})`, sample 0 is the action,
stationary agents are zeroed; `Action` is `src/tbsim/policies/common.py:10-66`.

`obs_dict` is either the reference's observation batch (`image` [B,34,224,224], `history_positions`, `history_yaws`,
`curr_speed`: the `ContextEncoder` of `context_utils.py` turns it into `cond_feat` / `curr_states` on the device), or
already carries `cond_feat` [B,256] and `curr_states` [B,4]; a different `context_encoder` callable may be passed in.
Parity: unpinned in the reference (no CLD implementation exists); tests check the composition against the
oracle's sample/decode chain and the world update against a NumPy restatement of `env_trajdata.py:452-468`.
"""
from __future__ import annotations

from typing import Callable, Mapping, Optional

import numpy as np
import torch

from .dm_model import DmModel
from .vae_model import VaeModel


class Action:
    """Container for sequences of 2-D positions and yaws (policies/common.py:10-66)."""

    def __init__(self, positions, yaws):
        assert positions.shape[:-1] == yaws.shape[:-1] and positions.shape[-1] == 2 and yaws.shape[-1] == 1
        self.positions, self.yaws = positions, yaws

    @property
    def trajectories(self):
        cat = np.concatenate if isinstance(self.positions, np.ndarray) else torch.cat
        return cat([self.positions, self.yaws], -1)

    def to_dict(self):
        return dict(positions=self.positions, yaws=self.yaws)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    def to_numpy(self):
        f = lambda x: x if isinstance(x, np.ndarray) else x.detach().cpu().numpy()
        return Action(f(self.positions), f(self.yaws))


class CldPolicy:
    def __init__(self, dm: DmModel, vae: VaeModel, context_encoder: Optional[Callable] = None,
                 disable_control_on_stationary: bool = False, moving_speed_th: float = 0.5):
        self.dm, self.vae = dm, vae
        self.context_encoder = context_encoder
        self.disable_control_on_stationary = disable_control_on_stationary   # config.yaml:99
        self.moving_speed_th = moving_speed_th                               # config.yaml:101

    def eval(self):
        return self

    @torch.no_grad()
    def get_action(self, obs_dict: Mapping, num_action_samples: int = 1, class_free_guide_w: float = 0.0,
                   step_index: int = 0, noise: Optional[Mapping] = None, guidance: Optional[Mapping] = None, **kwargs):
        if "cond_feat" in obs_dict:
            aux = obs_dict
        else:                                                       # obs -> aux_info (vae_model.py:84-88 pre_vae)
            aux = (self.context_encoder or self.vae.context_encoder)(obs_dict)
        cond, cs = aux["cond_feat"], aux["curr_states"]
        B, N = cond.shape[0], int(num_action_samples)
        out = self.dm({"history_positions": cond}, {k: aux[k] for k in ("cond_feat", "curr_states", "non_cond_feat") if k in aux},
                      {"num_samp": N}, noise=noise, class_free_guide_w=class_free_guide_w, guidance=guidance)
        a = out["aux_info"]
        traj = self.vae.engine.decode(out["pred_traj"], a["cond_feat"], a["curr_states"], descaled_output=True)
        traj = traj.reshape(B, N, 52, 6)
        pos, yaw = traj[..., :2].clone(), traj[..., 3:4].clone()
        if self.disable_control_on_stationary:                      # algos.py:2076-2083
            still = (cs[:, 2].abs() < self.moving_speed_th).to(pos.device)
            pos[still] = 0
            yaw[still] = 0
        act_idx = 0                                                 # "arbitrarily use the first sample", algos.py:2053-2054
        info = dict(action_samples=Action(pos, yaw).to_dict(), trajectories=traj)
        return Action(pos[:, act_idx], yaw[:, act_idx]), info


def closed_loop_rollout(policy: CldPolicy, cond_fn: Callable, centroid, yaw, curr_states, n_sim_steps: int,
                        n_step_action: int = 5, gather: Optional[Callable] = None, timers=None, **get_action_kwargs):
    """The loop of `rollout_episodes` (`src/tbsim/utils/env_utils.py:255-304`) kept on the device:
    obs -> get_action -> take `n_step_action` steps of the plan -> new world pose -> re-plan.
    `cond_fn(step, world [B,3], curr_states [B,4]) -> cond_feat [B,256]` stands in for the observation +
    ContextEncoder stage (it may return the raw observation batch instead: a dict with `image`, `history_*`,
    `curr_speed`, which get_action runs through the ContextEncoder); `gather(traj)` (e.g. `parallel.gather_trajectories`)
    runs once per sim step so every rank sees all agents' plans; `timers` (`cld_amd.timer.Timers`) collects the per-phase
    times under the reference's keys "obs" / "network" / "env_step" / "step" (env_utils.py:268-298).
    Returns the world poses after each sim step [n_sim_steps, B, 3]."""
    from contextlib import nullcontext
    tm = (lambda k: timers.timed(k)) if timers is not None else (lambda k: nullcontext())
    eng = policy.vae.engine
    world = torch.cat([torch.as_tensor(centroid), torch.as_tensor(yaw)[:, None]], dim=1).to(eng.device, torch.float32)
    cs = torch.as_tensor(curr_states).to(eng.device, torch.float32)
    poses = []
    for step in range(n_sim_steps):
        with tm("step"):
            with tm("obs"):
                o = cond_fn(step, world, cs)
                obs = o if isinstance(o, Mapping) else {"cond_feat": o, "curr_states": cs}
            with tm("network"):
                _, info = policy.get_action(obs, step_index=step, **get_action_kwargs)
                traj = info["trajectories"][:, 0].contiguous()
            with tm("env_step"):
                if gather is not None:
                    gather(traj)
                world, cs = eng.world_step(traj, world[:, :2].contiguous(), world[:, 2].contiguous(), n_step_action - 1)
        poses.append(world)
    return torch.stack(poses)
