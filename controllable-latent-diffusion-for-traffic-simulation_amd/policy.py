"""The policy surface the reference's rollout loop expects but CLD never implemented
(`rollout.py:95-100` calls `policy.get_action(obs)`; `DMLightningModule` has none -- SURVEY section 0).
The contract is upstream's `DiffuserTrafficModel.get_action` (`src/tbsim/algos/algos.py:2024-2099`):
returns `(Action(positions [B,T,2], yaws [B,T,1]), {"action_samples": {...}})`; the executed sample is sample 0, or -- with
guidance active -- the one `choose_action_from_guidance` picks from the per-sample guidance losses (`guide_as_filter_only`,
`guide_with_gt` honoured); stationary agents are zeroed; `Action` is `src/tbsim/policies/common.py:10-66`.

`obs_dict` is either the reference's observation batch (`image` [B,34,224,224], `history_positions`, `history_yaws`,
`curr_speed`: the `ContextEncoder` of `context_utils.py` turns it into `cond_feat` / `curr_states` on the device), or
already carries `cond_feat` [B,256] and `curr_states` [B,4]; a different `context_encoder` callable may be passed in.
Parity: CLD itself never implemented `get_action`; the sample selection by guidance loss is pinned by a golden recorded from the
reference's own `choose_action_from_guidance` (`tests/golden/select.npz`).  `choose_action_from_gt` is a restatement of the
evident intent and PARITY UNPINNED: the reference function reads an undefined name `T` and raises `NameError` as written
(`guidance_loss.py:67-99`), so nothing could be recorded from it.  The composition and the world update are tested against
`oracle/cld_oracle.py` (`get_action`, `world_step`: `env_trajdata.py:452-468`).
"""
from __future__ import annotations

from typing import Callable, Mapping, Optional

import numpy as np
import torch

from .dm_model import DmModel, repeat_guidance
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


SCENE_LEVEL_LOSSES = ("agent_collision", "social_group", "gptcollision", "gptkeepdistance")
LOSS_COLUMN = {"target_speed": 0, "speed_limit": 1, "acc_limit": 2, "target_pos_at_time": 3, "target_pos": 3}   # cld_guidance_losses


def choose_action_from_guidance(guide_losses: Mapping, guide_config_names, per_scene: bool = False, scene_of_agent=None):
    """`choose_action_from_guidance` (src/tbsim/utils/guidance_loss.py:22-66), agent-centric layout: `guide_losses` maps
    '<name>_scene_%03d_%02d' -> [B,N] (NaN outside the loss's agents) in dict order, `guide_config_names` lists the loss names of
    every scene.  Per scene the losses are nansum-med and every agent takes the sample with the smallest sum.  As written
    upstream the result of EVERY scene overwrites the indices of the whole batch (the scene mask is commented out, :49-50,62-63),
    so the LAST scene's choice wins and agents outside it get sample 0; that is the default here too (parity with upstream).
    `per_scene=True` with `scene_of_agent` [B] applies each scene's choice to its own agents only -- the evident intent."""
    accum = torch.stack([v for v in guide_losses.values()], dim=2)
    B, N = accum.shape[:2]
    act_idx = torch.zeros(B, dtype=torch.long, device=accum.device)
    scount = 0
    for si, names in enumerate(guide_config_names):
        ends = scount + len(names)
        scene_loss = torch.nansum(accum[..., scount:ends], dim=-1)
        scount = ends
        if any(nm in SCENE_LEVEL_LOSSES for nm in names):
            idx = torch.argmin(scene_loss.sum(dim=0)).expand(B)
        else:
            idx = torch.argmin(scene_loss, dim=-1)
        if per_scene:
            mask = torch.as_tensor(scene_of_agent).to(accum.device) == si
            act_idx = torch.where(mask, idx, act_idx)
        else:
            act_idx = idx
    return act_idx


def choose_action_from_gt(positions, target_positions, target_availabilities):
    """`choose_action_from_gt` (guidance_loss.py:67-99): per agent the sample with the smallest average displacement from the
    ground-truth future over its valid steps; agents with a sample whose steps are all invalid keep sample 0."""
    B, N, T_ = positions.shape[:3]
    endT = min(T_, target_positions.shape[1])
    gt = torch.as_tensor(target_positions).to(positions.device, positions.dtype)[:, :endT].unsqueeze(1)
    valid = torch.as_tensor(target_availabilities).to(positions.device)[:, :endT].unsqueeze(1).expand(B, N, endT).bool()
    err = torch.norm(positions[:, :, :endT] - gt, dim=-1)
    err = torch.where(valid, err, torch.full_like(err, float("nan")))
    ade = torch.nanmean(err, dim=-1)
    ok = torch.isnan(ade).sum(dim=-1) == 0
    act_idx = torch.zeros(B, dtype=torch.long, device=positions.device)
    if bool(ok.any()):
        act_idx[ok] = torch.argmin(ade, dim=-1)[ok]
    return act_idx


class CldPolicy:
    def __init__(self, dm: DmModel, vae: VaeModel, context_encoder: Optional[Callable] = None,
                 disable_control_on_stationary: bool = False, moving_speed_th: float = 0.5, select_per_scene: bool = False):
        self.dm, self.vae = dm, vae
        self.context_encoder = context_encoder
        self.disable_control_on_stationary = disable_control_on_stationary   # config.yaml:99
        self.moving_speed_th = moving_speed_th                               # config.yaml:101
        self.select_per_scene = select_per_scene      # False = upstream's sample selection as written (see choose_action_from_guidance)
        self._guidance = None
        self._guidance_cfg = None                     # (guidance_config_list, scene_index) behind self._guidance

    def eval(self):
        return self

    def set_guidance(self, guidance_config_list, scene_index, **opt):
        """Upstream `set_guidance` (algos.py; guidance_loss.py:2106-2175): keep a guidance configuration for the following
        `get_action` calls.  `opt`: lr / optimizer / perturb_th of the optimiser step (scene_edit_config.py:74-90), `output`
        (True | dict: guidance on the t = 0 output, upstream apply_guidance_output + final_step_opt_params), `intermediate`."""
        data_batch = opt.pop("data_batch", None)         # observation fields scene-coupled losses read (agent_collision: extent, world_from_agent, curr_speed)
        self._guidance = dict(guidance_from_config(guidance_config_list, scene_index, data_batch=data_batch), **opt)
        self._guidance_cfg = (guidance_config_list, torch.as_tensor(scene_index).reshape(-1).cpu())

    def clear_guidance(self):
        self._guidance = None
        self._guidance_cfg = None

    def _guide_losses(self, traj, g, B: int, N: int, from_cfg: bool):
        """-> (guide_losses dict name -> [B,N] as upstream keys them, per-scene loss names) from the library's per-agent values;
        `g`: the guidance dict with its per-agent tensors repeated num_samp times."""
        has_builtin = any(g.get(k) is not None for k in ("target_speed", "speed_limit", "acc_limit", "target_pos"))
        vals = (self.vae.engine.guidance_losses(traj.reshape(B * N, 52, 6), g).reshape(B, N, 4) if has_builtin
                else torch.full((B, N, 4), float("nan"), device=traj.device))
        nan = torch.full((B, N), float("nan"), device=vals.device)
        colv = mapv = None
        if g.get("agent_collision") is not None:     # per-agent values as upstream files them (unweighted; :2166-2168)
            colv = self.vae.engine.agent_collision(traj.reshape(B * N, 52, 6), dict(g["agent_collision"], num_samp=N), want_grad=False).reshape(B, N)
        if g.get("map_collision") is not None:
            mapv = self.vae.engine.map_collision(traj.reshape(B * N, 52, 6), dict(g["map_collision"], num_samp=N), want_grad=False).reshape(B, N)
        out, names = {}, []
        if from_cfg:
            cfg_list, scene_index = self._guidance_cfg
            _, local = torch.unique_consecutive(scene_index, return_inverse=True)
            for si, cfgs in enumerate(cfg_list):
                members = torch.nonzero(local == si).reshape(-1)
                names.append([c["name"] for c in cfgs])
                for gi, c in enumerate(cfgs):
                    idx = members if c.get("agents") is None else members[torch.as_tensor(c["agents"], dtype=torch.long)]
                    mask = torch.zeros(B, dtype=torch.bool)
                    mask[idx] = True
                    mask = mask.to(vals.device)
                    v = colv if c["name"] == "agent_collision" else (mapv if c["name"] == "map_collision" else vals[..., LOSS_COLUMN[c["name"]]])
                    out["%s_scene_%03d_%02d" % (c["name"], si, gi)] = torch.where(mask[:, None], v, nan)
        else:       # a plain `guidance=` dict: one scene, one entry per active term
            names.append([])
            for nm, col in (("target_speed", 0), ("speed_limit", 1), ("acc_limit", 2), ("target_pos", 3)):
                if bool((~torch.isnan(vals[..., col])).any()):
                    out["%s_scene_000_%02d" % (nm, len(names[0]))] = vals[..., col]
                    names[0].append(nm)
            if colv is not None:
                out["agent_collision_scene_000_%02d" % len(names[0])] = colv
                names[0].append("agent_collision")
            if mapv is not None:
                out["map_collision_scene_000_%02d" % len(names[0])] = mapv
                names[0].append("map_collision")
        return out, names

    @torch.no_grad()
    def get_action(self, obs_dict: Mapping, num_action_samples: int = 1, class_free_guide_w: float = 0.0,
                   guide_as_filter_only: bool = False, guide_with_gt: bool = False, guide_clean=False,
                   step_index: int = 0, noise: Optional[Mapping] = None, guidance: Optional[Mapping] = None, plan=None, **kwargs):
        """`DiffuserTrafficModel.get_action` (src/tbsim/algos/algos.py:2024-2099).  With guidance active (`set_guidance` or
        `guidance=`) the guidance losses of every sample are evaluated on the final output (diffuser.py:924-926), returned as
        info['guide_losses'], and the executed sample is the one `choose_action_from_guidance` picks (algos.py:2057-2064);
        `guide_as_filter_only`: sample without guidance and only filter (algos.py:1815); `guide_with_gt`: the sample closest to
        obs_dict['target_positions'] (algos.py:2055-2056); `guide_clean=True`: the guidance steps act on the model's clean
        prediction (diffuser.py:866-873; its "video_diff" variant, which re-derives the posterior, is not built and raises).  A
        collision config (`agent_collision`, guidance_loss.py:442-630) is filed under guide_losses and makes the sample choice
        scene-level, as upstream's SCENE_LEVEL_LOSSES do.  Not built: `plan` -- it raises instead of being ignored."""
        if guide_clean not in (False, True, 0, 1, None):
            raise NotImplementedError(f"guide_clean={guide_clean!r}: only the boolean form is built (diffuser.py:866-873)")
        if plan is not None:
            raise NotImplementedError("plan conditioning is not part of the CLD sampler")
        if kwargs:
            raise TypeError(f"get_action: unexpected arguments {sorted(kwargs)}")
        eng = self.vae.engine
        if "cond_feat" in obs_dict:
            aux = dict(obs_dict)
        else:                                                       # obs -> aux_info (vae_model.py:84-88 pre_vae)
            enc = self.context_encoder or self.vae.context_encoder
            aux = dict(enc(obs_dict, include_class_free_cond=True) if class_free_guide_w != 0.0 and enc is self.vae.context_encoder
                       else enc(obs_dict))
        cond, cs = aux["cond_feat"], aux["curr_states"]
        if class_free_guide_w != 0.0 and aux.get("non_cond_feat") is None:
            aux["non_cond_feat"] = eng.non_cond_feat(cs)            # upstream builds it itself (diffuser.py:390-411,459-471); raises without the ContextEncoder weights
        B, N = cond.shape[0], int(num_action_samples)
        g = guidance if guidance is not None else self._guidance
        if g is not None and guide_clean:
            g = dict(g, guide_clean=True)
        out = self.dm({"history_positions": cond}, {k: aux[k] for k in ("cond_feat", "curr_states", "non_cond_feat") if k in aux},
                      {"num_samp": N}, noise=noise, class_free_guide_w=class_free_guide_w,
                      guidance=None if guide_as_filter_only else g)
        a = out["aux_info"]
        traj = eng.decode(out["pred_traj"], a["cond_feat"], a["curr_states"], descaled_output=True).reshape(B, N, 52, 6)
        pos, yaw = traj[..., :2].clone(), traj[..., 3:4].clone()
        act_idx = torch.zeros(B, dtype=torch.long, device=pos.device)      # "arbitrarily use the first sample", algos.py:2053-2054
        info = {}
        if guide_with_gt and "target_positions" in obs_dict:
            act_idx = choose_action_from_gt(pos, obs_dict["target_positions"], obs_dict["target_availabilities"])
        elif g is not None:
            from_cfg = g is self._guidance and self._guidance_cfg is not None
            g_rep = repeat_guidance({k: v for k, v in g.items() if k != "curr_states"}, N, a["curr_states"])
            losses, names = self._guide_losses(traj, g_rep, B, N, from_cfg)
            scene_of_agent = None
            if self.select_per_scene and from_cfg:
                _, scene_of_agent = torch.unique_consecutive(self._guidance_cfg[1], return_inverse=True)
            if losses:
                act_idx = choose_action_from_guidance(losses, names, per_scene=scene_of_agent is not None, scene_of_agent=scene_of_agent)
            info["guide_losses"] = losses
        ar = torch.arange(B, device=pos.device)
        executed = traj[ar, act_idx].clone()
        if self.disable_control_on_stationary:                      # algos.py:2076-2083
            still = (cs[:, 2].abs() < self.moving_speed_th).to(pos.device)
            pos[still] = 0
            yaw[still] = 0
            executed[still] = executed[still] * torch.tensor([0.0, 0.0, 1.0, 0.0, 1.0, 1.0], device=pos.device)   # x, y, yaw of the executed plan
        info.update(action_samples=Action(pos, yaw).to_dict(), trajectories=traj, act_idx=act_idx, executed_trajectory=executed)
        return Action(pos[ar, act_idx], yaw[ar, act_idx]), info


def closed_loop_rollout(policy: CldPolicy, cond_fn: Callable, centroid, yaw, curr_states, n_sim_steps: int,
                        n_step_action: int = 5, gather: Optional[Callable] = None, timers=None, **get_action_kwargs):
    """The loop of `rollout_episodes` (`src/tbsim/utils/env_utils.py:255-304`) kept on the device:
    obs -> get_action -> take `n_step_action` steps of the plan -> new world pose -> re-plan.
    `cond_fn(step, world [B,3], curr_states [B,4], plans) -> cond_feat [B,256]` stands in for the observation +
    ContextEncoder stage (it may return the raw observation batch instead: a dict with `image`, `history_*`,
    `curr_speed`, which get_action runs through the ContextEncoder); `plans` is what `gather` returned for the previous
    sim step -- every rank's executed plans [B_all,52,6], the neighbour information the next observation is built from
    (None on the first step or without `gather`); three-argument callables are accepted too.  `gather(traj)` (e.g.
    `parallel.gather_trajectories`) runs once per sim step; `timers` (`cld_amd.timer.Timers`) collects the per-phase
    times under the reference's keys "obs" / "network" / "env_step" / "step" (env_utils.py:268-298).
    The world moves on the EXECUTED trajectory (the selected sample, stationary agents held in place).
    Returns the world poses after each sim step [n_sim_steps, B, 3]."""
    import inspect
    from contextlib import nullcontext
    tm = (lambda k: timers.timed(k)) if timers is not None else (lambda k: nullcontext())
    eng = policy.vae.engine
    world = torch.cat([torch.as_tensor(centroid), torch.as_tensor(yaw)[:, None]], dim=1).to(eng.device, torch.float32)
    cs = torch.as_tensor(curr_states).to(eng.device, torch.float32)
    try:
        four = len(inspect.signature(cond_fn).parameters) >= 4
    except (TypeError, ValueError):
        four = False
    poses, plans = [], None
    for step in range(n_sim_steps):
        with tm("step"):
            with tm("obs"):
                o = cond_fn(step, world, cs, plans) if four else cond_fn(step, world, cs)
                obs = o if isinstance(o, Mapping) else {"cond_feat": o, "curr_states": cs}
            with tm("network"):
                _, info = policy.get_action(obs, step_index=step, **get_action_kwargs)
                traj = info["executed_trajectory"].contiguous()
            with tm("env_step"):
                if gather is not None:
                    plans = gather(traj)
                world, cs = eng.world_step(traj, world[:, :2].contiguous(), world[:, 2].contiguous(), n_step_action - 1)
        poses.append(world)
    return torch.stack(poses)


def guidance_from_config(guidance_config_list, scene_index, horizon: int = 52, data_batch: Optional[Mapping] = None) -> dict:
    """Upstream's guidance configuration -> the `guidance=` dict of `DmModel.forward` / `Engine.sample`.

    `guidance_config_list` is what `DiffuserTrafficModel.set_guidance` takes (`src/tbsim/algos/algos.py`, built by
    `DiffuserGuidance`, `src/tbsim/utils/guidance_loss.py:2106-2175`): one list per scene of
    `{'name', 'weight', 'params', 'agents'}` dicts; `scene_index [B]` maps agents to scenes (consecutive runs).  Each loss
    is averaged over the agents it applies to within its scene and multiplied by its weight; that is folded into the
    per-agent scales of the kernel: weight / (agents * horizon) for the per-step losses, weight / agents for the waypoint
    losses.  Supported names: target_speed, speed_limit, acc_limit, target_pos_at_time, target_pos, agent_collision (needs
    `data_batch` with `extent`, `world_from_agent`, `curr_speed`: the observation fields upstream's loss reads,
    guidance_loss.py:506-510) and map_collision (`extent`, `raster_from_agent`, `drivable_map`, `curr_speed`, :773-775); at most
    one of each per scene, one parameter set per call; the others (social groups, stop signs, lane keeping, ...) are not built.  One speed / acceleration limit value per call."""
    scene_index = torch.as_tensor(scene_index).reshape(-1).cpu()
    B = scene_index.numel()
    _, local = torch.unique_consecutive(scene_index, return_inverse=True)
    if len(guidance_config_list) != int(local.max()) + 1:
        raise ValueError("guidance config list must hold one entry per scene")
    out: dict = {}
    ts_scale = torch.zeros(B); ts = torch.zeros(B, horizon); has_ts = False
    sl_scale = torch.zeros(B); al_scale = torch.zeros(B); sl = al = None
    tp = torch.zeros(B, 2); tt = torch.zeros(B, dtype=torch.int32); tp_scale = torch.zeros(B); has_tp = False
    col = mcol = None
    for si, cfgs in enumerate(guidance_config_list):
        members = torch.nonzero(local == si).reshape(-1)
        for cfg in cfgs:
            name, wgt, prm, agents = cfg["name"], float(cfg["weight"]), cfg["params"], cfg.get("agents")
            idx = members if agents is None else members[torch.as_tensor(agents, dtype=torch.long)]
            n = max(1, idx.numel())
            if name == "target_speed":       # params['target_speed'] is indexed by the whole batch (guidance_loss.py:231-241)
                if bool((ts_scale[idx] != 0).any()):
                    raise ValueError("two target_speed losses on one agent")
                full = torch.as_tensor(prm["target_speed"], dtype=torch.float32)
                ts[idx] = full[idx, :horizon]
                ts_scale[idx] = wgt / (n * horizon)
                has_ts = True
            elif name in ("speed_limit", "acc_limit"):
                val = float(prm[name])
                if name == "speed_limit":
                    if sl is not None and sl != val:
                        raise ValueError("one speed_limit value per call")
                    sl = val; sl_scale[idx] += wgt / (n * horizon)
                else:
                    if al is not None and al != val:
                        raise ValueError("one acc_limit value per call")
                    al = val; al_scale[idx] += wgt / (n * horizon)
            elif name in ("target_pos_at_time", "target_pos"):
                if bool((tp_scale[idx] != 0).any()):
                    raise ValueError("two waypoint losses on one agent")
                tp[idx] = torch.as_tensor(prm["target_pos"], dtype=torch.float32).reshape(-1, 2)
                if name == "target_pos_at_time":
                    tt[idx] = torch.as_tensor(prm["target_time"], dtype=torch.int32).reshape(-1)
                else:                        # any step >= int(min_target_time * horizon): encoded as -(m + 1)
                    tt[idx] = -(int(float(prm.get("min_target_time", 0.0)) * horizon) + 1)
                tp_scale[idx] = wgt / n
                has_tp = True
            elif name == "agent_collision":
                if data_batch is None or any(k not in data_batch for k in ("extent", "world_from_agent", "curr_speed")):
                    raise ValueError("agent_collision needs data_batch with extent, world_from_agent and curr_speed")
                S = int(local.max()) + 1
                key = (int(prm.get("num_disks", 5)), float(prm.get("buffer_dist", 0.2)), float(prm.get("decay_rate", 0.9)), float(prm.get("guide_moving_speed_th", 0.5)))
                if col is None:
                    col = dict(extent=data_batch["extent"], world_from_agent=data_batch["world_from_agent"], curr_speed=data_batch["curr_speed"],
                               scene_index=scene_index, weight=[0.0] * S, agents={}, num_disks=key[0], buffer_dist=key[1], decay_rate=key[2],
                               guide_moving_speed_th=key[3])
                elif (col["num_disks"], col["buffer_dist"], col["decay_rate"], col["guide_moving_speed_th"]) != key:
                    raise ValueError("one agent_collision parameter set per call")
                if col["weight"][si] != 0.0:
                    raise ValueError("two agent_collision losses on one scene")
                col["weight"][si] = wgt
                if agents is not None:
                    col["agents"][si] = list(agents)
                if prm.get("excluded_agents") is not None:
                    # batch indices, as upstream takes them (guidance_loss.py:447,586-593); each config's loss sees the pairs of ITS scene
                    # only, so a listed agent of another scene changes nothing there: keep this scene's
                    mem = set(int(b) for b in members)
                    col.setdefault("excluded_agents", []).extend(int(b) for b in prm["excluded_agents"] if int(b) in mem)
            elif name == "map_collision":
                need = ("extent", "raster_from_agent", "drivable_map", "curr_speed")
                if data_batch is None or any(k not in data_batch for k in need):
                    raise ValueError("map_collision needs data_batch with extent, raster_from_agent, drivable_map and curr_speed")
                if agents is not None:
                    raise NotImplementedError("map_collision on an `agents` subset (upstream itself cannot run it: guidance_loss.py:857-859)")
                S = int(local.max()) + 1
                key = (tuple(int(v) for v in prm.get("num_points_lw", (10, 10))), float(prm.get("decay_rate", 0.9)), float(prm.get("guide_moving_speed_th", 0.5)))
                if mcol is None:
                    mcol = dict({k: data_batch[k] for k in need}, scene_index=scene_index, weight=[0.0] * S, num_points_lw=key[0], decay_rate=key[1],
                                guide_moving_speed_th=key[2])
                elif (tuple(mcol["num_points_lw"]), mcol["decay_rate"], mcol["guide_moving_speed_th"]) != key:
                    raise ValueError("one map_collision parameter set per call")
                if mcol["weight"][si] != 0.0:
                    raise ValueError("two map_collision losses on one scene")
                mcol["weight"][si] = wgt
            else:
                raise NotImplementedError(f"guidance loss '{name}' is not built (target_speed, speed_limit, acc_limit, "
                                          f"target_pos_at_time, target_pos, agent_collision, map_collision are)")
    if has_ts:
        out["target_speed"], out["loss_scale"] = ts, ts_scale
    if sl is not None:
        out["speed_limit"] = (sl, sl_scale)
    if al is not None:
        out["acc_limit"] = (al, al_scale)
    if has_tp:
        out["target_pos"] = (tp, tt, tp_scale)
    if col is not None:
        col["agents"] = col["agents"] or None
        out["agent_collision"] = col
    if mcol is not None:
        out["map_collision"] = mcol
    if not out:
        raise ValueError("no guidance loss configured")
    return out
