"""Engine: one libcld_hip handle + its device workspace, driven with torch CUDA tensors.

PyTorch is plumbing here (device memory, streams); all arithmetic of the path
runs in the HIP library behind the C-ABI of include/cld.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import numpy as np
import torch

from . import _lib
from ._lib import CldConfig, CldError

T, D, COND = 52, 4, 256


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """Owns a `cld_handle`.  Weights come in under the reference's state_dict names."""

    def __init__(self, n_timesteps: int = 100, device="cuda:0", dynamics: Optional[Mapping] = None,
                 norm_info=None, step_time: float = 0.1, precision: str = "f32"):
        """precision: "f32" (exact fp32 MFMA) or "f16x2" (fp16 hi/lo split operands, fp32 accumulate; include/cld.h)."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise CldError("the CLD sampling path runs on an MI355X only (device must be cuda:N); no CPU fallback")
        if not torch.cuda.is_available():
            raise CldError("no HIP device visible; the CLD sampling path has no CPU fallback")
        cfg = CldConfig()
        self.lib.cld_default_config(C.byref(cfg))
        cfg.n_timesteps = int(n_timesteps)
        cfg.step_time = float(step_time)
        if precision not in _lib.PRECISIONS:
            raise CldError(f"unknown precision '{precision}' (f32 | f16x2)")
        cfg.precision = _lib.PRECISIONS[precision]
        if dynamics is not None:      # config.yaml:134-141
            if "acce_bound" in dynamics:
                cfg.acce_bound[0], cfg.acce_bound[1] = map(float, dynamics["acce_bound"])
            if "max_steer" in dynamics:
                cfg.max_steer = float(dynamics["max_steer"])
            if "max_yawvel" in dynamics:
                cfg.max_yawvel = float(dynamics["max_yawvel"])
        if norm_info is not None:     # config.yaml:161-164: [mean(6), std(6)]
            for i in range(6):
                cfg.norm_mean[i] = float(norm_info[0][i])
                cfg.norm_std[i] = float(norm_info[1][i])
        self.cfg = cfg
        self.n_timesteps = int(n_timesteps)
        self.stride = 1
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.cld_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise CldError(f"cld_create failed ({rc})")
        self._ws = None
        self._ctx_ws = None
        self._nc_map_feat = {}
        self._finalized = False
        self.precision = {v: k for k, v in _lib.PRECISIONS.items()}[int(self.lib.cld_get_precision(self._h))]
        n = self.n_timesteps
        xc, nc, lv = (np.empty(n, np.float32) for _ in range(3))
        self._check(self.lib.cld_get_schedule(self._h, xc.ctypes.data, nc.ctypes.data, lv.ctypes.data), "cld_get_schedule")
        self.x_t_cof, self.noise_cof, self.posterior_log_variance_clipped = xc, nc, lv

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc, what):
        _lib.check(self._h, rc, what)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, B: int):
        need = int(self.lib.cld_workspace_bytes(self._h, B))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return C.c_void_p(self._ws.data_ptr()), C.c_size_t(self._ws.numel())

    def _f32(self, t: torch.Tensor, shape=None) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise CldError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.cld_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_stride(self, stride: int):
        """DmModel.stride (dm_model.py:25,119): the sampling loop visits i in reversed(range(0, n_timesteps, stride))."""
        self._check(self.lib.cld_set_stride(self._h, int(stride)), "cld_set_stride")
        self.stride = int(stride)

    def force_kernel(self, which: str, form: str = "auto"):
        """Tests only (cld_debug_force_kernel): run the "guide" / "decode" / "encode" kernel of this engine in its "valu" or
        "mfma" formulation instead of letting the batch size pick ("auto")."""
        if which not in _lib.KERNELS or form not in _lib.FORMS:
            raise CldError(f"force_kernel: unknown kernel '{which}' or formulation '{form}'")
        self._check(self.lib.cld_debug_force_kernel(self._h, _lib.KERNELS[which], _lib.FORMS[form]), "cld_debug_force_kernel")

    @property
    def loop_steps(self) -> int:
        return len(range(0, self.n_timesteps, self.stride))

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Mapping, strict: bool = True):
        """Accepts reference keys ('model.*', 'lstm_dec.*'; 'dm.' / 'vae.lstmvae.' prefixes stripped)."""
        if self._finalized:
            raise CldError("weights already finalized")
        for k, v in sd.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            rc = self.lib.cld_load_weight(self._h, k.encode(), a.ctypes.data, a.size)
            if rc != 0 and strict:
                self._check(rc, f"cld_load_weight('{k}')")
        return self

    def finalize(self):
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_finalize(self._h, self._stream()), "cld_finalize")
        self._finalized = True
        return self

    # ------------------------------------------------------------------ compute
    def unet_forward(self, x, cond, t_idx: int):
        x = self._f32(x)
        B = x.shape[0]
        x = self._f32(x, (B, T, D)); cond = self._f32(cond, (B, COND))
        eps = torch.empty_like(x)
        ws, wsn = self._workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_unet_forward(self._h, _ptr(x), _ptr(cond), int(t_idx), _ptr(eps), B, ws, wsn,
                                                  self._stream()), "cld_unet_forward")
        return eps

    def unet_forward_rows(self, x, cond, t):
        """U-Net forward with one timestep per row: t [B] (any integer tensor)."""
        x = self._f32(x)
        B = x.shape[0]
        x = self._f32(x, (B, T, D)); cond = self._f32(cond, (B, COND))
        t = self._timesteps(t, B)
        eps = torch.empty_like(x)
        ws, wsn = self._workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_unet_forward_t(self._h, _ptr(x), _ptr(cond), _ptr(t), _ptr(eps), B, ws, wsn, self._stream()),
                        "cld_unet_forward_t")
        return eps

    #: Device-resident timesteps are CLAMPED to [0, n_timesteps) without a check, to keep the call asynchronous (host values always raise
    #: when out of range).  Set True to have device tensors checked too -- a synchronising read per call; for debugging a caller that
    #: draws timesteps itself (e.g. randint(0, n + 1) would otherwise be mapped to a valid step silently).  INTEGRATION.md, "Timesteps".
    check_device_timesteps = False

    def _timesteps(self, t, B: int) -> torch.Tensor:
        """Per-row timesteps -> int32 [B] on the device.  Host values (lists, CPU tensors) are range-checked here; a device
        tensor is clamped to [0, n_timesteps) on the device instead, so the call stays asynchronous (the kernels index tables
        with these values; training-style callers draw them with torch.randint on the device, dm_model.py:84)."""
        t = torch.as_tensor(t).reshape(-1)
        if t.numel() != B:
            raise CldError(f"expected {B} timesteps, got {t.numel()}")
        if t.device.type == "cpu":
            if t.numel() and (int(t.min()) < 0 or int(t.max()) >= self.n_timesteps):
                raise CldError(f"timestep out of range [0, {self.n_timesteps})")
            return t.to(self.device, torch.int32).contiguous()
        if self.check_device_timesteps:      # opt-in (Engine.check_device_timesteps = True): costs a device round trip per call
            lo, hi = int(t.min()), int(t.max())
            if lo < 0 or hi >= self.n_timesteps:
                raise CldError(f"timestep out of range [0, {self.n_timesteps}): device tensor holds values in [{lo}, {hi}]")
        return t.to(self.device).clamp(0, self.n_timesteps - 1).to(torch.int32).contiguous()

    def q_sample(self, z0, noise, t):
        """DmModel.q_sample (dm_model.py:91-96): sqrt(acp[t]) z0 + sqrt(1 - acp[t]) noise, per-row t."""
        z0 = self._f32(z0)
        B = z0.shape[0]
        z0 = self._f32(z0, (B, T, D)); noise = self._f32(noise, (B, T, D))
        t = self._timesteps(t, B)
        zn = torch.empty_like(z0)
        ws, wsn = self._workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_denoise_loss(self._h, _ptr(z0), _ptr(noise), None, _ptr(t), _ptr(zn), None, B,
                                                  ws, wsn, self._stream()), "cld_denoise_loss")
        return zn

    def denoise_loss(self, z0, noise, cond, t, want_z_noisy=False):
        """Forward half of DmModel.compute_losses (dm_model.py:82-96): per-sample MSE [B] between `noise` and the U-Net's
        prediction on q_sample(z0, t, noise); the reference's scalar loss is its mean."""
        z0 = self._f32(z0)
        B = z0.shape[0]
        z0 = self._f32(z0, (B, T, D)); noise = self._f32(noise, (B, T, D)); cond = self._f32(cond, (B, COND))
        t = self._timesteps(t, B)
        mse = torch.empty(B, dtype=torch.float32, device=self.device)
        zn = torch.empty_like(z0) if want_z_noisy else None
        ws, wsn = self._workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_denoise_loss(self._h, _ptr(z0), _ptr(noise), _ptr(cond), _ptr(t), _ptr(zn), _ptr(mse), B,
                                                  ws, wsn, self._stream()), "cld_denoise_loss")
        return (mse, zn) if want_z_noisy else mse

    def ddpm_step(self, x, cond, t_idx: int, z):
        x = self._f32(x)
        B = x.shape[0]
        x = self._f32(x, (B, T, D)); cond = self._f32(cond, (B, COND))
        z = None if z is None else self._f32(z, (B, T, D))
        xn, mean = torch.empty_like(x), torch.empty_like(x)
        sigma = C.c_float()
        ws, wsn = self._workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_ddpm_step(self._h, _ptr(x), _ptr(cond), int(t_idx), _ptr(z), _ptr(xn), _ptr(mean),
                                               C.byref(sigma), B, ws, wsn, self._stream()), "cld_ddpm_step")
        return xn, mean, float(sigma.value)

    def _guidance(self, g: Mapping, B: int):
        """dict(curr_states [B,4], target_speed [B,52] | None, loss_scale [B] | None, speed_limit (limit, scale) | None,
        acc_limit (limit, scale) | None, target_pos (pos [B,2], time index [B], scale) | None, lr | None, perturb_th | None | "sigma", optimizer "adam" | "sgd",
        grad_steps = 1, guide_clean = False, agent_collision: dict | None (see _collision), map_collision: dict | None (see _map_collision))
        -> (CldGuidance, tensors kept alive).  A `scale` is a per-agent tensor [B] (weight / (agents of the scene * 52),
        as DiffuserGuidance averages) or a scalar weight (divided by 52 here).  lr None = sigma_t; perturb_th None = no clip (what
        the reference's perturb() does), "sigma" = clip to sigma_t, a number = clip to it (include/cld.h)."""
        cs = self._f32(g["curr_states"], (B, 4))
        ts = None if g.get("target_speed") is None else self._f32(g["target_speed"], (B, T))
        ls = None if g.get("loss_scale") is None else self._f32(g["loss_scale"], (B,))
        # optional SpeedLimitLoss / AccLimitLoss terms: (limit, per-agent scale [B] or a scalar weight)
        def term(key):
            v = g.get(key)
            if v is None:
                return 0.0, None
            lim, sc = v
            sc = torch.full((B,), float(sc) / T, device=self.device) if not isinstance(sc, torch.Tensor) and np.isscalar(sc) else self._f32(sc, (B,))
            return float(lim), sc
        sl, sls = term("speed_limit")
        al, als = term("acc_limit")
        eg = None if g.get("ext_grad") is None else self._f32(g["ext_grad"], (B, T, 6))
        tp = tt = tps = None
        if g.get("target_pos") is not None:      # (positions [B,2], time index [B], per-agent scale [B] | scalar weight): TargetPosAtTimeLoss
            pos_, time_, sc = g["target_pos"]
            tp = self._f32(pos_, (B, 2))
            tt = torch.as_tensor(time_).to(self.device, torch.int32).contiguous()
            if tuple(tt.shape) != (B,):
                raise CldError(f"target_pos time index: expected shape ({B},), got {tuple(tt.shape)}")
            tps = torch.full((B,), float(sc), device=self.device) if not isinstance(sc, torch.Tensor) and np.isscalar(sc) else self._f32(sc, (B,))
        th = g.get("perturb_th")
        opt = g.get("optimizer", "adam")
        if opt not in _lib.OPTIMIZERS:
            raise CldError(f"unknown guidance optimizer '{opt}' (adam | sgd)")
        # guidance on the t = 0 output (upstream apply_guidance_output + final_step_opt_params, scene_edit_config.py:84-91):
        # "output": True | dict(lr, perturb_th, optimizer); "intermediate": False switches the t > 0 steps off
        fo = g.get("output")
        fo = {} if fo is True else fo
        fopt = (fo or {}).get("optimizer", "adam")
        if fopt not in _lib.OPTIMIZERS:
            raise CldError(f"unknown guidance optimizer '{fopt}' (adam | sgd)")
        fth = (fo or {}).get("perturb_th", None)       # None = no clip: what upstream's perturb() does (guidance_loss.py:2237,2273-2276)
        col = ckeep = mcol = mkeep = None
        if g.get("agent_collision") is not None:
            col, ckeep = self._collision(g["agent_collision"], B)
        if g.get("map_collision") is not None:
            mcol, mkeep = self._map_collision(g["map_collision"], B)
        cg = _lib.CldGuidance(cs.data_ptr(), None if ts is None else ts.data_ptr(), None if ls is None else ls.data_ptr(),
                              float(g["lr"]) if g.get("lr") else 0.0,
                              -1.0 if th is None else (0.0 if th == "sigma" else float(th)), _lib.OPTIMIZERS[opt],
                              sl, al, None if sls is None else sls.data_ptr(), None if als is None else als.data_ptr(),
                              None if tp is None else tp.data_ptr(), None if tt is None else tt.data_ptr(),
                              None if tps is None else tps.data_ptr(), None if eg is None else eg.data_ptr(),
                              0 if fo is None or fo is False else 1, 0 if g.get("intermediate", True) else 1,
                              float((fo or {}).get("lr", 0.3) or 0.0),
                              -1.0 if fth is None else (0.0 if fth == "sigma" else float(fth)), _lib.OPTIMIZERS[fopt],
                              int(g.get("grad_steps", 1) or 1), int((fo or {}).get("grad_steps", 1) or 1), 1 if g.get("guide_clean") else 0,
                              None if col is None else C.addressof(col), None if mcol is None else C.addressof(mcol))
        return cg, (cs, ts, ls, sls, als, tp, tt, tps, eg, col, ckeep, mcol, mkeep)

    def _collision(self, c: Mapping, B: int):
        """dict(extent [A,3], world_from_agent [A,3,3], curr_speed [A], scene_index [A] (consecutive blocks) | scene_sizes,
        weight: scalar or per-scene sequence (0 = scene not guided), agents: optional {scene: local indices} (upstream's
        `agents` of a guidance config), excluded_agents: optional batch indices (upstream's `excluded_agents`), num_samp = 1, num_disks = 5, buffer_dist = 0.2, decay_rate = 0.9,
        guide_moving_speed_th = 0.5) -> (CldCollision, tensors kept alive): upstream's AgentCollisionLoss
        (src/tbsim/utils/guidance_loss.py:442-630) configured per scene as DiffuserGuidance does (:2106-2172).  B = A * num_samp."""
        N = int(c.get("num_samp", 1))
        if B % N:
            raise CldError(f"agent_collision: {B} rows are not a multiple of num_samp = {N}")
        A = B // N
        ext = self._f32(c["extent"], (A, 3)); wfa = self._f32(c["world_from_agent"], (A, 3, 3)); spd = self._f32(c["curr_speed"], (A,))
        if c.get("scene_sizes") is not None:
            sizes = [int(v) for v in c["scene_sizes"]]
        else:
            si = torch.as_tensor(c["scene_index"]).cpu()
            _, counts = torch.unique_consecutive(si, return_counts=True)
            sizes = [int(v) for v in counts]
        if sum(sizes) != A or min(sizes) < 1:
            raise CldError(f"agent_collision: scene sizes {sizes} do not cover the {A} agents")
        S = len(sizes)
        start = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=self.device)
        wv = c.get("weight", 1.0)
        wts = torch.full((S,), float(wv), device=self.device) if np.isscalar(wv) else self._f32(torch.as_tensor(wv, dtype=torch.float32), (S,))
        guided = None
        if c.get("agents"):
            gm = np.zeros(A, np.uint8)
            offs = np.concatenate([[0], np.cumsum(sizes)])
            for s_ in range(S):
                sub = c["agents"].get(s_)
                if sub is None:
                    gm[offs[s_]:offs[s_ + 1]] = 1
                else:
                    gm[offs[s_] + np.asarray(sub, dtype=np.int64)] = 1
            guided = torch.from_numpy(gm).to(self.device)
        excluded = None
        if c.get("excluded_agents") is not None:        # upstream's `excluded_agents` (:447,586-593): batch indices; a pair of two flagged agents is not penalised
            em = np.zeros(A, np.uint8)
            idx = np.asarray(list(c["excluded_agents"]), dtype=np.int64).reshape(-1)
            if idx.size and (idx.min() < 0 or idx.max() >= A):
                raise CldError(f"agent_collision: excluded_agents out of range for {A} agents")
            em[idx] = 1
            excluded = torch.from_numpy(em).to(self.device)
        cc = _lib.CldCollision(ext.data_ptr(), wfa.data_ptr(), spd.data_ptr(), start.data_ptr(), wts.data_ptr(),
                               None if guided is None else guided.data_ptr(), S, N, int(c.get("num_disks", 5)), max(sizes),
                               float(c.get("buffer_dist", 0.2)), float(c.get("decay_rate", 0.9)), float(c.get("guide_moving_speed_th", 0.5)),
                               None if excluded is None else excluded.data_ptr())
        return cc, (ext, wfa, spd, start, wts, guided, excluded)

    def _scene_blocks(self, c: Mapping, A: int, what: str):
        """-> (scene sizes, start offsets on the device, per-scene weights on the device) of a scene-structured loss config."""
        if c.get("scene_sizes") is not None:
            sizes = [int(v) for v in c["scene_sizes"]]
        elif c.get("scene_index") is not None:
            _, counts = torch.unique_consecutive(torch.as_tensor(c["scene_index"]).cpu(), return_counts=True)
            sizes = [int(v) for v in counts]
        else:
            sizes = [A]
        if sum(sizes) != A or min(sizes) < 1:
            raise CldError(f"{what}: scene sizes {sizes} do not cover the {A} agents")
        start = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=self.device)
        wv = c.get("weight", 1.0)
        wts = torch.full((len(sizes),), float(wv), device=self.device) if np.isscalar(wv) else self._f32(torch.as_tensor(wv, dtype=torch.float32), (len(sizes),))
        return sizes, start, wts

    def _map_collision(self, c: Mapping, B: int):
        """dict(extent [A,3], raster_from_agent [A,3,3], drivable_map [A,H,W] (non-zero = drivable), curr_speed [A], scene_index |
        scene_sizes (default: one scene), weight: scalar or per-scene sequence, num_samp = 1, num_points_lw = (10, 10),
        decay_rate = 0.9, guide_moving_speed_th = 0.5) -> (CldMapCollision, tensors kept alive): upstream's MapCollisionLoss
        (src/tbsim/utils/guidance_loss.py:717-875)."""
        N = int(c.get("num_samp", 1))
        if B % N:
            raise CldError(f"map_collision: {B} rows are not a multiple of num_samp = {N}")
        A = B // N
        ext = self._f32(c["extent"], (A, 3)); rfa = self._f32(c["raster_from_agent"], (A, 3, 3)); spd = self._f32(c["curr_speed"], (A,))
        dm = torch.as_tensor(c["drivable_map"])
        if dm.dim() != 3 or dm.shape[0] != A:
            raise CldError(f"map_collision: drivable_map must be [{A},H,W], got {tuple(dm.shape)}")
        if not (dm.dtype == torch.uint8 and dm.device == self.device):     # (a resident uint8 map is taken as it is: non-zero = drivable)
            dm = (dm != 0).to(self.device, torch.uint8)
        dm = dm.contiguous()
        sizes, start, wts = self._scene_blocks(c, A, "map_collision")
        nl, nw = (int(v) for v in c.get("num_points_lw", (10, 10)))
        cc = _lib.CldMapCollision(ext.data_ptr(), rfa.data_ptr(), dm.data_ptr(), spd.data_ptr(), start.data_ptr(), wts.data_ptr(),
                                  len(sizes), N, int(dm.shape[1]), int(dm.shape[2]), nl, nw, float(c.get("decay_rate", 0.9)),
                                  float(c.get("guide_moving_speed_th", 0.5)))
        return cc, (ext, rfa, dm, spd, start, wts)

    def map_collision(self, traj, cfg: Mapping, grad_in=None, want_grad=True):
        """Upstream's MapCollisionLoss on decoded plans [B,52,6] (descaled, sample-minor rows) -> (per-plan values [B],
        d total / d traj [B,52,6]); cld_map_collision_loss."""
        traj = self._f32(traj)
        B = traj.shape[0]
        traj = self._f32(traj, (B, T, 6))
        cc, keep = self._map_collision(cfg, B)
        gi = None if grad_in is None else self._f32(grad_in, (B, T, 6))
        loss = torch.empty(B, dtype=torch.float32, device=self.device)
        grad = torch.empty(B, T, 6, dtype=torch.float32, device=self.device) if want_grad else None
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_map_collision_loss(self._h, _ptr(traj), C.byref(cc), _ptr(gi), _ptr(loss), _ptr(grad), B, self._stream()),
                        "cld_map_collision_loss")
        return (loss, grad) if want_grad else loss

    def agent_collision(self, traj, collision: Mapping, grad_in=None, want_grad=True):
        """Upstream's AgentCollisionLoss on decoded plans [B,52,6] (descaled, sample-minor rows) -> (per-agent values [B] as
        upstream files them under guide_losses, d total / d traj [B,52,6]); cld_agent_collision."""
        traj = self._f32(traj)
        B = traj.shape[0]
        traj = self._f32(traj, (B, T, 6))
        cc, keep = self._collision(collision, B)
        gi = None if grad_in is None else self._f32(grad_in, (B, T, 6))
        loss = torch.empty(B, dtype=torch.float32, device=self.device)
        grad = torch.empty(B, T, 6, dtype=torch.float32, device=self.device) if want_grad else None
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_agent_collision(self._h, _ptr(traj), C.byref(cc), _ptr(gi), _ptr(loss), _ptr(grad), B, self._stream()),
                        "cld_agent_collision")
        return (loss, grad) if want_grad else loss

    def guidance_losses(self, traj, guidance: Mapping):
        """Per-agent values of the built-in guidance losses on decoded trajectories [B,52,6] (descaled) -> [B,4] =
        (target_speed, speed_limit, acc_limit, waypoint), NaN where a term is off for the agent: what upstream reports as
        `guide_losses` and selects samples by (guidance_loss.py:2143-2172, algos.py:2057-2064)."""
        traj = self._f32(traj)
        B = traj.shape[0]
        traj = self._f32(traj, (B, T, 6))
        g = dict(guidance)
        g.setdefault("curr_states", torch.zeros(B, 4, device=self.device))
        cg, keep = self._guidance(g, B)
        out = torch.empty(B, 4, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_guidance_losses(self._h, _ptr(traj), C.byref(cg), _ptr(out), B, self._stream()), "cld_guidance_losses")
        return out

    def guidance_step(self, mean, cond, guidance: Mapping, sigma: float, z=None, want_grad=False):
        """One guidance step on a posterior mean [B,52,4] (upstream PerturbationGuidance.perturb, guidance_loss.py:2221-2282)
        -> guided mean (and x_next = guided + sigma z when z is given, dL/dmean when want_grad)."""
        mean = self._f32(mean)
        B = mean.shape[0]
        mean = self._f32(mean, (B, T, D)); cond = self._f32(cond, (B, COND))
        z = None if z is None else self._f32(z, (B, T, D))
        cg, keep = self._guidance(guidance, B)
        mg = torch.empty_like(mean)
        xn = torch.empty_like(mean) if z is not None else None
        gr = torch.empty_like(mean) if want_grad else None
        ws, wsn = self._workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_guidance_step(self._h, _ptr(mean), _ptr(cond), C.byref(cg), C.c_float(sigma), _ptr(z),
                                                   _ptr(mg), _ptr(xn), _ptr(gr), B, ws, wsn, self._stream()), "cld_guidance_step")
        out = (mg,) + ((xn,) if z is not None else ()) + ((gr,) if want_grad else ())
        return out[0] if len(out) == 1 else out

    def decode_vjp(self, z, cond, curr_states, grad_traj):
        """Vector-Jacobian product of `decode` (LSTM decoder + descale + unicycle roll-out, descaled output): grad_traj
        [B,52,6] = dL/dtraj -> dL/dz [B,52,4].  With it any loss evaluated on decoded trajectories (torch autograd over
        upstream's guidance losses included) can be pulled back to the latent."""
        z = self._f32(z)
        B = z.shape[0]
        gd = {"curr_states": curr_states, "ext_grad": grad_traj, "lr": 1.0, "optimizer": "sgd"}
        _, grad = self.guidance_step(z, cond, gd, sigma=0.0, want_grad=True)
        return grad

    def sample_with_loss(self, x_T, cond, curr_states, noise, loss_fn, lr: Optional[float] = 0.3, optimizer: str = "adam",
                         perturb_th=None):
        """Ancestral loop with a CALLER-DEFINED guidance loss: `loss_fn(traj [B,52,6] descaled) -> scalar` is torch code
        (e.g. upstream's guidance losses, `src/tbsim/utils/guidance_loss.py`, which couple agents or sample rasters).  Per
        step t > 0: U-Net + posterior mean (HIP), decode the mean (HIP), dL/dtraj by torch autograd over `loss_fn` only,
        then decoder + roll-out backward, the optimiser step and the noise add (HIP, `cld_guidance_step` with `ext_grad`).
        Same semantics as `sample(..., guidance=)` (upstream diffuser.py:844-929); slower, because the loop is driven from
        Python.  `noise` [steps,B,52,4] is required.  -> (x0, x1)."""
        x = self._f32(x_T)
        B = x.shape[0]
        cond = self._f32(cond, (B, COND)); cs = self._f32(curr_states, (B, 4))
        n = self.loop_steps
        noise = self._f32(noise, (n, B, T, D))
        x1 = None
        for it in range(n):
            i = (n - 1 - it) * self.stride
            xn, mean, sigma = self.ddpm_step(x, cond, i, noise[it])
            if i == 0:
                x = xn
                break
            traj = self.decode(mean, cond, cs, descaled_output=True).requires_grad_(True)
            with torch.enable_grad():
                (gtraj,) = torch.autograd.grad(loss_fn(traj), traj)
            gd = {"curr_states": cs, "ext_grad": gtraj, "lr": lr, "perturb_th": perturb_th, "optimizer": optimizer}
            _, x = self.guidance_step(mean, cond, gd, sigma, z=noise[it])
            if i == 1:
                x1 = x.clone()
        return x, x1

    def sample(self, x_T, cond, noise=None, seed: int = 0, want_x1=True, want_logp=True,
               non_cond=None, guidance_w: float = 0.0, guidance: Optional[Mapping] = None):
        """Full ancestral loop.  With `non_cond` [B,256] and guidance_w != 0: classifier-free guidance
        (eps = (1+w) eps_cond - w eps_uncond, upstream diffuser.py:787), both passes as one 2B batch per step.
        With `guidance` (see `_guidance`): the posterior mean of every step t > 0 takes one optimiser step on the
        target-speed loss through decoder + roll-out before the noise is added (upstream diffuser.py:844-929)."""
        x_T = self._f32(x_T)
        B = x_T.shape[0]
        x_T = self._f32(x_T, (B, T, D)); cond = self._f32(cond, (B, COND))
        n = self.loop_steps                 # loop iterations = noise slabs (n_timesteps with the reference's stride 1)
        noise = None if noise is None else self._f32(noise, (n, B, T, D))
        x0 = torch.empty_like(x_T)
        # x1 exists only when the loop visits step 1 (dm_model.py:126-127): stride 1 and at least two timesteps
        x1 = torch.empty_like(x_T) if (want_x1 and 1 in range(0, self.n_timesteps, self.stride)) else None
        logp = torch.empty(B, dtype=torch.float32, device=self.device) if want_logp else None
        cfg = non_cond is not None and guidance_w != 0.0
        with torch.cuda.device(self.device):
            if guidance is not None:
                cg, keep = self._guidance(guidance, B)
                non_cond = self._f32(non_cond, (B, COND)) if cfg else None
                ws, wsn = self._workspace(2 * ((B + 15) // 16 * 16) if cfg else B)
                self._check(self.lib.cld_sample_guided(self._h, _ptr(x_T), _ptr(noise), _ptr(cond), _ptr(non_cond),
                                                       C.c_float(guidance_w), C.byref(cg), n, _ptr(x0), _ptr(x1), _ptr(logp), B,
                                                       C.c_uint64(seed), ws, wsn, self._stream()), "cld_sample_guided")
            elif cfg:
                non_cond = self._f32(non_cond, (B, COND))
                ws, wsn = self._workspace(2 * ((B + 15) // 16 * 16))
                self._check(self.lib.cld_sample_cfg(self._h, _ptr(x_T), _ptr(noise), _ptr(cond), _ptr(non_cond),
                                                    C.c_float(guidance_w), n, _ptr(x0), _ptr(x1), _ptr(logp), B,
                                                    C.c_uint64(seed), ws, wsn, self._stream()), "cld_sample_cfg")
            else:
                ws, wsn = self._workspace(B)
                self._check(self.lib.cld_sample(self._h, _ptr(x_T), _ptr(noise), _ptr(cond), n, _ptr(x0), _ptr(x1),
                                                _ptr(logp), B, C.c_uint64(seed), ws, wsn, self._stream()), "cld_sample")
        return x0, x1, logp

    def sample_step(self, x_t, cond, t_idx: int, z=None, non_cond=None, guidance_w: float = 0.0, guidance: Optional[Mapping] = None,
                    want_grad: bool = False):
        """One iteration of `sample` at timestep t_idx on a given x_t (cld_sample_step; upstream p_sample, diffuser.py:844-929)
        -> dict(x_next, mean [posterior mean before guidance], sigma, and on a guided step mean_guided [, grad])."""
        x_t = self._f32(x_t)
        B = x_t.shape[0]
        x_t = self._f32(x_t, (B, T, D)); cond = self._f32(cond, (B, COND))
        z = None if z is None else self._f32(z, (B, T, D))
        cfg = non_cond is not None and guidance_w != 0.0
        non_cond = self._f32(non_cond, (B, COND)) if cfg else None
        cg = keep = None
        if guidance is not None:
            cg, keep = self._guidance(guidance, B)
        guided = guidance is not None and (bool(guidance.get("intermediate", True)) if t_idx > 0 else bool(guidance.get("output")))
        xn, mean = torch.empty_like(x_t), torch.empty_like(x_t)
        mg = torch.empty_like(x_t) if guided else None
        gr = torch.empty_like(x_t) if guided and want_grad else None
        ws, wsn = self._workspace(2 * ((B + 15) // 16 * 16) if cfg else B)
        sigma = C.c_float()
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_sample_step(self._h, _ptr(x_t), _ptr(cond), _ptr(non_cond), C.c_float(guidance_w),
                                                 None if cg is None else C.byref(cg), int(t_idx), _ptr(z), _ptr(xn), _ptr(mean),
                                                 _ptr(mg), _ptr(gr), C.byref(sigma), B, ws, wsn, self._stream()), "cld_sample_step")
        out = {"x_next": xn, "mean": mean, "sigma": float(sigma.value)}
        if guided:
            out["mean_guided"] = mg
            if want_grad:
                out["grad"] = gr
        return out

    def log_prob(self, x_t, x_tm1, cond, t_idx: int):
        x_t = self._f32(x_t)
        M = x_t.shape[0]
        x_t = self._f32(x_t, (M, T, D)); x_tm1 = self._f32(x_tm1, (M, T, D)); cond = self._f32(cond, (M, COND))
        out = torch.empty(M, dtype=torch.float32, device=self.device)
        ws, wsn = self._workspace(M)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_log_prob(self._h, _ptr(x_t), _ptr(x_tm1), _ptr(cond), int(t_idx), _ptr(out), M,
                                              ws, wsn, self._stream()), "cld_log_prob")
        return out

    def lstm_decode(self, z, cond):
        z = self._f32(z)
        B = z.shape[0]
        z = self._f32(z, (B, T, D)); cond = self._f32(cond, (B, COND))
        act = torch.empty(B, T, 2, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_lstm_decode(self._h, _ptr(z), _ptr(cond), _ptr(act), B, self._stream()),
                        "cld_lstm_decode")
        return act

    def action_to_state(self, act, curr_states, scaled_input=True, descaled_output=False):
        act = self._f32(act)
        B = act.shape[0]
        act = self._f32(act, (B, T, 2)); cs = self._f32(curr_states, (B, 4))
        traj = torch.empty(B, T, 6, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_action_to_state(self._h, _ptr(act), _ptr(cs), _ptr(traj), B, int(scaled_input),
                                                     int(descaled_output), self._stream()), "cld_action_to_state")
        return traj

    def decode(self, z, cond, curr_states, descaled_output=True, want_act=False):
        z = self._f32(z)
        B = z.shape[0]
        z = self._f32(z, (B, T, D)); cond = self._f32(cond, (B, COND)); cs = self._f32(curr_states, (B, 4))
        traj = torch.empty(B, T, 6, dtype=torch.float32, device=self.device)
        act = torch.empty(B, T, 2, dtype=torch.float32, device=self.device) if want_act else None
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_decode(self._h, _ptr(z), _ptr(cond), _ptr(cs), _ptr(traj), _ptr(act), B,
                                            int(descaled_output), self._stream()), "cld_decode")
        return (traj, act) if want_act else traj

    def traj2z(self, x6_scaled, cond, noise=None):
        """LSTMVAE.traj2z: -> (z, mu, logvar), each [B,52,4]."""
        x = self._f32(x6_scaled)
        B = x.shape[0]
        x = self._f32(x, (B, T, 6)); cond = self._f32(cond, (B, COND))
        noise = None if noise is None else self._f32(noise, (B, T, D))
        z, mu, lv = (torch.empty(B, T, D, dtype=torch.float32, device=self.device) for _ in range(3))
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_traj2z(self._h, _ptr(x), _ptr(cond), _ptr(noise), _ptr(z), _ptr(mu), _ptr(lv), B,
                                            self._stream()), "cld_traj2z")
        return z, mu, lv

    def vae_loss(self, x6_scaled, act_out, mu, logvar, beta: float):
        """VaeModel.compute_vae_loss (vae_model.py:89-99), forward only -> tensor [3] = (loss, recon, kld)."""
        x = self._f32(x6_scaled)
        B = x.shape[0]
        x = self._f32(x, (B, T, 6)); a = self._f32(act_out, (B, T, 2)); m = self._f32(mu, (B, T, D)); l = self._f32(logvar, (B, T, D))
        out = torch.empty(3, dtype=torch.float32, device=self.device)
        ws, wsn = self._workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_vae_loss(self._h, _ptr(x), _ptr(a), _ptr(m), _ptr(l), C.c_float(beta), _ptr(out), B, ws, wsn,
                                              self._stream()), "cld_vae_loss")
        return out

    def state_to_state_and_action(self, positions, yaws, curr_speed, scaled_output=False):
        p = self._f32(positions)
        B = p.shape[0]
        p = self._f32(p, (B, T, 2)); y = self._f32(yaws, (B, T, 1)); v = self._f32(curr_speed, (B,))
        out = torch.empty(B, T, 6, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_state_to_state_and_action(self._h, _ptr(p), _ptr(y), _ptr(v), _ptr(out), B,
                                                               int(scaled_output), self._stream()),
                        "cld_state_to_state_and_action")
        return out

    def context_encode(self, image, curr_states, want_map_feat=False):
        """ContextEncoder.forward (models/context_utils.py:40-61): image [B,34,224,224] NCHW, curr_states [B,4]
        -> cond_feat [B,256] (and the resnet18 fc output [B,256] when `want_map_feat`)."""
        image = self._f32(image)
        B = image.shape[0]
        image = self._f32(image, (B, 34, 224, 224)); cs = self._f32(curr_states, (B, 4))
        cond = torch.empty(B, COND, dtype=torch.float32, device=self.device)
        mf = torch.empty(B, 256, dtype=torch.float32, device=self.device) if want_map_feat else None
        need = int(self.lib.cld_context_workspace_bytes(self._h, B))
        if self._ctx_ws is None or self._ctx_ws.numel() < need:
            self._ctx_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_context_encode(self._h, _ptr(image), _ptr(cs), _ptr(cond), _ptr(mf), B,
                                                    C.c_void_p(self._ctx_ws.data_ptr()), C.c_size_t(self._ctx_ws.numel()),
                                                    self._stream()), "cld_context_encode")
        return (cond, mf) if want_map_feat else cond

    def non_cond_feat(self, curr_states, cond_fill_value: float = -1.0):
        """Unconditional features of classifier-free guidance (upstream diffuser.py:390-411,459-471): the combine MLP on
        [state features | map features of a raster filled with `cond_fill_value`].  The filled raster is the same for every
        agent, so its map feature is computed once per fill value and broadcast."""
        cs = self._f32(curr_states)
        B = cs.shape[0]
        cs = self._f32(cs, (B, 4))
        key = float(cond_fill_value)
        if key not in self._nc_map_feat:
            img = torch.full((1, 34, 224, 224), key, dtype=torch.float32, device=self.device)
            _, mf = self.context_encode(img, torch.zeros(1, 4, device=self.device), want_map_feat=True)
            self._nc_map_feat[key] = mf
        out = torch.empty(B, COND, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_context_combine(self._h, _ptr(self._nc_map_feat[key]), 1, _ptr(cs), _ptr(out), B, self._stream()),
                        "cld_context_combine")
        return out

    def compute_reward(self, traj, traj_scaled, raster_from_agent, drivable_map, other_pos=None, other_avail=None,
                       collision_thresh: float = 0.8):
        """models/rl/criticmodel.py:7-64 per agent -> (reward, offroad, collision), each [B]."""
        traj = self._f32(traj)
        B = traj.shape[0]
        traj = self._f32(traj, (B, T, 6))
        ts = None if traj_scaled is None else self._f32(traj_scaled, (B, T, 6))
        R = self._f32(raster_from_agent, (B, 3, 3))
        dm = torch.as_tensor(drivable_map).to(self.device).ne(0).to(torch.uint8).contiguous()
        H, W = int(dm.shape[-2]), int(dm.shape[-1])
        S = To = 0
        op = oa = None
        if other_pos is not None and torch.as_tensor(other_pos).numel() > 0:
            op = self._f32(other_pos)
            S, To = int(op.shape[1]), int(op.shape[2])
            oa = torch.as_tensor(other_avail).to(self.device).ne(0).to(torch.uint8).contiguous()
        r, o, c = (torch.empty(B, dtype=torch.float32, device=self.device) for _ in range(3))
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_compute_reward(self._h, _ptr(traj), _ptr(ts), _ptr(R), _ptr(dm), H, W, _ptr(op), _ptr(oa), S, To,
                                                    C.c_float(collision_thresh), _ptr(r), _ptr(o), _ptr(c), B, self._stream()),
                        "cld_compute_reward")
        return r, o, c

    def failure_rate_compute(self, state_action, batch: Mapping):
        """models/rl/criticmodel.py:114-145: {'offroad_failure_rate', 'collision_failure_rate', 'overall_failure_rate'} from the
        per-agent offroad / collision terms of `compute_reward` (an agent fails if it leaves the drivable area at any step,
        respectively comes within 0.8 m of another agent); same batch keys as the reference."""
        _, off, col = self.compute_reward(state_action, None, batch["raster_from_agent"], batch["drivable_map"],
                                          batch.get("all_other_agents_future_positions"),
                                          batch.get("all_other_agents_future_availability"))
        o = float((off < 0).float().mean())
        c = float((col < 0).float().mean())
        return {"offroad_failure_rate": o, "collision_failure_rate": c, "overall_failure_rate": (o + c) / 2.0}

    def world_step(self, traj, centroid, yaw, k: int):
        """env_trajdata.py:452-468 for plan step k -> (world [B,3] = (x, y, h), next curr_states [B,4])."""
        traj = self._f32(traj)
        B = traj.shape[0]
        traj = self._f32(traj, (B, T, 6)); centroid = self._f32(centroid, (B, 2)); yaw = self._f32(yaw, (B,))
        world = torch.empty(B, 3, dtype=torch.float32, device=self.device)
        ncs = torch.empty(B, 4, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.cld_world_step(self._h, _ptr(traj), _ptr(centroid), _ptr(yaw), int(k), _ptr(world),
                                                _ptr(ncs), B, self._stream()), "cld_world_step")
        return world, ncs

    # ------------------------------------------------------------------ measurement
    def profile_enable(self, on: bool = True):
        self._check(self.lib.cld_profile_enable(self._h, int(on)), "cld_profile_enable")

    def profile_read(self):
        """-> (summed ms, launches, algorithmic FLOP) of the dominant conv kernel since profile_enable(True)."""
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self._check(self.lib.cld_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(fl)), "cld_profile_read")
        return ms.value, n.value, fl.value

    def profile_read_executed(self):
        """-> (MFMA FLOP executed by the launches profile_read timed, in the form each took; algorithmic FLOP, executed MFMA
        FLOP and launch count of the most recent U-Net evaluation)."""
        ex, ea, ee, nl = C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        self._check(self.lib.cld_profile_read_executed(self._h, C.byref(ex), C.byref(ea), C.byref(ee), C.byref(nl)),
                    "cld_profile_read_executed")
        return ex.value, ea.value, ee.value, nl.value
