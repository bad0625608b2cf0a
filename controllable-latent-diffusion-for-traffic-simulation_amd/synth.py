"""Deterministic synthetic weights / inputs / noise for the CLD sampling path.

There is no nuScenes data and no trained checkpoint (reference `.gitignore:3`
excludes `*.ckpt`), so every parity test and the benchmark run on random-init
weights and synthetic agents.  The 17.4 MB of U-Net weights are *regenerated*
from a seed on whichever machine needs them instead of being shipped: a
counter-based generator (splitmix64 -> uniform / Box-Muller) in pure NumPy, so
the container that makes the golden fixtures and the GPU box see the same
bits.

Tensor names and shapes follow the reference `state_dict` layout
(`models/dm/dm_model.py:60-66` -> `src/tbsim/models/temporal.py:51-120`,
`models/vae/lstm_vae.py:28-43`); the init distribution mirrors PyTorch's
default (U(+-1/sqrt(fan_in)) for conv/linear weights and biases, GroupNorm
gamma=1, beta=0), optionally with jittered GroupNorm affine parameters so the
affine path is exercised by parity tests.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 counters."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _key(seed: int, name: str) -> np.uint64:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def _u01(seed: int, name: str, n: int, stream: int = 0) -> np.ndarray:
    """n doubles in (0, 1), a pure function of (seed, name, stream, index)."""
    base = _key(seed, name)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(stream)) & _M64
        bits = _splitmix64(_splitmix64(ctr ^ base) ^ base)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def uniform(seed: int, name: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = _u01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed: int, name: str, shape) -> np.ndarray:
    """Standard normal via Box-Muller on two independent counter streams."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = _u01(seed, name, n, stream=0)
    u2 = _u01(seed, name, n, stream=1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------- #
# architecture constants of the reference config (config.yaml:91-172)
# --------------------------------------------------------------------------- #
HORIZON = 52          # config.yaml:107
LATENT = 4            # config.yaml:133 vae.latent_size
COND = 256            # config.yaml:118 cond_feat_dim
BASE_DIM = 32         # config.yaml:106
DIM_MULTS = (2, 4, 8)  # config.yaml:109-112
TIME_DIM = BASE_DIM   # temporal.py:72
HIDDEN = 64           # config.yaml:132 vae.hidden_size
N_TIMESTEPS = 100     # dm_model.py:20 ctor default


def unet_shapes(latent=LATENT, cond=COND, base=BASE_DIM, mults=DIM_MULTS) -> "OrderedDict[str, tuple]":
    """`model.*` state_dict entries of TemporalMapUnet in registration order
    (temporal.py:51-120; block layout temporal.py:18-33, diffuser_helpers.py:50-64)."""
    dims = [latent] + [base * m for m in mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    tc = cond + base
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(p, o, i):
        s[p + ".weight"] = (o, i)
        s[p + ".bias"] = (o,)

    def conv(p, o, i, k):
        s[p + ".weight"] = (o, i, k)
        s[p + ".bias"] = (o,)

    def gn(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def resblock(p, ci, co):
        lin(p + ".time_mlp.1", co, tc)
        for j, cin in ((0, ci), (1, co)):
            conv(f"{p}.blocks.{j}.block.0", co, cin, 5)
            gn(f"{p}.blocks.{j}.block.2", co)
        if ci != co:
            conv(p + ".residual_conv", co, ci, 1)

    lin("model.time_mlp.1", base * 4, base)
    lin("model.time_mlp.3", base, base * 4)
    n_res = len(in_out)
    for ind, (ci, co) in enumerate(in_out):
        resblock(f"model.downs.{ind}.0", ci, co)
        resblock(f"model.downs.{ind}.1", co, co)
        if ind < n_res - 1:
            conv(f"model.downs.{ind}.2.conv", co, co, 3)
    mid = dims[-1]
    resblock("model.mid_block1", mid, mid)
    resblock("model.mid_block2", mid, mid)
    for ind, (ci, co) in enumerate(reversed(in_out[1:])):
        resblock(f"model.ups.{ind}.0", co * 2, ci)
        resblock(f"model.ups.{ind}.1", ci, ci)
        # ConvTranspose1d weight is [C_in, C_out, 4] (diffuser_helpers.py:45)
        conv(f"model.ups.{ind}.2.conv", ci, ci, 4)
    fin = in_out[1][0]
    conv("model.final_conv.0.block.0", fin, fin, 5)
    gn("model.final_conv.0.block.2", fin)
    conv("model.final_conv.1", latent, fin, 1)
    return s


def decoder_shapes(latent=LATENT, hidden=HIDDEN, cond=COND) -> "OrderedDict[str, tuple]":
    """`lstm_dec.*` entries of the LSTM-VAE decoder (lstm_vae.py:28-43)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    g = 4 * hidden
    s["lstm_dec.lstm.weight_ih_l0"] = (g, latent)
    s["lstm_dec.lstm.weight_hh_l0"] = (g, hidden)
    s["lstm_dec.lstm.bias_ih_l0"] = (g,)
    s["lstm_dec.lstm.bias_hh_l0"] = (g,)
    s["lstm_dec.lstm.weight_ih_l1"] = (g, hidden)
    s["lstm_dec.lstm.weight_hh_l1"] = (g, hidden)
    s["lstm_dec.lstm.bias_ih_l1"] = (g,)
    s["lstm_dec.lstm.bias_hh_l1"] = (g,)
    s["lstm_dec.cond2hidden.weight"] = (hidden, cond)
    s["lstm_dec.cond2hidden.bias"] = (hidden,)
    s["lstm_dec.hid2act.weight"] = (2, hidden)
    s["lstm_dec.hid2act.bias"] = (2,)
    return s


def encoder_shapes(hidden=HIDDEN, cond=COND, latent=LATENT) -> "OrderedDict[str, tuple]":
    """`lstm_enc.*`, `mu.*`, `logvar.*` entries of LSTMVAE (lstm_vae.py:6-19,82-83)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    g = 4 * hidden
    s["lstm_enc.lstm.weight_ih_l0"] = (g, 6)
    s["lstm_enc.lstm.weight_hh_l0"] = (g, hidden)
    s["lstm_enc.lstm.bias_ih_l0"] = (g,)
    s["lstm_enc.lstm.bias_hh_l0"] = (g,)
    s["lstm_enc.lstm.weight_ih_l1"] = (g, hidden)
    s["lstm_enc.lstm.weight_hh_l1"] = (g, hidden)
    s["lstm_enc.lstm.bias_ih_l1"] = (g,)
    s["lstm_enc.lstm.bias_hh_l1"] = (g,)
    s["lstm_enc.cond2hidden.weight"] = (hidden, cond)
    s["lstm_enc.cond2hidden.bias"] = (hidden,)
    s["mu.weight"] = (latent, hidden)
    s["mu.bias"] = (latent,)
    s["logvar.weight"] = (latent, hidden)
    s["logvar.bias"] = (latent,)
    return s


def make_encoder_weights(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    shapes = encoder_shapes()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in shapes.items():
        if ".lstm." in name:
            bound = 1.0 / np.sqrt(HIDDEN)
        elif name.endswith(".weight"):
            bound = 1.0 / np.sqrt(shape[1])
        else:
            bound = 1.0 / np.sqrt(shapes[name[: -len("bias")] + "weight"][1])
        out[name] = uniform(seed, name, shape, -bound, bound)
    return out


def _fan_in(name: str, shape: tuple) -> int:
    if name.endswith(".2.conv.weight") and ".ups." in name:
        # ConvTranspose1d: PyTorch computes fan_in from dim 1 of [C_in, C_out, k]
        return shape[1] * shape[2]
    if len(shape) == 3:
        return shape[1] * shape[2]
    return shape[1]


def make_unet_weights(seed: int = 0, affine_jitter: bool = False) -> "OrderedDict[str, np.ndarray]":
    """PyTorch-default-like init of every `model.*` tensor, keyed by state_dict name."""
    shapes = unet_shapes()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in shapes.items():
        is_gn = ".block.2." in name
        if is_gn:
            if name.endswith(".weight"):
                out[name] = uniform(seed, name, shape, 0.5, 1.5) if affine_jitter else np.ones(shape, np.float32)
            else:
                out[name] = uniform(seed, name, shape, -0.3, 0.3) if affine_jitter else np.zeros(shape, np.float32)
            continue
        if name.endswith(".weight"):
            bound = 1.0 / np.sqrt(_fan_in(name, shape))
        else:
            wshape = shapes[name[: -len("bias")] + "weight"]
            bound = 1.0 / np.sqrt(_fan_in(name[: -len("bias")] + "weight", wshape))
        out[name] = uniform(seed, name, shape, -bound, bound)
    return out


def perturb_unet_weights(w, seed: int, rel: float = 0.02, final_abs: float = 0.05) -> "OrderedDict[str, np.ndarray]":
    """Weights "after an optimiser step" (the PPO update of src/trainers/guide_dm_trainer.py:127-183 moves every DM tensor
    between sampling and `log_prob`): every tensor is moved by `rel` of its rms along a seeded normal direction, the output
    layer `model.final_conv.1` by `final_abs` absolute -- enough that the t = 0 posterior mean moves by ~1e-2, far above the
    fp32 rounding of O(1) latents (noise_cof[0] = 0.025 damps whatever the U-Net changes)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, v in w.items():
        rms = float(np.sqrt(np.mean(np.square(v, dtype=np.float64)))) if v.size else 0.0
        step = final_abs if name.startswith("model.final_conv.1.") else rel * rms
        out[name] = (v + step * normal(seed, "perturb:" + name, v.shape)).astype(np.float32)
    return out


def make_decoder_weights(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """nn.LSTM init is U(+-1/sqrt(hidden)) for every tensor; Linear as above."""
    shapes = decoder_shapes()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in shapes.items():
        if ".lstm." in name:
            bound = 1.0 / np.sqrt(HIDDEN)
        elif name.endswith(".weight"):
            bound = 1.0 / np.sqrt(shape[1])
        else:
            bound = 1.0 / np.sqrt(shapes[name[: -len("bias")] + "weight"][1])
        out[name] = uniform(seed, name, shape, -bound, bound)
    return out


def make_inputs(B: int, seed: int = 1) -> dict:
    """Synthetic boundary inputs: cond_feat ~ N(0,1) [B,256],
    curr_states = (0, 0, v~U[0,15], 0) [B,4]."""
    cond = normal(seed, "cond_feat", (B, COND))
    cs = np.zeros((B, 4), np.float32)
    cs[:, 2] = uniform(seed, "curr_speed", (B,), 0.0, 15.0)
    return {"cond_feat": cond, "curr_states": cs}


def make_collision_scene(scene_sizes, seed: int = 1, spacing: float = 3.5) -> dict:
    """Synthetic scene geometry for upstream's AgentCollisionLoss (src/tbsim/utils/guidance_loss.py:442-630): the agents of every
    scene stand on a jittered grid `spacing` metres apart, all heading roughly the same way, so that the faster ones run into
    the ones ahead within the 5.2-s horizon.  -> extent [B,3] (length, width, height), world_from_agent [B,3,3] (rotation by the
    agent's world heading + its world position), scene_index [B] (consecutive blocks), curr_speed [B] (a few below the
    0.5 m/s `guide_moving_speed_th`)."""
    B = int(sum(scene_sizes))
    ext = np.stack([uniform(seed, "col_len", (B,), 3.8, 5.2), uniform(seed, "col_wid", (B,), 1.7, 2.2), np.full((B,), 1.6, np.float32)], axis=1)
    th = uniform(seed, "col_heading", (B,), -0.25, 0.25).astype(np.float64) + 0.6
    pos = np.zeros((B, 2))
    scene_index = np.zeros((B,), np.int64)
    b = 0
    for si, n in enumerate(scene_sizes):
        cols = int(np.ceil(np.sqrt(n)))
        for k in range(n):
            pos[b] = (100.0 * si + spacing * (k % cols), spacing * (k // cols))
            scene_index[b] = si
            b += 1
    pos += uniform(seed, "col_jitter", (B, 2), -0.6, 0.6)
    W = np.zeros((B, 3, 3), np.float64)
    W[:, 0, 0] = np.cos(th); W[:, 0, 1] = -np.sin(th); W[:, 1, 0] = np.sin(th); W[:, 1, 1] = np.cos(th)
    W[:, :2, 2] = pos; W[:, 2, 2] = 1.0
    speed = uniform(seed, "col_speed", (B,), 0.0, 12.0)
    speed[::5] = 0.2                                    # stationary agents: no gradient to them, no loss of their own
    return {"extent": ext.astype(np.float32), "world_from_agent": W.astype(np.float32), "scene_index": scene_index,
            "curr_speed": speed.astype(np.float32)}


def make_collision_trajectories(B: int, N: int, curr_speed, seed: int = 1) -> np.ndarray:
    """[B,N,52,6] descaled (x, y, v, yaw, acc, yaw-rate) plans in the agent frame that drive forward at about curr_speed with
    a little lateral and heading wobble, different per sample."""
    dt = 0.1
    v = np.clip(curr_speed[:, None, None] + np.cumsum(normal(seed, "colt_acc", (B, N, HORIZON)) * 0.4, axis=2), 0.0, 20.0)
    yaw = np.cumsum(normal(seed, "colt_yr", (B, N, HORIZON)) * 0.02, axis=2)
    x = np.cumsum(v * np.cos(yaw) * dt, axis=2)
    y = np.cumsum(v * np.sin(yaw) * dt, axis=2)
    acc = np.gradient(v, dt, axis=2)
    yr = np.gradient(yaw, dt, axis=2)
    return np.stack((x, y, v, yaw, acc, yr), axis=-1).astype(np.float32)


def make_map_scene(B: int, seed: int = 1, half_width_m=(1.6, 3.5)) -> dict:
    """Synthetic inputs of upstream's MapCollisionLoss (src/tbsim/utils/guidance_loss.py:717-875): raster_from_agent [B,3,3]
    (2 px/m, the agent at pixel (56, 112), a small rotation: trajdata_utils.py:380-389), a drivable map [B,224,224] that is a
    straight road band of a per-agent half width along the agent's heading (so that a 2-m-wide agent with some lateral offset
    hangs over the kerb for part of its plan), extents and speeds."""
    th = uniform(seed, "map_rot", (B,), -0.2, 0.2).astype(np.float64)
    R = np.zeros((B, 3, 3), np.float64)
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1] = 2 * np.cos(th), -2 * np.sin(th), 2 * np.sin(th), 2 * np.cos(th)
    R[:, 0, 2], R[:, 1, 2], R[:, 2, 2] = 56.0, 112.0, 1.0
    hw = uniform(seed, "map_halfwidth", (B,), half_width_m[0], half_width_m[1]).astype(np.float64)
    yy, xx = np.meshgrid(np.arange(224), np.arange(224), indexing="ij")
    dmap = np.zeros((B, 224, 224), bool)
    for b in range(B):                       # pixel -> agent frame: p = R^-1 (pix - t); drivable where |p_y| < half width
        Ri = np.linalg.inv(R[b, :2, :2])
        py = Ri[1, 0] * (xx - 56.0) + Ri[1, 1] * (yy - 112.0)
        dmap[b] = np.abs(py) < hw[b]
    ext = np.stack([uniform(seed, "map_len", (B,), 3.8, 5.2), uniform(seed, "map_wid", (B,), 1.7, 2.2), np.full((B,), 1.6, np.float32)], axis=1)
    speed = uniform(seed, "map_speed", (B,), 1.0, 12.0)
    speed[::4] = 0.2
    return {"raster_from_agent": R.astype(np.float32), "drivable_map": dmap, "extent": ext.astype(np.float32), "curr_speed": speed.astype(np.float32)}


def make_map_trajectories(B: int, N: int, curr_speed, seed: int = 1) -> np.ndarray:
    """[B,N,52,6] plans that drive forward with a lateral drift of up to a couple of metres (across the kerb of make_map_scene)."""
    tr = make_collision_trajectories(B, N, curr_speed, seed)
    drift = uniform(seed, "mapt_drift", (B, N, 1), -0.06, 0.06) * np.arange(HORIZON, dtype=np.float32)[None, None, :]
    off = uniform(seed, "mapt_off", (B, N, 1), -1.2, 1.2)
    tr[..., 1] += off + drift
    tr[..., 3] += uniform(seed, "mapt_yaw", (B, N, 1), -0.15, 0.15)
    return tr.astype(np.float32)


def make_future(B: int, seed: int = 1) -> dict:
    """Synthetic ground-truth futures for the encoder path: a unicycle roll-out with smooth random controls in
    the agent frame -> target_positions [B,52,2], target_yaws [B,52,1], curr_speed [B]."""
    dt = 0.1
    v0 = uniform(seed, "fut_speed", (B,), 0.0, 15.0).astype(np.float64)
    acc = np.cumsum(normal(seed, "fut_acc", (B, HORIZON)).astype(np.float64) * 0.3, axis=1)
    yr = np.cumsum(normal(seed, "fut_yr", (B, HORIZON)).astype(np.float64) * 0.05, axis=1)
    yr[0] = 2.5          # one agent spins through +-pi so the yaw wrap of angle_diff is exercised
    v = np.clip(v0[:, None] + np.cumsum(acc * dt, axis=1), 0.0, 30.0)
    yaw = np.cumsum(yr * dt, axis=1)
    yaw = (yaw + np.pi) % (2 * np.pi) - np.pi
    x = np.cumsum(v * np.cos(yaw) * dt, axis=1)
    y = np.cumsum(v * np.sin(yaw) * dt, axis=1)
    return {"target_positions": np.stack((x, y), axis=-1).astype(np.float32),
            "target_yaws": yaw[..., None].astype(np.float32), "curr_speed": v0.astype(np.float32)}


def make_noise(B: int, steps: int, seed: int = 123) -> dict:
    """x_T [B,52,4] and per-step noise [steps,B,52,4]; slab s is consumed by the
    s-th loop iteration (i = steps-1-s), matching the reference RNG order: one
    `randn` for x_T then one `randn_like` per step incl. t=0 (dm_model.py:110,153)."""
    xT = normal(seed, "x_T", (B, HORIZON, LATENT))
    z = normal(seed, "step_noise", (steps, B, HORIZON, LATENT))
    return {"x_T": xT, "noise": z}


# --------------------------------------------------------------------------- #
# ContextEncoder: models/context_utils.py:8-61
# --------------------------------------------------------------------------- #
RASTER_C, RASTER_HW = 34, 224      # 31 history planes + 3 semantic planes (trajdata_utils.py:409-420,536-544); config.yaml:81
STATE_FEAT = 64                    # config.yaml:119 curr_state_feat_dim
MAP_FEAT = 256                     # config.yaml:120 map_feature_dim
CTX = "context_encoder."
RESNET = CTX + "map_encoder.encoder_heads.map_model."     # MapEncoder -> feature extractor -> RasterizedMapEncoder.map_model


def context_shapes() -> "OrderedDict[str, tuple]":
    """state_dict entries of VaeModel.context_encoder: the two MLPs (base_models.py:21-96: `_model` =
    Sequential(Linear, LayerNorm, ReLU, ..., Linear)) and torchvision's resnet18 with the 34-channel conv1 and the
    512 -> 256 fc that RasterizedMapEncoder installs (base_models.py:559-614).  `num_batches_tracked` entries are
    omitted (integers, unused in eval)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def mlp(p, d_in, hidden, d_out):
        i, d = 0, d_in
        for h in hidden:
            s[f"{p}._model.{i}.weight"] = (h, d); s[f"{p}._model.{i}.bias"] = (h,)
            s[f"{p}._model.{i + 1}.weight"] = (h,); s[f"{p}._model.{i + 1}.bias"] = (h,)      # LayerNorm
            i += 3
            d = h
        s[f"{p}._model.{i}.weight"] = (d_out, d); s[f"{p}._model.{i}.bias"] = (d_out,)

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.{k}"] = (c,)

    mlp(CTX + "agent_state_encoder", 4, (STATE_FEAT, STATE_FEAT), STATE_FEAT)
    r = RESNET
    s[r + "conv1.weight"] = (64, RASTER_C, 7, 7)
    bn(r + "bn1", 64)
    cin = 64
    for li, c in enumerate((64, 128, 256, 512), start=1):
        for b in range(2):
            p = f"{r}layer{li}.{b}"
            s[p + ".conv1.weight"] = (c, cin if b == 0 else c, 3, 3)
            bn(p + ".bn1", c)
            s[p + ".conv2.weight"] = (c, c, 3, 3)
            bn(p + ".bn2", c)
            if b == 0 and cin != c:
                s[p + ".downsample.0.weight"] = (c, cin, 1, 1)
                bn(p + ".downsample.1", c)
        cin = c
    s[r + "fc.weight"] = (MAP_FEAT, 512)
    s[r + "fc.bias"] = (MAP_FEAT,)
    n = STATE_FEAT + MAP_FEAT
    mlp(CTX + "process_cond_mlp", n, (n, n, COND, COND), COND)
    return s


def make_context_weights(seed: int = 0, stat_jitter: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Random ContextEncoder weights.  Convolutions: He-style U(+-sqrt(6/fan_in)) (keeps activations O(1) through the
    20 layers); Linear: PyTorch default.  With `stat_jitter` the BatchNorm running statistics / affine and the LayerNorm
    affine are randomised so that eval-mode BN and LN arithmetic is exercised (default init would make BN an identity)."""
    shapes = context_shapes()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[1]
        if leaf == "running_mean":
            out[name] = uniform(seed, name, shape, -0.2, 0.2) if stat_jitter else np.zeros(shape, np.float32)
        elif leaf == "running_var":
            out[name] = uniform(seed, name, shape, 0.5, 1.5) if stat_jitter else np.ones(shape, np.float32)
        elif len(shape) == 4:
            bound = np.sqrt(6.0 / (shape[1] * shape[2] * shape[3]))
            out[name] = uniform(seed, name, shape, -bound, bound)
        elif len(shape) == 2:
            bound = 1.0 / np.sqrt(shape[1])
            out[name] = uniform(seed, name, shape, -bound, bound)
        else:   # 1-D: Linear bias, or BN / LN affine
            wname = name[: -len(leaf)] + "weight"
            if leaf == "bias" and len(shapes[wname]) == 2:
                bound = 1.0 / np.sqrt(shapes[wname][1])
                out[name] = uniform(seed, name, shape, -bound, bound)
            elif leaf == "weight":
                out[name] = uniform(seed, name, shape, 0.7, 1.3) if stat_jitter else np.ones(shape, np.float32)
            else:
                out[name] = uniform(seed, name, shape, -0.2, 0.2) if stat_jitter else np.zeros(shape, np.float32)
    return out


def make_raster(B: int, seed: int = 1, dense: bool = False) -> np.ndarray:
    """Synthetic `image` [B,34,224,224] with the reference raster's structure (trajdata_utils.py:123-156,409-420):
    planes 0..30 are history frames that are zero except a +1 pixel trail (the agent) and a few -1 pixels
    (neighbours); planes 31..33 are semantic layers in [0,1] (here smooth random blobs thresholded).  `dense`
    fills every plane with U(-1,1) instead (worst case for a data-dependent kernel; used by parity tests)."""
    if dense:
        return uniform(seed, "raster_dense", (B, RASTER_C, RASTER_HW, RASTER_HW), -1.0, 1.0)
    img = np.zeros((B, RASTER_C, RASTER_HW, RASTER_HW), np.float32)
    nb = 6
    px = uniform(seed, "raster_px", (B, 31, 1 + nb, 2), 8.0, RASTER_HW - 8.0).astype(np.int64)
    b_idx = np.arange(B)[:, None]
    p_idx = np.arange(31)[None, :]
    img[b_idx, p_idx, px[:, :, 0, 1], px[:, :, 0, 0]] = 1.0
    for k in range(1, 1 + nb):
        img[b_idx, p_idx, px[:, :, k, 1], px[:, :, k, 0]] = -1.0
    coarse = uniform(seed, "raster_sem", (B, 3, 14, 14), 0.0, 1.0)
    sem = np.repeat(np.repeat(coarse, 16, axis=2), 16, axis=3)
    img[:, 31:34] = (sem > 0.5).astype(np.float32)
    return img


def make_reward_inputs(B: int, seed: int = 1, S: int = 5, T_other: int = 52) -> dict:
    """Synthetic PPO-reward inputs (models/rl/criticmodel.py:7-64): a smooth trajectory [B,52,6] in the agent frame (scaled copy
    via the config's mean / std), raster_from_agent = 2 px/m with the ego at (56, 112) and a small per-agent rotation
    (trajdata_utils.py:380-389), a blobby drivable map, and `S` other agents that shadow the trajectory at random offsets so
    that some pairs fall inside the 0.8 m collision radius."""
    fut = make_future(B, seed)
    pos, yaw = fut["target_positions"].astype(np.float64), fut["target_yaws"].astype(np.float64)
    v = np.linalg.norm(np.diff(np.concatenate([np.zeros((B, 1, 2)), pos], axis=1), axis=1), axis=-1) / 0.1
    acc = np.diff(np.concatenate([fut["curr_speed"].astype(np.float64)[:, None], v], axis=1), axis=1) / 0.1
    yr = np.diff(np.concatenate([np.zeros((B, 1)), yaw[..., 0]], axis=1), axis=1) / 0.1
    traj = np.concatenate([pos, v[..., None], yaw, acc[..., None], yr[..., None]], axis=-1).astype(np.float32)
    mean = np.array([13.162, -0.13891, 5.0223, -0.0046415, -0.0080072, -0.0013546], np.float32)
    std = np.array([13.0717, 2.2462, 3.6187, 0.2210, 2.5770, 0.0840], np.float32)
    th = uniform(seed, "reward_rot", (B,), -0.3, 0.3).astype(np.float64)
    R = np.zeros((B, 3, 3), np.float64)
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1] = 2 * np.cos(th), -2 * np.sin(th), 2 * np.sin(th), 2 * np.cos(th)
    R[:, 0, 2], R[:, 1, 2], R[:, 2, 2] = 56.0, 112.0, 1.0
    coarse = uniform(seed, "reward_map", (B, 28, 28), 0.0, 1.0)
    dmap = np.repeat(np.repeat(coarse, 8, axis=1), 8, axis=2) > 0.02
    off = normal(seed, "reward_other_off", (B, S, 1, 2)) * 1.5 + normal(seed, "reward_other_jit", (B, S, T_other, 2)) * 0.5
    other = pos[:, None, :T_other].astype(np.float32) + off
    avail = uniform(seed, "reward_avail", (B, S, T_other), 0.0, 1.0) > 0.3
    return {"traj": traj, "traj_scaled": (traj - mean) / std, "raster_from_agent": R.astype(np.float32), "drivable_map": dmap,
            "other_pos": other.astype(np.float32), "other_avail": avail}
