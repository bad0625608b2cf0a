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
