"""Numerics study (CPU): what would an f16x2-split conv path cost in accuracy?
Every conv input activation and weight is replaced by hi + lo fp16 planes (22-bit mantissa), products
hi*hi + hi*lo + lo*hi (lo*lo dropped), fp32 accumulation; activations are ALSO stored at 22 bits between
layers (as the HIP path would).  Compares against the golden vectors recorded from the reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from cld_amd import synth
from oracle import cld_oracle as O
from tests.conftest import load_golden

WSCALE = 64.0
def split(t, scale=1.0):
    t = t * scale
    hi = t.half().float()
    lo = (t - hi).half().float()
    return hi, lo

_orig_conv1d, _orig_convT = F.conv1d, F.conv_transpose1d
FIRST = {"n": 0}
def conv1d_split(x, w, b=None, stride=1, padding=0):
    if x.shape[1] == 4:          # the 4-channel latent stays on the exact fp32 path
        return _orig_conv1d(x, w, b, stride=stride, padding=padding)
    xh, xl = split(x); wh, wl = split(w, WSCALE)
    y = (_orig_conv1d(xh, wh, None, stride=stride, padding=padding) + _orig_conv1d(xh, wl, None, stride=stride, padding=padding)
         + _orig_conv1d(xl, wh, None, stride=stride, padding=padding)) / WSCALE
    return y if b is None else y + b[None, :, None]
def convT_split(x, w, b=None, stride=1, padding=0):
    xh, xl = split(x); wh, wl = split(w, WSCALE)
    y = (_orig_convT(xh, wh, None, stride=stride, padding=padding) + _orig_convT(xh, wl, None, stride=stride, padding=padding)
         + _orig_convT(xl, wh, None, stride=stride, padding=padding)) / WSCALE
    return y if b is None else y + b[None, :, None]

torch.set_num_threads(8)
for jitter, tag in ((False, "default"), (True, "jitter")):
    w = O.to_torch(synth.make_unet_weights(0, affine_jitter=jitter))
    meta, g = load_golden(f"unet_forward_{tag}")
    B = meta["B"]
    x = torch.from_numpy(synth.normal(meta["in_seed"], "unet_x", (B, 52, 4))) * 3.0
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    t = torch.tensor(meta["t"], dtype=torch.long)
    e32 = O.unet_forward(w, x, cond, t)
    F.conv1d, F.conv_transpose1d = conv1d_split, convT_split
    es = O.unet_forward(w, x, cond, t)
    F.conv1d, F.conv_transpose1d = _orig_conv1d, _orig_convT
    print(f"single forward ({tag}): fp32-oracle vs golden {np.abs(e32.numpy()-g['eps']).max():.2e} | split vs golden {np.abs(es.numpy()-g['eps']).max():.2e} (bar 2e-5 in the GPU tests, 1e-4 in SURVEY)")

for n, jitter in ((10, True), (100, False), (100, True)):
    tag = f"sample_n{n}_{'jitter' if jitter else 'default'}"
    meta, g = load_golden(tag)
    B = meta["B"]
    w = O.to_torch(synth.make_unet_weights(0, affine_jitter=jitter))
    s = O.schedule(n)
    cond = torch.from_numpy(synth.make_inputs(B, meta["in_seed"])["cond_feat"])
    nz = synth.make_noise(B, n, meta["noise_seed"])
    F.conv1d, F.conv_transpose1d = conv1d_split, convT_split
    out = O.sample(w, s, torch.from_numpy(nz["x_T"]), torch.from_numpy(nz["noise"]), cond)
    F.conv1d, F.conv_transpose1d = _orig_conv1d, _orig_convT
    sc = float(np.abs(g["pred_traj"]).max()); err = float(np.abs(out["pred_traj"].numpy() - g["pred_traj"]).max())
    print(f"{tag}: split-chain max|dx0| = {err:.3e} at max|x0| = {sc:.3e} -> rel {err/sc:.2e} (bar 1e-3)")
