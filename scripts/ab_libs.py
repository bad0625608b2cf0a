"""A/B of library builds in ONE process on one device (interleaved rounds, median and min per build):
    python3 scripts/ab_libs.py <agents> <libA.so> <libB.so> [...]        # U-Net evaluations
    AB_MODE=sample python3 scripts/ab_libs.py ...                         # whole 100-step sampling calls
Each build is loaded under its own path (ctypes keeps them apart); all share torch's HIP runtime."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import _lib, synth
from cld_amd.engine import Engine

B = int(sys.argv[1])
libs = sys.argv[2:]
mode = os.environ.get("AB_MODE", "unet")
prec = os.environ.get("AB_PRECISION", "f32")
engines = []
import ctypes
ALL = dict(_lib.SIGNATURES)
for path in libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    probe = ctypes.CDLL(_lib.LIB_PATH)          # older builds may lack the newest debug entries: bind what the build exports
    _lib.SIGNATURES = {k: v for k, v in ALL.items() if hasattr(probe, k)}
    e = Engine(100, "cuda:0", precision=prec)
    e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
    engines.append(e)
g = torch.Generator(device="cuda"); g.manual_seed(B)
x = torch.randn(B, 52, 4, device="cuda", generator=g); c = torch.randn(B, 256, device="cuda", generator=g)
nc = torch.randn(B, 256, device="cuda", generator=g)
z = torch.randn(100, B, 52, 4, device="cuda", generator=g) if mode != "unet" else None
cfg = float(os.environ.get("AB_CFG", "0"))


def work(e):
    if mode == "unet":
        for _ in range(10):
            e.unet_forward(x, c, 50)
    else:
        e.sample(x, c, noise=z, non_cond=nc if cfg else None, guidance_w=cfg)


for e in engines:
    work(e)
torch.cuda.synchronize()
times = [[] for _ in engines]
for r in range(int(os.environ.get("AB_ROUNDS", "9"))):
    for i, e in enumerate(engines):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); work(e); t.record(); torch.cuda.synchronize()
        times[i].append(s.elapsed_time(t) / (10 if mode == "unet" else 1))
for path, ts in zip(libs, times):
    per = "U-Net eval" if mode == "unet" else "sample call"
    med = statistics.median(ts)
    rate = B * (1 if mode == "unet" else 100) / (med * 1e-3)
    print(f"{os.path.basename(path):28s} B={B} {per}: median {med*1e3:9.1f} us  min {min(ts)*1e3:9.1f} us  ({rate*119232512*(2 if cfg else 1)/1e12:6.1f} TFLOP/s)  rounds {[round(v*1e3) for v in ts]}")
