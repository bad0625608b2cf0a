"""Throughput of the sampling call (no decode) over batch sizes -> for the DESIGN.md scaling table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
dev = torch.device("cuda:0")
prec = os.environ.get("CLD_SWEEP_PRECISION", "f32")
e = Engine(100, dev, precision=prec); e.load_state_dict(synth.make_unet_weights(0)); e.finalize()
form = os.environ.get("CLD_SWEEP_CONV5")          # direct | winograd: force the form of the k5 layers at L = 13 / 26 (default: by rows)
if form: e.force_kernel("conv5", form)
print("precision", prec, "conv5", form or "auto")
for B in [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024, 2048, 4096, 8192]:
    g = torch.Generator(device=dev); g.manual_seed(B)
    x = torch.randn(B, 52, 4, device=dev, generator=g); c = torch.randn(B, 256, device=dev, generator=g)
    z = torch.randn(100, B, 52, 4, device=dev, generator=g)
    e.sample(x, c, noise=z); torch.cuda.synchronize()
    reps = 3 if B <= 2048 else 2
    t0 = time.perf_counter()
    for _ in range(reps): e.sample(x, c, noise=z)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    v = B * 100 / t
    print(f"B={B:6d}: {t*1e3:8.1f} ms/sample  {v:12,.0f} step.agent/s  {v*119232512/1e12:6.1f} TFLOP/s ({v*119232512/1e12/157.3*100:4.1f}% of fp32-MFMA peak)", flush=True)
    del x, c, z
