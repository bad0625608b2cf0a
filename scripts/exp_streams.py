"""Experiment: does running two independent half-batches on two HIP streams (two workgroups per CU in
different phases) raise MFMA utilisation?  usage: python scripts/exp_streams.py B_total"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine

Btot = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = 3
dev = torch.device("cuda:0")
wu = synth.make_unet_weights(0)

def mk(B):
    e = Engine(100, dev); e.load_state_dict(wu); e.finalize()
    g = torch.Generator(device=dev); g.manual_seed(B)
    return e, torch.randn(B, 52, 4, device=dev, generator=g), torch.randn(B, 256, device=dev, generator=g), \
        torch.randn(100, B, 52, 4, device=dev, generator=g)

def run_single(B):
    e, x, c, z = mk(B)
    e.sample(x, c, noise=z); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): e.sample(x, c, noise=z)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def run_dual(B, nstream=2, floors=None):
    parts = [mk(B // nstream) for _ in range(nstream)]
    if floors:      # asymmetric LDS requests: two workgroups of the same stream cannot share a CU, one of each can
        for (e, *_), f in zip(parts, floors):
            e.lib.cld_debug_lds_floor(e._h, f)
    streams = [torch.cuda.Stream(dev) for _ in range(nstream)]
    for (e, x, c, z), s in zip(parts, streams):
        with torch.cuda.stream(s): e.sample(x, c, noise=z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for (e, x, c, z), s in zip(parts, streams):
            with torch.cuda.stream(s): e.sample(x, c, noise=z)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

print("lib", os.environ.get("CLD_LIB_PATH", "default"))
for B in (Btot // 2, Btot):
    t = run_single(B)
    print(f"single  B={B}: {t*1e3:.1f} ms  {B*100/t:,.0f} step.agent/s")
for ns in (2, 4):
    t = run_dual(Btot, ns)
    print(f"{ns} streams x B={Btot//ns}: {t*1e3:.1f} ms  {Btot*100/t:,.0f} step.agent/s")
for fl in ((100 * 1024, 60 * 1024), (84 * 1024, 76 * 1024), (82 * 1024, 82 * 1024)):
    t = run_dual(Btot, 2, fl)
    print(f"2 streams x B={Btot//2}, LDS floors {fl}: {t*1e3:.1f} ms  {Btot*100/t:,.0f} step.agent/s")
