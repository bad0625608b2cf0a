"""Small-batch latency of the sampling call: wall time vs the sum of kernel durations (is it launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
dev = torch.device("cuda:0")
e = Engine(100, dev); e.load_state_dict(synth.make_unet_weights(0)); e.finalize()
for B in [int(a) for a in sys.argv[1:]] or [8, 64, 256]:
    g = torch.Generator(device=dev); g.manual_seed(B)
    x = torch.randn(B, 52, 4, device=dev, generator=g); c = torch.randn(B, 256, device=dev, generator=g)
    z = torch.randn(100, B, 52, 4, device=dev, generator=g)
    e.sample(x, c, noise=z); torch.cuda.synchronize()
    t0 = time.perf_counter(); e.sample(x, c, noise=z); t_launch = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"B={B}: wall {t*1e3:.1f} ms, host enqueue {t_launch*1e3:.1f} ms")
    if hasattr(torch.cuda, "CUDAGraph"):
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            e.sample(x, c, noise=z)
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=s):
                out = e.sample(x, c, noise=z)
        torch.cuda.synchronize()
        gr.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize(); tg = time.perf_counter() - t0
        ref = e.sample(x, c, noise=z)
        print(f"   graph replay {tg*1e3:.1f} ms   identical={torch.equal(out[0], ref[0])}")
