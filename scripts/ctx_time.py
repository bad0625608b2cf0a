"""Time cld_context_encode on resident rasters (dense and structured); optional per-kernel stats via rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = Engine(n_timesteps=10)
eng.load_state_dict(synth.make_unet_weights(0)); eng.load_state_dict(synth.make_context_weights(0)); eng.finalize()
if len(sys.argv) > 2: eng.force_kernel("context", sys.argv[2])      # direct | winograd: the 3x3 / stride-1 convolutions
g = torch.Generator(device="cuda"); g.manual_seed(1)
img = torch.empty(B, 34, 224, 224, device="cuda").uniform_(-1, 1, generator=g)
cs = torch.rand(B, 4, device="cuda", generator=g)
for tag in ("dense", "structured"):
    if tag == "structured":
        img.zero_()
        px = torch.randint(8, 216, (B, 31, 7, 2), device="cuda", generator=g)
        bi = torch.arange(B, device="cuda")[:, None, None].expand(B, 31, 7); pi = torch.arange(31, device="cuda")[None, :, None].expand(B, 31, 7)
        img[bi, pi, px[..., 1], px[..., 0]] = 1.0
        img[:, 31:] = (torch.rand(B, 3, 14, 14, device="cuda", generator=g) > 0.5).float().repeat_interleave(16, 2).repeat_interleave(16, 3)
    eng.context_encode(img, cs); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): eng.context_encode(img, cs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f"{sys.argv[2] if len(sys.argv) > 2 else 'auto'} {tag}: B={B} {dt*1e3:.2f} ms  {B/dt:.0f} agents/s  {B*6.062e9/dt/1e12:.1f} dense-equivalent TFLOP/s")
