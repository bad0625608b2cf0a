"""Maximum-size check: one U-Net evaluation + one DDPM step + decode at very large agent counts; rows of the big batch must
equal the same rows evaluated in a small batch (no 32-bit offset overflow anywhere: one activation is B x 52 x 256 x 4 bytes
= 3.5 GB at B = 65,536)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
dev = torch.device("cuda:0")
e = Engine(100, dev); e.load_state_dict(synth.make_unet_weights(0, affine_jitter=True)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
for B in [int(a) for a in sys.argv[1:]] or [16384, 65536]:
    g = torch.Generator(device=dev); g.manual_seed(B)
    x = torch.randn(B, 52, 4, device=dev, generator=g); c = torch.randn(B, 256, device=dev, generator=g)
    z = torch.randn(B, 52, 4, device=dev, generator=g)
    cs = torch.zeros(B, 4, device=dev); cs[:, 2] = 5.0
    eps = e.unet_forward(x, c, 40)
    xn, mean, _ = e.ddpm_step(x, c, 40, z)
    traj = e.decode(x, c, cs, descaled_output=True)
    idx = torch.tensor([0, 1, B // 2 - 1, B // 2, B - 4097, B - 2, B - 1], device=dev)
    idx = torch.cat([idx, torch.randint(0, B, (4089,), device=dev, generator=g)])      # 4,096 rows -> same 64-column tiling
    eps_s = e.unet_forward(x[idx].contiguous(), c[idx].contiguous(), 40)
    xn_s, _, _ = e.ddpm_step(x[idx].contiguous(), c[idx].contiguous(), 40, z[idx].contiguous())
    tr_s = e.decode(x[idx].contiguous(), c[idx].contiguous(), cs[idx].contiguous(), descaled_output=True)
    torch.cuda.synchronize()
    print(f"B={B}: eps max diff {float((eps[idx]-eps_s).abs().max()):.3e}  x' {float((xn[idx]-xn_s).abs().max()):.3e}  traj {float((traj[idx]-tr_s).abs().max()):.3e}  "
          f"finite {bool(torch.isfinite(eps).all())}  workspace {int(e.lib.cld_workspace_bytes(e._h, B))/2**30:.1f} GiB", flush=True)
    del x, c, z, eps, xn, mean, traj
    torch.cuda.empty_cache()

# full 100-step chain at 65,536 agents with explicit noise (a 5.4 GB [100, B, 52, 4] tensor: offsets past 2^32 bytes)
B = 65536
g = torch.Generator(device=dev); g.manual_seed(7)
xT = torch.randn(B, 52, 4, device=dev, generator=g); c = torch.randn(B, 256, device=dev, generator=g)
z = torch.randn(100, B, 52, 4, device=dev, generator=g)
x0, x1, lp = e.sample(xT, c, noise=z)
idx = torch.cat([torch.tensor([0, B // 2, B - 1], device=dev), torch.randint(0, B, (4093,), device=dev, generator=g)])
x0s, x1s, lps = e.sample(xT[idx].contiguous(), c[idx].contiguous(), noise=z[:, idx].contiguous())
torch.cuda.synchronize()
print("chain: x0 rows equal", bool(torch.equal(x0[idx], x0s)), "x1", bool(torch.equal(x1[idx], x1s)), "logp", bool(torch.equal(lp[idx], lps)),
      "max|x0|", float(x0.abs().max()), "max diff", float((x0[idx] - x0s).abs().max()))
