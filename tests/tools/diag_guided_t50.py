"""Diagnostic (GPU box): where does cld_guidance_step's gradient leave the oracle's at t = 50 of the configs[2] chain?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
from oracle import cld_oracle as O

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
n, B = 100, 2048
torch.set_num_threads(16)
e = Engine(n, "cuda:0", precision=prec)
e.load_state_dict(synth.make_unet_weights(0, affine_jitter=True)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
w, wd = O.to_torch(synth.make_unet_weights(0, affine_jitter=True)), O.to_torch(synth.make_decoder_weights(0))
inp, nz = synth.make_inputs(B, 7), synth.make_noise(B, n, 9)
cond, cs = torch.from_numpy(inp["cond_feat"]).cuda(), torch.from_numpy(inp["curr_states"]).cuda()
tgt = torch.from_numpy(synth.uniform(7, "tgt", (B, 52), 0.0, 12.0)).cuda()
non_cond = torch.from_numpy(synth.normal(7, "non_cond_feat", (B, 256))).cuda()
x, z = torch.from_numpy(nz["x_T"]).cuda(), torch.from_numpy(nz["noise"]).cuda()
gd = {"curr_states": cs, "target_speed": tgt, "lr": 0.3, "optimizer": "adam"}
for it in range(n):
    i = n - 1 - it
    if i in (50, 1):
        got = e.sample_step(x, cond, i, z=z[it], non_cond=non_cond, guidance_w=2.0, guidance=gd, want_grad=True)
        mean = got["mean"].cpu()
        for form in ("auto", "mfma", "valu"):
            e.force_kernel("guide", form)
            _, gr = e.guidance_step(mean, cond, gd, sigma=0.0, want_grad=True)
            e.force_kernel("guide", "auto")
            _, g32 = O.guidance_step(wd, mean, cond.cpu(), cs.cpu(), tgt.cpu(), None, 0.3, None, "adam")
            wd64 = {k: v.double() for k, v in wd.items()}
            _, g64 = O.guidance_step(wd64, mean.double(), cond.cpu().double(), cs.cpu().double(), tgt.cpu().double(), None, 0.3, None, "adam")
            err = (gr.cpu().double() - g64).abs().amax(dim=(1, 2))
            e32 = (g32.double() - g64).abs().amax(dim=(1, 2))
            top = torch.topk(err, 5)
            print(f"t={i} form={form}: max|g64|={float(g64.abs().max()):.3e}  gpu-vs-f64 max {float(err.max()):.3e}  f32oracle-vs-f64 max {float(e32.max()):.3e}; agents over 1e-7: {int((err > 1e-7).sum())}")
            print("   worst agents", top.indices.tolist(), [f"{v:.2e}" for v in top.values.tolist()], " their f32-oracle err", [f"{float(e32[k]):.2e}" for k in top.indices])
        b = int(top.indices[0])
        tr64 = O.decode(wd64, mean[b:b+1].double(), cond.cpu()[b:b+1].double(), cs.cpu()[b:b+1].double(), True)[0]
        trg = e.decode(mean[b:b+1], cond[b:b+1], cs[b:b+1], descaled_output=True)[0].cpu()
        print("   worst agent", b, "v (f64):", [f"{v:.4f}" for v in tr64[:, 2].tolist()])
        print("   acc (f64):", [f"{v:.4f}" for v in tr64[:, 4].tolist()])
        print("   yawrate (f64):", [f"{v:.4f}" for v in tr64[:, 5].tolist()])
        print("   traj gpu-f64 max diff per channel", (trg.double() - tr64).abs().amax(dim=0).tolist())
        print("   |mean| max of agent", float(mean[b].abs().max()))
    else:
        got = e.sample_step(x, cond, i, z=z[it], non_cond=non_cond, guidance_w=2.0, guidance=gd)
    x = got["x_next"]
