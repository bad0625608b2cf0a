"""Time one guidance step (LSTM forward + BPTT + roll-out backward, cld_guidance_step) per kernel formulation:
    python3 scripts/guide_time.py 2048 [1024 4096 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import _lib, synth
from cld_amd.engine import Engine
e = Engine(10, "cuda:0"); e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
for B in [int(a) for a in sys.argv[1:]] or [2048]:
    g = torch.Generator(device="cuda"); g.manual_seed(B)
    mean = torch.randn(B, 52, 4, device="cuda", generator=g); cond = torch.randn(B, 256, device="cuda", generator=g)
    cs = torch.zeros(B, 4, device="cuda"); cs[:, 2] = torch.rand(B, device="cuda", generator=g) * 15
    z = torch.randn(B, 52, 4, device="cuda", generator=g)
    gd = {"curr_states": cs, "target_speed": torch.rand(B, 52, device="cuda", generator=g) * 12, "lr": 0.3, "optimizer": "adam"}
    outs = {}
    for name, form in (("quad  (8 agents, 4x4x1)", 3), ("mfma (16 agents, 16x16x4)", 2), ("valu  (2 agents)", 1)):
        e._check(e.lib.cld_debug_force_kernel(e._h, 0, form), "force")
        for _ in range(60 if not outs else 3):      # the first formulation timed also warms the clocks up
            out = e.guidance_step(mean, cond, gd, 0.5, z=z, want_grad=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                e.guidance_step(mean, cond, gd, 0.5, z=z)
            t.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(t) / 10)
        outs[name] = out
        print(f"B={B} {name:26s}: median {sorted(ts)[3]*1e3:8.1f} us  min {min(ts)*1e3:8.1f} us (incl. ~3 small torch allocations per call)  samples " + " ".join(f"{x*1e3:.0f}" for x in ts))
    ga = outs["mfma (16 agents, 16x16x4)"][2]
    print("   max |grad quad - grad mfma| =", float((outs["quad  (8 agents, 4x4x1)"][2] - ga).abs().max()), " max |grad valu - grad mfma| =", float((outs["valu  (2 agents)"][2] - ga).abs().max()), " max |grad| =", float(ga.abs().max()))
