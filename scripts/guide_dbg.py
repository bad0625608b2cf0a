"""Diagnostic: compare the guidance kernel's formulations on a library built with -DCLD_QDEBUG=1 (grad_out then carries the forward
actions and dL/daction instead of dL/dz):  CLD_LIB=libcld_hip_qdbg.so python3 scripts/guide_dbg.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
e = Engine(10, "cuda:0"); e.load_state_dict(synth.make_unet_weights(0)); e.load_state_dict(synth.make_decoder_weights(0)); e.finalize()
B = 19
g = torch.Generator(device="cuda"); g.manual_seed(B)
mean = torch.randn(B, 52, 4, device="cuda", generator=g); cond = torch.randn(B, 256, device="cuda", generator=g)
cs = torch.zeros(B, 4, device="cuda"); cs[:, 2] = torch.rand(B, device="cuda", generator=g) * 15
gd = {"curr_states": cs, "target_speed": torch.rand(B, 52, device="cuda", generator=g) * 12, "lr": 0.3, "optimizer": "sgd"}
outs = {}
for name, form in (("quad", 4), ("mfma8", 2), ("valu", 1)):
    e._check(e.lib.cld_debug_force_kernel(e._h, 0, form), "force")
    outs[name] = e.guidance_step(mean, cond, gd, 0.0, want_grad=True)[1].cpu()
a, b = outs["quad"], outs["mfma8"]
print("max |quad - mfma8| per channel:", (a - b).abs().amax(dim=(0, 1)))
print("per agent:", (a - b).abs().amax(dim=(1, 2)))
d = (a - b).abs().reshape(B, 208); print("first half (layer-1 gate grads) max diff", d[:, :104].max(), "second half (layer-0)", d[:, 104:].max(), "scale", b.abs().max())
d = (a - b).abs().reshape(B, 208); print("down max diff per unit", d[:, :64].amax(0)); print("rec1 max diff per unit", d[:, 64:128].amax(0)); print("per agent down", d[:, :64].amax(1))
print("quad down[0]", a.reshape(B,208)[0,:16]); print("mfma8 down[0]", b.reshape(B,208)[0,:16])
print("quad[0,:3]", a[0, :3]); print("mfma8[0,:3]", b[0, :3])
print("valu vs mfma8:", (outs["valu"] - b).abs().max())
