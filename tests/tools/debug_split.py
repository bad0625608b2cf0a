import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from cld_amd import synth
from cld_amd.engine import Engine
from oracle import cld_oracle as O
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
B = 16
wn = synth.make_unet_weights(0, affine_jitter=True)
w = O.to_torch(wn)
e = Engine(100, "cuda:0", precision=prec); e.load_state_dict(wn); e.finalize()
x = torch.from_numpy(synth.normal(1, "dx", (B, 52, 4)))
cond = torch.from_numpy(synth.make_inputs(B, 1)["cond_feat"])
t = 50
e.unet_forward(x, cond, t); torch.cuda.synchronize()
NBUF_OFF, ACT = 3 * 208 + 1792, 3328
def buf(idx, C, L, s22):
    bp = (B + 15) // 16 * 16
    ws = e._ws.view(torch.float32); off = bp * NBUF_OFF + idx * bp * ACT
    raw = ws[off: off + bp * C * L]
    if s22:
        h = raw.view(torch.float16).reshape(bp, L, C // 8, 2, 8).float()
        return (h[:, :, :, 0] + h[:, :, :, 1]).reshape(bp, L, C)[:B].permute(0, 2, 1).cpu()
    return raw.reshape(bp, L, C)[:B].permute(0, 2, 1).cpu()
s22 = prec == "f16x2"
# oracle pieces of block 0
h = x.transpose(1, 2)
te = O.sinusoidal_emb(torch.full((B,), t), 32)
te = F.linear(te, w["model.time_mlp.1.weight"], w["model.time_mlp.1.bias"]); te = F.linear(F.mish(te), w["model.time_mlp.3.weight"], w["model.time_mlp.3.bias"])
tc = torch.cat([te, cond], -1)
p = "model.downs.0.0"
R0 = F.conv1d(h, w[p + ".residual_conv.weight"], w[p + ".residual_conv.bias"])
H0 = O.conv_block(h, w, p + ".blocks.0") + F.linear(F.mish(tc), w[p + ".time_mlp.1.weight"], w[p + ".time_mlp.1.bias"])[:, :, None]
A0 = O.conv_block(H0, w, p + ".blocks.1") + R0
stop = int(os.environ.get("CLD_DEBUG_STOP", "99"))      # honoured by -DCLD_EXPERIMENTS builds only (select one with CLD_LIB_PATH)
print("stop", stop)
print("R0 (b0) err", float((buf(0, 64, 52, s22) - R0).abs().max()))
if stop == 2: print("H0 (b1) err", float((buf(1, 64, 52, s22) - H0).abs().max()))
if stop == 3: print("A0 (b2) err", float((buf(2, 64, 52, s22) - A0).abs().max()), "max|A0|", float(A0.abs().max()))
