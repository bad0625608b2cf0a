"""Diagnostic: in-kernel cycle stamps of the Winograd head chain (chain_wino.hip; needs the -DCLD_STAMPS build via CLD_LIB_PATH).
    CLD_LIB_PATH=.../libcld_stamps.so python3 scripts/chainw_stamps.py 4096"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from cld_amd import synth
from cld_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
e = Engine(100, dev); e.load_state_dict(synth.make_unet_weights(0)); e.finalize()
e.force_kernel("unet", "chainw")      # the four-agent tile the phase table was first read on; the library itself takes two-agent tiles at this size
x = torch.randn(B, 52, 4, device=dev); c = torch.randn(B, 256, device=dev)
buf = torch.zeros(16 * 4096, dtype=torch.int64, device=dev)
for rep in range(3):
    buf.zero_()
    e._check(e.lib.cld_debug_stamps(e._h, C.c_void_p(buf.data_ptr()), 0), "stamps")
    e.unet_forward(x, c, 50); torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(-1, 16).astype(np.int64)
s = s[s[:, 0] != 0]
t0 = s[:, 0].min()
names = ["entry + latent conv + its epilogue", "layer 1: transform + 2 x 256 MFMAs", "layer 1: epilogue", "layer 2: transform + MFMAs", "layer 2: epilogue",
         "layer 3: transform + MFMAs", "layer 3: epilogue", "image + k3s2 loop", "stores"]
d = np.diff(s[:, :10], axis=1)
clk = (s[:, 9] - s[:, 0]) / np.maximum((s[:, 15] - s[:, 14]), 1) * 100.0   # MHz
print(f"Winograd head chain: {len(s)} workgroups; kernel span {s[:,9].max()-t0} cyc; workgroup life mean {(s[:,9]-s[:,0]).mean():.0f}; clock ~{np.median(clk):.0f} MHz")
print(f"   MFMA issue per wave: {80 + 3 * 512 + 336} x 32 = {(80 + 3 * 512 + 336) * 32} cycles")
for k, nme in enumerate(names):
    print(f"   {nme:36s} mean {d[:,k].mean():9.0f}  min {d[:,k].min():8d}  max {d[:,k].max():8d}")
for lo_, hi_, tag in ((0, 256, "ids 0..255 (first slot of a CU)"), (256, 512, "ids 256..511 (second slot)"), (512, 1024, "ids 512.. (second generation)")):
    if len(s) >= hi_:
        dd = d[lo_:hi_]
        print(f"   {tag}: life {(s[lo_:hi_, 9] - s[lo_:hi_, 0]).mean():.0f}; phases " + " ".join(f"{dd[:, k].mean():.0f}" for k in range(9)))
st = np.sort(s[:, 0] - t0)
print("   workgroup start times (cycles), every 64th:", st[::64].tolist())
