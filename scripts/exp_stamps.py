"""Diagnostic: in-kernel cycle stamps of one conv launch (needs the -DCLD_STAMPS build via CLD_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from cld_amd import synth
from cld_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
layers = [int(a) for a in sys.argv[2:]] or [14]
dev = torch.device("cuda:0")
e = Engine(100, dev); e.load_state_dict(synth.make_unet_weights(0)); e.finalize()
x = torch.randn(B, 52, 4, device=dev); c = torch.randn(B, 256, device=dev)
buf = torch.zeros(16 * 4096, dtype=torch.int64, device=dev)
for layer in layers:
    for rep in range(3):
        buf.zero_()
        e._check(e.lib.cld_debug_stamps(e._h, C.c_void_p(buf.data_ptr()), layer), "stamps")
        e.unet_forward(x, c, 50); torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(-1, 16).astype(np.int64)
    s = s[s[:, 0] != 0]
    t0 = s[:, 0].min()
    names = ["entry->loop", "K loop", "final barrier", "O-tile write+sync", "epi math", "stores"]
    d = np.diff(s[:, :7], axis=1)
    clk = (s[:, 6] - s[:, 0]) / np.maximum((s[:, 9] - s[:, 8]), 1) * 100.0   # MHz
    print(f"layer {layer}: {len(s)} workgroups; entry skew max {s[:,0].max()-t0} cyc; kernel span {s[:,6].max()-t0} cyc; clock ~{np.median(clk):.0f} MHz")
    for k, nme in enumerate(names):
        print(f"   {nme:18s} mean {d[:,k].mean():9.0f}  min {d[:,k].min():8d}  max {d[:,k].max():8d}")
    ch = np.diff(np.concatenate([s[:, 1:2], s[:, 10:13]], axis=1), axis=1)
    print("   first chunks (cycles):", ch.mean(axis=0).round())
