"""GPU check of cld_context_encode against the oracle (debug aid; the real tests live in tests/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
from oracle import cld_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = synth.make_context_weights(0)
eng = Engine(n_timesteps=10)
eng.load_state_dict(synth.make_unet_weights(0))
eng.load_state_dict(w)
eng.finalize()
img = torch.from_numpy(synth.make_raster(B, 1, dense=True))
cs = torch.from_numpy(synth.make_inputs(B, 1)["curr_states"])
cond, mf = eng.context_encode(img.cuda(), cs.cuda(), want_map_feat=True)
torch.cuda.synchronize()
taps = {}
torch.set_num_threads(8)
ref = O.context_encode(O.to_torch(w), img, cs, taps)
e1 = (mf.cpu() - taps["map_feat"]).abs().max().item()
e2 = (cond.cpu() - ref).abs().max().item()
print(f"B={B} map_feat max|d|={e1:.3e} (max|ref|={taps['map_feat'].abs().max():.3e})  cond max|d|={e2:.3e} (max|ref|={ref.abs().max():.3e})")
if B >= 64:
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.context_encode(img.cuda(), cs.cuda())
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"time {dt*1e3:.2f} ms  {B/dt:.1f} agents/s  {B*6.07e9/dt/1e12:.1f} TFLOP/s")
