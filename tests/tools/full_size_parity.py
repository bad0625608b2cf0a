"""One-off check at BASELINE configs[1] size (1,024 agents, 100 steps): the HIP chain against the oracle on the same
noise.  The oracle needs ~2-3 minutes of CPU for this, so it is not part of the test suite; the result is quoted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cld_amd import synth
from cld_amd.engine import Engine
from oracle import cld_oracle as O

B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 100
wu, wd = synth.make_unet_weights(0, affine_jitter=True), synth.make_decoder_weights(0)
inp, nz = synth.make_inputs(B, 1), synth.make_noise(B, n, 123)
cond, cs = torch.from_numpy(inp["cond_feat"]), torch.from_numpy(inp["curr_states"])
xT, z = torch.from_numpy(nz["x_T"]), torch.from_numpy(nz["noise"])
for prec in ("f32", "f16x2"):
    e = Engine(n, "cuda:0", precision=prec); e.load_state_dict(wu); e.load_state_dict(wd); e.finalize()
    x0, x1, lp = e.sample(xT, cond, noise=z)
    traj = e.decode(x0, cond, cs, descaled_output=True)
    torch.cuda.synchronize()
    if prec == "f32":
        torch.set_num_threads(int(os.environ.get("CLD_CPU_THREADS", "16")))
        t0 = time.time()
        with torch.no_grad():
            ref = O.sample(O.to_torch(wu), O.schedule(n), xT, z, cond)
            reft = O.decode(O.to_torch(wd), ref["pred_traj"], cond, cs)
        print(f"oracle: {time.time()-t0:.0f} s", flush=True)
    s = float(ref["pred_traj"].abs().max())
    d = float((x0.cpu() - ref["pred_traj"]).abs().max())
    dt = float((traj.cpu() - reft).abs().max())
    print(f"{prec}: B={B} steps={n} max|x0|={s:.3e} max|dx0|={d:.3e} rel={d/s:.2e}  max|dtraj|={dt:.3e} (max|traj|={float(reft.abs().max()):.3e})  "
          f"logp dev={float((lp.cpu()-ref['log_prob_final']).abs().max()):.2e}", flush=True)
