import re, subprocess, sys, os
pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "controllable-latent-diffusion-for-traffic-simulation_amd")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-o", "/dev/null",
                      os.path.join(pkg, "csrc", "conv_block.hip"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]+\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    t = re.findall(r"Li(\d+)E", k)
    print(" ".join(f"{x:>3s}" for x in t), {a: v.get(a) for a in ("VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize", "Occupancy")})
