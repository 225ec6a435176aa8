#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE summaries scripts/collect_profiles.sh leaves in a directory:
   python scripts/traffic_from_pmc.py gpurun_out/prof r02_b
HBM-side bytes per m17gpu_rx_blocks launch = sum over the chain's kernels of (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024
(FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); the signal generator's kernels are left out."""
import json, os, sys
src, tag = sys.argv[1], sys.argv[2]
def pm(path):
    out, k = {}, None
    for line in open(path):
        line = line.rstrip()
        if "dispatches" in line: k = line.split(" dispatches")[0].replace("void ", "").strip()
        elif "mean/dispatch" in line: out[k] = float(line.split()[-1])
    return out
res = {"_comment": ("HBM-side bytes per m17gpu_rx_blocks launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                    "scripts/collect_profiles.sh), KB x 1024, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                    "coalesced reads); raw counters: profiles/%s_pmc_*.txt; built by scripts/traffic_from_pmc.py. Includes the streams "
                    "one kernel writes for the next to read, or a wave for itself (discriminator rows 1.5 KB per channel-block; frame slots "
                    "1.6 KB per stream frame in the full chain).") % tag, "_profile_prefix": tag, "_per_kernel_MB": {}}
for wl, key in (("full", "full:16384x16"), ("full_12", "full:16384x12"), ("frontend", "frontend:1024x50"), ("frontend_16384x16", "frontend:16384x16"),
                ("frontend_16384x12", "frontend:16384x12"), ("frontend_16384x48", "frontend:16384x48"), ("full_noisy", "full-noisy:16384x16")):
    if not os.path.exists(os.path.join(src, f"pmc_FETCH_SIZE_{wl}.txt")):
        continue
    f = pm(os.path.join(src, f"pmc_FETCH_SIZE_{wl}.txt")); w = pm(os.path.join(src, f"pmc_WRITE_SIZE_{wl}.txt"))
    tot = 0.0
    for k in f:
        if "k_gen" in k or "k_reset" in k: continue
        b = (2 * f[k] + w.get(k, 0.0)) * 1024
        res["_per_kernel_MB"][f"{key} {k.replace('m17dev::', '')}"] = round(b / 1e6, 1)
        tot += b
    res[key] = int(tot)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
