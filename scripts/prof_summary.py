#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a short text summary."""
import csv, glob, sys, os
d = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    out.append(f"# {os.path.relpath(f, d)}")
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        out.append(", ".join(f"{k}={r[k]}" for k in r))
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        k = r.get("Kernel_Name", "?")
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(k, [0, 0, r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")), r.get("LDS_Block_Size", ""), r.get("Grid_Size", ""), r.get("Workgroup_Size", "")])
        a[0] += 1; a[1] += dur
    out.append(f"# {os.path.relpath(f, d)} (per-kernel aggregate of the trace)")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f"{k[:90]}: calls={a[0]} total_us={a[1]/1e3:.1f} avg_us={a[1]/a[0]/1e3:.2f} vgpr={a[2]} lds={a[3]} grid={a[4]} wg={a[5]}")
print("\n".join(out))
