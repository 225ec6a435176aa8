"""Prints the judged numbers of a bench line:  python scripts/show_bench.py FILE.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("step %.4f ms  value %.1f %s  frac %.4f  %s" % (d["ms_per_step"], d["value"], d["unit"], d["roofline"]["frac"], d["roofline"]["avg_ms"]))
if d.get("noisy"): print("noisy", {k: d["noisy"][k] for k in d["noisy"] if k != "workload"})
if d.get("fir_stage"): print("fir_stage frac %.4f  %s" % (d["fir_stage"]["frac"], d["fir_stage"]["avg_ms"]))
if d.get("fir_stage_16384"): print("fir_stage_16384 frac %.4f  %s" % (d["fir_stage_16384"]["frac"], d["fir_stage_16384"]["avg_ms"]))
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"])
for k in ("two_contexts", "step_12_blocks", "fir_stage_16384x12", "fir_stage_16384x48"):
    if d.get(k): print(k, {q: d[k][q] for q in ("ms", "value", "frac", "frac_wall") if q in d[k]})
print("frac_wall", d["roofline"].get("frac_wall"), "path", d["roofline"].get("path"), "ranks", d.get("ranks"), "collectives", {k: v for k, v in (d.get("collectives") or {}).items() if k != "note"})
