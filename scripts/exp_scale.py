import sys, subprocess, os, json
cfgs = [(1024, 50), (4096, 25), (16384, 12)]
variants = [("4", "0")]
for C, nblk in cfgs:
    for s, lpc in variants:
        for wl in ("frontend", "full"):
            env = dict(os.environ, M17GPU_SYNC_IMPL=s, M17GPU_LANES_PER_CHANNEL=lpc)
            r = subprocess.run([sys.executable, "bench.py", "--channels", str(C), "--blocks", str(nblk), "--steps", "6", "--warmup", "2",
                                "--no-cpu-baseline", "--unique", "128", "--workload", wl], env=env, capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"C={C:6d} nblk={nblk:3d} sync={s} lpc={lpc:>2} {wl:8s}: {d['value']:9.0f} Msym/s  {d['roofline']['avg_ms']}  frac {d['roofline']['frac']:.4f}", flush=True)
            except Exception as e:
                print("fail", C, nblk, s, lpc, wl, r.stderr[-300:], flush=True)
