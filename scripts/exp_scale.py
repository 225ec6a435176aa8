import sys, subprocess, os, json
cfgs = [(1024, 50), (4096, 25), (16384, 12)]
for C, nblk in cfgs:
    for s in (1, 2, 3):
        for fe in ((1, 2) if C * nblk < 300000 else (1, 2)):
            env = dict(os.environ, M17GPU_SYNC_IMPL=str(s), M17GPU_FE_IMPL=str(fe))
            r = subprocess.run([sys.executable, "bench.py", "--channels", str(C), "--blocks", str(nblk), "--steps", "6", "--warmup", "2",
                                "--no-cpu-baseline", "--unique", "128"], env=env, capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"C={C:6d} nblk={nblk:3d} sync={s} fe={fe}: {d['value']:9.0f} Msym/s  {d['roofline']['avg_ms']}  frac {d['roofline']['frac']:.4f}", flush=True)
            except Exception as e:
                print("fail", C, nblk, s, fe, r.stderr[-300:], flush=True)
