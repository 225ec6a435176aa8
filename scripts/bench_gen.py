"""GPU-side signal source: time m17gpu_gen_batch and the host generator on the same job."""
import sys, time, json, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
res = []
for C, nblk, eb, cut in ((1024, 50, 200.0, 0.0), (1024, 50, 8.0, 6250.0), (16384, 12, 8.0, 6250.0)):
    rx = m.Receiver(C, nblk)
    rx.gen_batch(nblk, ebn0_db=eb, noise_cutoff_hz=cut); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): rx.gen_batch(nblk, ebn0_db=eb, noise_cutoff_hz=cut)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    hC = min(C, 256)
    t1 = time.perf_counter(); m.generate_batch(hC, nblk, ebn0_db=eb, noise_cutoff_hz=cut, nthreads=os.cpu_count()); th = (time.perf_counter() - t1) * C / hC
    res.append({"channels": C, "blocks": nblk, "ebn0_db": eb, "noise_cutoff_hz": cut, "gpu_ms": round(dt * 1e3, 2),
                "gpu_Msamples_per_s": round(C * nblk * 1920 / dt / 1e6, 1), "iq_GBps_written": round(C * nblk * 7680 / dt / 1e9, 1),
                "host_generator_ms_all_threads_scaled": round(th * 1e3, 1), "host_threads": os.cpu_count()})
    rx.close()
print(json.dumps(res, indent=1))
