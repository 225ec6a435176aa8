"""Throughput of the wide-band ingest decimator against the HBM roofline.
Algorithmic bytes per output sample: 8 inputs x 4 B read + 4 B written = 36 B."""
import sys, json, torch
sys.path.insert(0, '.')
import m17_sdr_amd as m
C, nblk = 1024, 10
n_in = 15360 * nblk                       # 384 kHz samples for nblk 40-ms blocks
wide = torch.randint(-30000, 30000, (C, n_in, 2), dtype=torch.int16, device="cuda")
rx = m.Receiver(C, 1)
for _ in range(3): out = rx.pluto_decimate(wide)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = rx.pluto_decimate(wide)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
byts = C * (n_in // 8) * 36
print(json.dumps({"kernel": "k_pluto_decimate(+hist)", "channels": C, "in_samples_per_channel": n_in, "ms": round(ms, 4),
                  "achieved_GBps": round(byts / ms / 1e6, 1), "frac_of_8TBps": round(byts / ms / 1e6 / 8000, 4),
                  "out_Msps": round(C * (n_in // 8) / ms / 1e3, 1)}))
