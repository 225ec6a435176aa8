"""PCIe-inclusive rate for information: pinned host IQ -> device copy + m17gpu_rx_blocks, per step."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
C, nblk = 1024, 50
rx = m.Receiver(C, nblk)
dev = rx.gen_batch(nblk)["iq"]
host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True); host.copy_(dev)
out = rx.alloc_outputs(nblk, want_syms=True)
def step():
    dev.copy_(host, non_blocking=True)
    rx.rx_blocks(dev, 0, out)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
gb = dev.numel() * 2 / 1e9
print(f"copy + front end: {dt*1e3:.2f} ms per step of {gb:.3f} GB -> {gb/dt:.1f} GB/s over PCIe, {C*nblk*192/dt/1e6:.0f} Msym/s")
