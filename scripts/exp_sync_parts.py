"""Time the timing loop alone (framer skipped through the stage entry m17gpu_sync_samples, lock forced)
against timing + framer (m17gpu_sync_frame) on the same discriminator stream."""
import sys, os, time, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import m17_sdr_amd as m
from m17_sdr_amd.api import _ptr, _stream, _check
Cn, nblk = 1024, 50
rx = m.Receiver(Cn, nblk)
iq = rx.gen_batch(nblk)["iq"]
disc, offs = rx.frontend(iq)
out = rx.alloc_outputs(nblk, want_syms=True)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
L = m.lib()
ms_full = t(lambda: rx.sync_frame(disc, out))
ms_tim = t(lambda: _check(L.m17gpu_sync_samples(rx._ctx, _ptr(disc), nblk, 1, _ptr(out["syms"]), _ptr(out["nsyms"]), _stream()), "ss"))
ms_tim_nosym = t(lambda: _check(L.m17gpu_sync_samples(rx._ctx, _ptr(disc), nblk, 1, _ptr(out["syms"]), None, _stream()), "ss"))
print(f"sync_impl={os.environ.get('M17GPU_SYNC_IMPL','default')}: timing+framer {ms_full:.4f} ms, timing only (locked threshold) {ms_tim:.4f} ms, same without nsyms {ms_tim_nosym:.4f} ms")
