import sys, os, subprocess
code = """
import sys, torch, numpy as np
sys.path.insert(0, '.')
import m17_sdr_amd as m
Cn, nblk = 1024, 50
sig = m.generate_batch(128, nblk, n_stream_frames=40)
iq = torch.from_numpy(sig['iq']).cuda().repeat(8,1,1,1).contiguous()
rx = m.Receiver(Cn, nblk); out = rx.alloc_outputs(nblk, want_syms=True)
for _ in range(2): rx.rx_blocks(iq, 0, out)
torch.cuda.synchronize(); rx.set_profiling(True)
for _ in range(5): rx.rx_blocks(iq, 0, out)
torch.cuda.synchronize(); ms, n = rx.kernel_ms(); print('%.1f us' % (ms[0]*1e3))
"""
for impl, name in [("2","quad full"),("101","no sum phase"),("102","no limiter"),("103","no sum, no limiter"),("104","no loads"),("107","nothing but cvt+disc"),("1","lane-per-cb")]:
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, M17GPU_FE_IMPL=impl), capture_output=True, text=True)
    print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
