import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, '.')
import m17_sdr_amd as m
from tests import oracle
Cn, nblk = 130, 12
sig = m.generate_batch(Cn, nblk, n_stream_frames=8, ebn0_db=200.0)
iq = sig["iq"]
rx = m.Receiver(Cn, nblk)
disc, offs = rx.frontend(torch.from_numpy(iq).cuda())
disc = disc.cpu().numpy(); offs = offs.cpu().numpy()
och = oracle.Channels(Cn)
bad = 0
ref_disc = np.zeros_like(disc)
for c in range(Cn):
    for b in range(nblk):
        d, raw, off = oracle.frontend(iq[c, b], och.buf[c])
        ref_disc[c, b] = d
        if not np.array_equal(d.view(np.uint32), disc[c, b].view(np.uint32)):
            bad += 1
            if bad < 4:
                i = np.nonzero(d.view(np.uint32) != disc[c, b].view(np.uint32))[0]
                print("disc mismatch c", c, "b", b, "n", len(i), "first", i[:5], d[i[:3]], disc[c, b][i[:3]], "off", off, offs[c, b])
print("frontend mismatching blocks:", bad)
# sync_frame from the oracle's discriminator stream
rx2 = m.Receiver(Cn, nblk)
out = rx2.alloc_outputs(nblk, want_syms=True)
rx2.sync_frame(torch.from_numpy(ref_disc).cuda(), out)
torch.cuda.synchronize()
och2 = oracle.Channels(Cn)
ref = och2.rx_blocks(iq, mode=0)
ns = out["nsyms"].cpu().numpy()
idx = np.argwhere(ns != ref["nsyms"])
print("nsyms mismatches", len(idx), idx[:10])
for c, b in idx[:3]:
    print("chan", c, "blk", b, "gpu", ns[c], "ref", ref["nsyms"][c])
gs = out["syms"].cpu().numpy(); rs = ref["syms"]
for c in range(Cn):
    if not np.array_equal(gs[c].view(np.uint32), rs[c].view(np.uint32)):
        i = np.nonzero(gs[c].view(np.uint32) != rs[c].view(np.uint32))[0]
        print("sym mismatch chan", c, "first idx", i[:5], "of", len(i), gs[c][i[:4]], rs[c][i[:4]], "cum nsyms", np.cumsum(ref["nsyms"][c])[:12])
        break
