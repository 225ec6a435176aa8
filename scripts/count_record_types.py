import sys, os, torch, numpy as np
sys.path.insert(0, '/root/repo')
import m17_sdr_amd as m
C, nblk = int(sys.argv[1]), int(sys.argv[2])
rx = m.Receiver(C, nblk)
iq = rx.gen_batch(nblk * 3, n_stream_frames=40, ebn0_db=200.0)["iq"]
for k in range(3):
    out = rx.rx_blocks(iq[:, k*nblk:(k+1)*nblk].contiguous(), 1, rx.alloc_outputs(nblk))
    torch.cuda.synchronize()
    recs = out["recs"].cpu().numpy(); counts = out["counts"].cpu().numpy()
    typ = recs[:, :, 0]; flags = recs[:, :, 4].astype(np.uint16) | (recs[:, :, 5].astype(np.uint16) << 8)
    valid = np.arange(recs.shape[1])[None, :] < counts[:, None]
    parsed = valid & ((flags & 2) != 0)
    print("call", k, "records", int(valid.sum()), "parsed by type", {t: int((parsed & (typ == t)).sum()) for t in range(6)},
          "channels with a parsed packet frame", int((parsed & (typ == 3)).any(axis=1).sum()))
