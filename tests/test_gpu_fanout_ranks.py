"""The C-ABI fan-out entries (include/m17gpu.h: m17gpu_shard_scatter_iq, m17gpu_pack_records,
m17gpu_shard_gather_packed) at world size 2 and 3 on ONE GPU: the ranks are threads of one process, each with its own
context and stream, and the library is bound (m17gpu_shard_set_library) to tests/compat/libm17loop_transport.so, which
matches ncclSend / ncclRecv between them.  What real RCCL needs a multi-GPU node for -- the sizes of every exchange, the
collective verdict of the packed gather, no operation left without a partner after a refusal -- runs here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import ctypes as C, os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.environ["M17_ROOT"])
import m17_sdr_amd as m
from tests import oracle
lib = m.lib()
loop_path = os.path.join(os.environ["M17_ROOT"], "tests", "compat", "libm17loop_transport.so")
assert lib.m17gpu_shard_set_library(loop_path.encode()) == 0
loop = C.CDLL(loop_path)
loop.loop_comm.restype = C.c_void_p
loop.loop_set_limit.argtypes = [C.c_double]
loop.loop_set_limit(15.0)
world, total, nblk = int(os.environ["M17_WORLD"]), int(os.environ["M17_CHANNELS"]), 10
sig = m.generate_batch(total, nblk, n_stream_frames=4)
whole = oracle.Channels(total).rx_blocks(sig["iq"], mode=1, want_syms=False)
want_offs = np.concatenate([[0], np.cumsum(whole["counts"])]).astype(np.int32)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
res = [None] * world
bar = threading.Barrier(world)

def _range(r):
    a, b = C.c_int(), C.c_int()
    lib.m17gpu_shard_range(r, world, total, C.byref(a), C.byref(b))
    return a.value, b.value

def rank_main(rank):
    try:
        lo, hi = C.c_int(), C.c_int()
        lib.m17gpu_shard_range(rank, world, total, C.byref(lo), C.byref(hi))
        lo, hi = lo.value, hi.value
        comm = C.c_void_p(loop.loop_comm(rank, world))
        stream = torch.cuda.Stream()
        st = C.c_void_p(stream.cuda_stream)
        out = {}
        with torch.cuda.stream(stream):
            rx = m.Receiver(max(1, hi - lo), nblk)              # (an empty range still needs a handle: its C is never used)
            if hi - lo == 0:
                rx.C = 0
            full = torch.from_numpy(sig["iq"]).cuda() if rank == 0 else None
            mine = torch.zeros((max(1, hi - lo), nblk, 1920, 2), dtype=torch.int16, device="cuda")
            cap = 2 * nblk + 2
            if hi > lo:
                rc = lib.m17gpu_shard_scatter_iq(rx._ctx, comm, rank, world, 0, p(full), total, nblk, p(mine), st)
                out["scatter"] = (rc, lib.m17gpu_last_error().decode())
                stream.synchronize()
                assert rc == 0 and torch.equal(mine.cpu(), torch.from_numpy(sig["iq"][lo:hi])), out
                o = rx.alloc_outputs(nblk)
                rx.rx_blocks(mine, 1, o)
                packed, offs = rx.pack_records(o)
            pall = torch.zeros((total * cap, 64), dtype=torch.uint8, device="cuda") if rank == 0 else None
            oall = torch.full((total + 1,), -1, dtype=torch.int32, device="cuda") if rank == 0 else None
            totals = (C.c_int32 * world)()
            if hi == lo:
                packed = offs = None                             # a rank without channels: no buffers, capacity 0 -- but it CALLS
            full_cap = int(packed.shape[0]) if packed is not None else 0
            last = max(r for r in range(world) if _range(r)[1] > _range(r)[0])       # the last rank that has channels

            def gather(cap_mine, cap_all, buffers=True):
                return lib.m17gpu_shard_gather_packed(rx._ctx, comm, rank, world, 0, p(packed) if buffers else None, cap_mine,
                                                      p(offs) if buffers else None, total, p(pall), cap_all, p(oall), totals, st)
            # EVERY rank of the communicator makes every call, also one whose channel range is empty (world > channels)
            # 1. the gathering rank's buffer too small: refused on EVERY rank, nothing left in the transport
            bar.wait()
            rc = gather(full_cap, 3)
            out["small_all"] = (rc, lib.m17gpu_last_error().decode())
            bar.wait()
            out["pending_1"] = loop.loop_pending()
            # 2. the last rank's own buffer too small (as if it had packed into 2 rows): refused on every rank
            bar.wait()
            rc = gather(2 if rank == last else full_cap, total * cap)
            out["small_mine"] = (rc, lib.m17gpu_last_error().decode())
            bar.wait()
            out["pending_2"] = loop.loop_pending()
            # 2b. that rank passes NO buffers at all (a rank-local argument error): its "no" travels with the verdicts, every
            # rank returns together -- it used to return alone, in front of the exchange, and leave the gathering rank waiting
            bar.wait()
            rc = gather(0 if rank == last else full_cap, total * cap, buffers=(rank != last))
            out["no_buffers"] = (rc, lib.m17gpu_last_error().decode())
            bar.wait()
            out["pending_2b"] = loop.loop_pending()
            # 3. and the communicator is as good as new: the real gather
            bar.wait()
            rc = gather(full_cap, total * cap)
            out["good"] = (rc, lib.m17gpu_last_error().decode())
            stream.synchronize()
            bar.wait()
            out["pending_3"] = loop.loop_pending()
            if rank == 0 and rc == 0:
                assert np.array_equal(oall.cpu().numpy(), want_offs), (oall.cpu().numpy(), want_offs)
                rows = pall.cpu().numpy()
                for c in range(total):
                    assert rows[want_offs[c]:want_offs[c + 1]].tobytes() == whole["recs"][c, :whole["counts"][c]].tobytes(), c
                spans = [int(whole["counts"][_range(r)[0]:_range(r)[1]].sum()) for r in range(world)]
                assert list(totals) == spans, (list(totals), spans)
        res[rank] = out
    except BaseException as e:                                   # noqa: BLE001 -- reported by the parent
        res[rank] = {"error": repr(e)}
        try:
            bar.abort()
        except Exception:
            pass

threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads: t.start()
for t in threads: t.join()
ERR_ARG = -3
for r, o in enumerate(res):
    assert o is not None and "error" not in o, (r, o)
    assert o["small_all"][0] == ERR_ARG and "refused on every rank" in o["small_all"][1], (r, o)
    assert o["small_mine"][0] == ERR_ARG and "refused on every rank" in o["small_mine"][1], (r, o)
    assert o["no_buffers"][0] == ERR_ARG and "refused on every rank" in o["no_buffers"][1], (r, o)
    assert o["good"][0] == 0, (r, o)
    assert o["pending_1"] == 0 and o["pending_2"] == 0 and o["pending_2b"] == 0 and o["pending_3"] == 0, (r, o)
print("FANOUT_RANKS_OK", world, total)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world,channels", [(2, 7), (3, 8), (4, 3)])          # (4, 3): world > channels, the last rank has none
def test_capi_fanout_protocol_between_ranks_on_one_gpu(tmp_path, world, channels):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, M17_ROOT=ROOT, M17_WORLD=str(world), M17_CHANNELS=str(channels), OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "FANOUT_RANKS_OK" in r.stdout, r.stdout[-4000:]
