"""GPU tests of the multi-rank launch path.  This file sorts first on purpose: the 2-rank rehearsal
starts child processes, and it does so before the test process itself has touched the GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks_gloo_rehearsal():
    """`python bench.py --gpus 2` must start two ranks itself (no torchrun) and report n_gpus == 2, with the
    fan-out legs (shard.scatter_iq -> compute -> shard.gather_records) timed apart.  On a 1-GPU box the two
    ranks share the card and the collectives run over gloo (M17_BENCH_BACKEND=gloo); on a multi-GPU node
    the same code path runs one rank per GPU over RCCL."""
    env = dict(os.environ, M17_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--channels", "96", "--blocks", "6", "--no-cpu-baseline", "--no-noisy"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak"
    assert line["config"]["channels_total"] == 192 and line["value"] > 0
    assert line["fanout"]["fanout_ms"] > 0 and line["fanout"]["gather_ms"] > 0
    assert line["roofline"]["calls_timed"] == 4 and line["roofline"]["frac"] > 0
    # the gather ships content, not capacity: the valid rows + 4 bytes per channel
    fan = line["fanout"]
    assert 96 * 4 < fan["records_bytes_per_rank"] < fan["records_bytes_unpacked"] * 6 // 10, fan
    assert line["with_fanout"]["ms_per_step"] > 0 and line["with_fanout_overlapped"]["ms_per_step"] > 0
    # who took part: every rank's device and channel range, the group's own size; what a caller gets beside what the kernels take
    assert [r["rank"] for r in line["ranks"]] == [0, 1] and [r["shard_range"] for r in line["ranks"]] == [[0, 96], [96, 192]]
    assert all(r["pci_bus_id"] for r in line["ranks"]) and line["collectives"]["world_size"] == 2
    assert line["collectives"]["distinct_devices"] == 1                     # the rehearsal: two ranks on this box's one card
    assert 0 < line["roofline"]["frac_wall"] <= line["roofline"]["frac"] * 1.02 and line["roofline"]["path"]["fir"] in (1, 5)


def test_bench_drives_the_capi_fanout_entries():
    """`bench.py --fanout capi`: the legs go through the C-ABI entries north_star names (m17gpu_shard_scatter_iq,
    m17gpu_pack_records, m17gpu_shard_gather_packed) on an ncclComm_t made with RCCL's C API.  One GPU = one rank: the
    communicator, every call and the stream plumbing of the double-buffered leg are exercised; the sends themselves
    need a second GPU (no multi-GPU box has been available in any round)."""
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "M17_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--channels", "128", "--blocks", "6", "--no-cpu-baseline", "--no-noisy", "--no-fir-stage", "--fanout", "capi"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    fan = line["fanout"]
    assert "fanout_error" not in fan, fan
    assert fan["transport"].startswith("C-ABI") and fan["fanout_ms"] > 0 and fan["gather_ms"] > 0
    assert 128 * 4 < fan["records_bytes_per_rank"] < fan["records_bytes_unpacked"] * 6 // 10
    assert line["with_fanout_overlapped"]["ms_per_step"] > 0
    # RCCL's own view of the communicator the legs ran on (degenerate at one rank), next to the rank table
    assert line["collectives"]["ncclCommCount"] == 1 and fan["ncclCommCount"] == 1 and fan["ncclCommCuDevice"] == 0
    assert len(line["ranks"]) == 1 and line["ranks"][0]["shard_range"] == [0, 128] and line["ranks"][0]["pci_bus_id"]
    assert line["roofline"]["frac_wall"] > 0


def test_bench_line_survives_a_failing_fanout_leg():
    """A fan-out leg that throws on ONE rank must not cost the measurement nor strand the other rank: the legs end
    on every rank (all-reduced failure flag) and rank 0 still prints the line, with the error in it."""
    env = dict(os.environ, M17_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4", M17_BENCH_INJECT_FANOUT_FAILURE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--channels", "64", "--blocks", "4", "--no-cpu-baseline", "--no-noisy"],
                       env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "fanout_error" in line["fanout"] and "with_fanout" not in line


def test_bench_line_survives_a_hanging_fanout_leg():
    """A transfer leg that never returns on one rank (its peer then waits in the agreement all-reduce for ever) must
    not cost the measurement either: the watchdog prints the line without the legs and ends every rank -- with an exit
    code of its own (3: neither "clean" nor a crash), which the rank launcher passes on together with the line."""
    env = dict(os.environ, M17_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4", M17_BENCH_INJECT_FANOUT_HANG="1",
               M17_BENCH_FANOUT_TIMEOUT="8")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--channels", "64", "--blocks", "4", "--no-cpu-baseline", "--no-noisy"],
                       env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 3, (r.returncode, r.stdout[-2000:] + r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert "did not finish" in line["fanout"]["fanout_error"] and "with_fanout" not in line


def test_scatter_and_gather_with_device_tensors_over_rccl():
    """shard.scatter_iq / shard.gather_records with CUDA tensors on the nccl (= RCCL) backend.  One device is
    all a test box has, so the group has one rank: the calls, tensor placement and the record layout are the
    ones the N-rank bench uses; the result must equal the receiver's own output and the oracle's."""
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    import m17_sdr_amd as m
    from m17_sdr_amd.shard import gather_records, scatter_iq
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        C, nblk = 50, 8
        sig = m.generate_batch(C, nblk, n_stream_frames=4)
        full = torch.from_numpy(sig["iq"]).cuda()
        mine = scatter_iq(full, C, nblk, src=0)
        assert mine.is_cuda and mine.shape == full.shape
        rx = m.Receiver(C, nblk)
        out = rx.rx_blocks(mine.contiguous(), 1, rx.alloc_outputs(nblk))
        gr, gc = gather_records(out["recs"], out["counts"], dst=0)
        torch.cuda.synchronize()
        assert gr.is_cuda and gc.is_cuda
        ref = oracle.Channels(C).rx_blocks(sig["iq"], mode=1, want_syms=False)
        np.testing.assert_array_equal(gc.cpu().numpy(), ref["counts"])
        recs = gr.cpu().numpy().view(oracle.REC_DTYPE).reshape(C, -1)
        for c in range(C):
            assert recs[c, :ref["counts"][c]].tobytes() == ref["recs"][c, :ref["counts"][c]].tobytes()
        rx.close()
    finally:
        dist.destroy_process_group()


def test_capi_fanout_entries_with_a_one_rank_rccl_communicator():
    """m17gpu_shard_scatter_iq / m17gpu_shard_gather_records (include/m17gpu.h), the C-ABI a C++ host drives the
    8-GPU loop with: an ncclComm_t made here through RCCL's C API (one rank -- one device is all a test box has), the
    two calls on a stream around m17gpu_rx_blocks, result against the oracle.  With one rank the group is empty and
    the shard moves device to device; range arithmetic is checked for 8 ranks against shard.channel_range."""
    import ctypes as C
    import torch
    import m17_sdr_amd as m
    from m17_sdr_amd.shard import channel_range
    lib = m.lib()
    for world, total in ((8, 131072), (8, 1003), (3, 7), (5, 3)):
        for r in range(world):
            lo, hi = C.c_int(), C.c_int()
            lib.m17gpu_shard_range(r, world, total, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == channel_range(r, world, total)
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UID(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UID(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    torch.zeros(1, device="cuda:0")
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        Cn, nblk = 37, 6
        sig = m.generate_batch(Cn, nblk, n_stream_frames=4)
        full = torch.from_numpy(sig["iq"]).cuda()
        mine = torch.empty_like(full)
        rx = m.Receiver(Cn, nblk)
        st = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = lib.m17gpu_shard_scatter_iq(rx._ctx, comm, 0, 1, 0, p(full), Cn, nblk, p(mine), st)
        assert rc == 0, lib.m17gpu_last_error()
        out = rx.rx_blocks(mine, 1, rx.alloc_outputs(nblk))
        allr, allc = torch.zeros_like(out["recs"]), torch.zeros_like(out["counts"])
        rc = lib.m17gpu_shard_gather_records(rx._ctx, comm, 0, 1, 0, p(out["recs"]), p(out["counts"]), out["rec_cap"],
                                             Cn, p(allr), p(allc), st)
        assert rc == 0, lib.m17gpu_last_error()
        torch.cuda.synchronize()
        assert torch.equal(mine, full)
        ref = oracle.Channels(Cn).rx_blocks(sig["iq"], mode=1, want_syms=False)
        np.testing.assert_array_equal(allc.cpu().numpy(), ref["counts"])
        recs = allr.cpu().numpy().view(oracle.REC_DTYPE).reshape(Cn, -1)
        for c in range(Cn):
            assert recs[c, :ref["counts"][c]].tobytes() == ref["recs"][c, :ref["counts"][c]].tobytes()
        # the packed form: m17gpu_pack_records, m17gpu_shard_gather_packed, m17gpu_unpack_records against the oracle
        packed, offs = rx.pack_records(out)
        pall = torch.zeros_like(packed)
        oall = torch.full((Cn + 1,), -1, dtype=torch.int32, device="cuda")
        totals = (C.c_int32 * 1)()
        rc = lib.m17gpu_shard_gather_packed(rx._ctx, comm, 0, 1, 0, p(packed), int(packed.shape[0]), p(offs), Cn, p(pall), int(pall.shape[0]), p(oall), totals, st)
        assert rc == 0, lib.m17gpu_last_error()
        torch.cuda.synchronize()
        want_offs = np.concatenate([[0], np.cumsum(ref["counts"])]).astype(np.int32)
        np.testing.assert_array_equal(oall.cpu().numpy(), want_offs)
        np.testing.assert_array_equal(offs.cpu().numpy(), want_offs)
        assert totals[0] == want_offs[-1] > 0
        rows = pall.cpu().numpy()
        for c in range(Cn):
            assert rows[want_offs[c]:want_offs[c + 1]].tobytes() == ref["recs"][c, :ref["counts"][c]].tobytes()
        ur, uc = rx.unpack_records(pall, oall, out["rec_cap"])
        torch.cuda.synchronize()
        np.testing.assert_array_equal(uc.cpu().numpy(), ref["counts"])
        ur = ur.cpu().numpy().view(oracle.REC_DTYPE).reshape(Cn, -1)
        for c in range(Cn):
            assert ur[c, :ref["counts"][c]].tobytes() == ref["recs"][c, :ref["counts"][c]].tobytes() and not ur[c, ref["counts"][c]:].view(np.uint8).any()
        # a destination too small for the step's records is refused, not overrun
        assert lib.m17gpu_shard_gather_packed(rx._ctx, comm, 0, 1, 0, p(packed), int(packed.shape[0]), p(offs), Cn, p(pall), 3, p(oall), None, st) == m.ERR_ARG
        # ... and so is a source buffer smaller than what m17gpu_pack_records counted (rows beyond its capacity were never written)
        assert lib.m17gpu_shard_gather_packed(rx._ctx, comm, 0, 1, 0, p(packed), 3, p(offs), Cn, p(pall), int(pall.shape[0]), p(oall), None, st) == m.ERR_ARG
        # the refusals left the communicator usable
        assert lib.m17gpu_shard_gather_packed(rx._ctx, comm, 0, 1, 0, p(packed), int(packed.shape[0]), p(offs), Cn, p(pall), int(pall.shape[0]), p(oall), totals, st) == 0
        # a context that does not hold the rank's range is refused
        assert lib.m17gpu_shard_scatter_iq(rx._ctx, comm, 0, 2, 0, p(full), Cn, nblk, p(mine), st) != 0
        rx.close()
    finally:
        rccl.ncclCommDestroy(comm)


def test_two_receivers_do_not_depend_on_the_current_device():
    """Every C-ABI entry selects its context's device itself (and restores the caller's); the Python handle takes
    the stream of ITS device.  With one GPU the check is that a foreign 'current device' request is harmless and
    that create() rejects a device that does not exist."""
    import torch
    import m17_sdr_amd as m
    n = torch.cuda.device_count()
    with pytest.raises(RuntimeError):
        m.Receiver(2, 2, device=n)                         # no such device
    rx = [m.Receiver(8, 4, device=d) for d in range(min(n, 2))] * (2 if n == 1 else 1)
    sig = m.generate_batch(8, 4, n_stream_frames=2)
    ref = oracle.Channels(8).rx_blocks(sig["iq"], mode=1, want_syms=False)
    for k, r in enumerate(rx[:2]):
        torch.cuda.set_device((k + 1) % n)                 # "current" device differs from the receiver's when n > 1
        iq = torch.from_numpy(sig["iq"]).to(f"cuda:{r.device}")
        r.reset()
        out = r.rx_blocks(iq, 1, r.alloc_outputs(4))
        torch.cuda.synchronize(r.device)
        np.testing.assert_array_equal(out["counts"].cpu().numpy(), ref["counts"])
        assert torch.cuda.current_device() == (k + 1) % n
    for r in set(rx):
        r.close()
    torch.cuda.set_device(0)
