"""CPU tests (no GPU): the C-ABI library loads and exports every symbol that
include/m17gpu.h declares; compute entry points refuse to run without a device;
the channel-sharding helper works across 2 processes over gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "m17gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(m17g(?:pu|en)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import m17_sdr_amd as m
    from m17_sdr_amd import _lib
    lib = m.lib()
    names = _declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/m17gpu.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype"
    assert C.sizeof(_lib.Rec) == 64


def test_library_exports_nothing_the_header_does_not_declare():
    """The shipped library's dynamic symbols in the m17gpu_ / m17gen_ name space are exactly the header's: experiments and
    measurement hooks belong in the instrumented build (make stamps), not in libm17gpu.so."""
    import subprocess
    from m17_sdr_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if re.match(r".* [TW] m17g(pu|en)_", ln)})
    declared = _declared_symbols()
    assert exported == declared, sorted(set(exported) ^ set(declared))


def test_header_is_plain_c_and_links_from_a_c_program(tmp_path):
    """The drop-in boundary is a C ABI: include/m17gpu.h must compile as C99 (no C++, no torch, no HIP or RCCL types) and a
    C program must link against libm17gpu.so with nothing but that header -- here one that touches only host-side entries
    (no compute call without a GPU)."""
    hdr = os.path.join(ROOT, "include", "m17gpu.h")
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-x", "c", hdr], check=True)
    src = tmp_path / "host.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "m17gpu.h"
int main(void) {
    int lo = -1, hi = -1;
    m17gpu_shard_range(3, 8, 131072, &lo, &hi);
    uint8_t lsf[30]; memset(lsf, 0, sizeof lsf);
    m17gpu_lsf_fields f;
    if (m17gpu_parse_lsf(lsf, &f) != M17GPU_OK) return 2;
    m17gpu_ctx *ctx = NULL;
    int rc = m17gpu_create(&ctx, 4, 2, 0);          /* no device here: must fail loudly, never fall back */
    printf("%d %d %d %s\n", lo, hi, rc, f.dst_call);
    if (rc == M17GPU_OK) m17gpu_destroy(ctx);
    return 0;
}
""")
    exe = tmp_path / "host"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", os.path.join(ROOT, "m17_sdr_amd"), "-lm17gpu", "-Wl,-rpath," + os.path.join(ROOT, "m17_sdr_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == "49152" and out[1] == "65536"
    import torch
    if not torch.cuda.is_available():
        assert int(out[2]) == -1                       # M17GPU_ERR_NO_DEVICE


def test_integration_host_loop_compiles():
    """The C++ host loop INTEGRATION.md section B.2 describes (scatter, step, pack, packed gather, double-buffered) is kept as
    tests/compat/host_loop_8gpu.cpp and must compile against include/m17gpu.h alone."""
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "compat", "host_loop_8gpu.cpp")], check=True)


def test_compat_shim_exports_reference_signatures():
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "m17_sdr_amd", "libm17compat.so")],
                         capture_output=True, text=True, check=True).stdout
    for mangled in ["_Z18m17_viterbi_decodePfPhi", "_Z19m17_dsp_demap_framePfS_", "_Z17m_17_golay_decodejRt",
                    "_Z10m17_dsp_rxP6scmplxi", "_Z19m17_rx_sync_samplesPfS_i", "_Z18m17_de_correlate_1PfS_i",
                    "_Z17m17_de_interleavePfS_i", "_Z14m17_de_punc_p2PfS_i", "_Z19hard_decode_24_bitsPf",
                    "_Z11pack_1_to_8PhS_i", "_Z20m17_crc_array_encodePhi"]:
        assert mangled in out, mangled


def test_no_cpu_fallback():
    import torch
    import m17_sdr_amd as m
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = m.lib()
    assert lib.m17gpu_device_count() == 0
    ctx = C.c_void_p()
    rc = lib.m17gpu_create(C.byref(ctx), 4, 2, 0)
    assert rc == -1 and b"no HIP device" in lib.m17gpu_last_error()
    with pytest.raises(RuntimeError):
        m.Receiver(4, 2)


def test_channel_range_partition():
    from m17_sdr_amd.shard import channel_range
    for world in (1, 2, 3, 8):
        for c in (1, 7, 1024, 131072):
            spans = [channel_range(r, world, c) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == c
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["M17_ROOT"])
import m17_sdr_amd as m
from m17_sdr_amd.shard import channel_range, gather_packed, gather_records, scatter_iq
from tests import oracle
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
C, nblk = int(os.environ.get("M17_TEST_CHANNELS", "7")), 10
full = None
if rank == 0:
    full = torch.from_numpy(m.generate_batch(C, nblk, n_stream_frames=4)["iq"])
mine = scatter_iq(full, C, nblk, src=0, device=torch.device("cpu"))
lo, hi = channel_range(rank, world, C)
assert mine.shape[0] == hi - lo
# the oracle stands in for the per-GPU receive chain in this CPU test
ref = oracle.Channels(hi - lo).rx_blocks(mine.numpy().copy(), mode=1)
recs = torch.from_numpy(ref["recs"].view(np.uint8).reshape(hi - lo, ref["recs"].shape[1], 64).copy())
counts = torch.from_numpy(ref["counts"].copy())
gr, gc = gather_records(recs, counts, dst=0)
if rank == 0:
    whole = oracle.Channels(C).rx_blocks(full.numpy().copy(), mode=1)
    assert np.array_equal(gc.numpy(), whole["counts"])
    assert gr.numpy().tobytes() == whole["recs"].view(np.uint8).tobytes()
    print("GATHER_OK", int(gc.sum()))
# the packed form (what bench.py and a C++ host gather): valid rows only, channel-major, + exclusive-scan offsets --
# packed here as m17gpu_pack_records packs on the device
offs = torch.from_numpy(np.concatenate([[0], np.cumsum(ref["counts"])]).astype(np.int32))
rows = np.concatenate([ref["recs"][c, :ref["counts"][c]] for c in range(hi - lo)] + [ref["recs"][:0, 0]])
packed = torch.zeros((max(1, (hi - lo) * recs.shape[1]), 64), dtype=torch.uint8)          # capacity, mostly unused
packed[:len(rows)] = torch.from_numpy(rows.view(np.uint8).reshape(-1, 64).copy())
pa, oa, totals = gather_packed(packed, offs, dst=0)
if rank == 0:
    assert totals == [int(whole["counts"][a:b].sum()) for a, b in (channel_range(r, world, C) for r in range(world))]
    want_offs = np.concatenate([[0], np.cumsum(whole["counts"])]).astype(np.int32)
    assert np.array_equal(oa.numpy(), want_offs) and pa.shape[0] == want_offs[-1]
    for c in range(C):
        assert pa[want_offs[c]:want_offs[c + 1]].numpy().tobytes() == whole["recs"][c, :whole["counts"][c]].tobytes()
    print("PACKED_OK", pa.shape[0] * 64, "bytes of records against", gr.numel(), "unpacked")
# a rank whose packed buffer is smaller than its step's records (pack_records wrote no row beyond it): every rank is
# told in the one exchange all of them take part in, every rank raises, none is left in a send or a receive
short = rank == world - 1 and len(rows) > 1
try:
    gather_packed(packed[:1] if short else packed, offs, dst=0)
    refused = False
except ValueError as e:
    refused = "refused on every rank" in str(e)
flags = [None] * world
dist.all_gather_object(flags, (refused, len(rows)))
if flags[world - 1][1] > 1:
    assert all(f[0] for f in flags), flags
    pa2, oa2, _ = gather_packed(packed, offs, dst=0)            # and the group is as good as new
    if rank == 0:
        assert torch.equal(pa2, pa) and torch.equal(oa2, oa)
        print("REFUSAL_OK")
elif rank == 0:
    print("REFUSAL_OK (last rank holds too few records to come up short)")
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,channels,port", [(2, 7, 29517), (3, 5, 29518), (3, 2, 29519)])
def test_scatter_gather_over_gloo(tmp_path, world, channels, port):
    """The N > 1 exchanges on the CPU: scatter of the IQ, gather of the records -- unpacked and packed -- at world sizes 2 and
    3, with uneven channel ranges and (2 channels over 3 ranks) a rank that owns nothing; the gathered whole must equal the
    oracle run on all channels at once."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, M17_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="2", M17_TEST_CHANNELS=str(channels))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0] and "PACKED_OK" in outs[0] and "REFUSAL_OK" in outs[0], outs


def test_net_frame_format():
    """m17_net.cpp:25-74: 54-byte M17-over-IP frame; CRC over the whole frame is zero."""
    import m17_sdr_amd as m
    from tests import oracle
    lib = m.lib()
    lsf = np.arange(30, dtype=np.uint8)
    pl = np.arange(100, 116, dtype=np.uint8)
    out = np.zeros(54, np.uint8)
    assert lib.m17gpu_format_net_frame(0xBEEF, oracle.vp(lsf), 0x1234, oracle.vp(pl), 0, oracle.vp(out)) == 54
    assert bytes(out[:4]) == b"M17 " and out[4] == 0xBE and out[5] == 0xEF
    assert bytes(out[6:34]) == bytes(lsf[:28]) and out[34] == 0x12 and out[35] == 0x34
    assert bytes(out[36:52]) == bytes(pl)
    assert oracle.L().m17o_crc(bytes(out), 54) == 0
    call = lib.m17gen_encode_call(b"M17-M17 A")
    lib.m17gpu_format_net_frame(1, oracle.vp(lsf), 0, oracle.vp(pl), call, oracle.vp(out))
    assert int.from_bytes(bytes(out[6:12]), "big") == call and oracle.L().m17o_crc(bytes(out), 54) == 0


def test_parse_lsf_fields_against_survey_kats():
    """SURVEY 8(c) callsign KATs through m17gpu_parse_lsf, cross-checked with the oracle's decoder."""
    import ctypes as C
    import numpy as np
    from m17_sdr_amd import _lib
    from tests import oracle
    L = _lib.load()

    class Fields(C.Structure):
        _fields_ = [("dst", C.c_uint64), ("src", C.c_uint64), ("dst_call", C.c_char * 10), ("src_call", C.c_char * 10),
                    ("p_s", C.c_uint8), ("dt", C.c_uint8), ("et", C.c_uint8), ("est", C.c_uint8), ("can", C.c_uint8),
                    ("reserved", C.c_uint8), ("meta", C.c_uint8 * 14), ("crc", C.c_uint16), ("crc_ok", C.c_uint8)]

    assert L.m17gen_encode_call(b"G4GUO/P  ") == 0x00102C8DA29F and L.m17gen_encode_call(b"AB1CD    ") == 0x0000009FDD51
    lsf = np.zeros(30, np.uint8)
    meta = np.arange(14, dtype=np.uint8)
    tw = (3 << 7) | (2 << 1) | 1                       # CAN 3, voice, stream
    L.m17gen_build_lsf(0xFFFFFFFFFFFF, 0x00102C8DA29F, tw, meta.ctypes.data_as(C.c_void_p), lsf.ctypes.data_as(C.c_void_p))
    f = Fields()
    assert L.m17gpu_parse_lsf(lsf.ctypes.data_as(C.c_void_p), C.byref(f)) == 0
    assert f.dst == 0xFFFFFFFFFFFF and f.dst_call == b"BROADCAST"
    assert f.src == 0x00102C8DA29F and f.src_call == b"G4GUO/P  "
    assert (f.p_s, f.dt, f.et, f.est, f.can, f.reserved) == (1, 2, 0, 0, 3, 0)
    assert bytes(f.meta) == meta.tobytes() and f.crc_ok == 1
    buf = C.create_string_buffer(10)
    oracle.L().m17o_decode_call(C.c_uint64(0x0000009FDD51), buf)
    lsf[6:12] = [0, 0, 0, 0x9F, 0xDD, 0x51]
    L.m17gpu_parse_lsf(lsf.ctypes.data_as(C.c_void_p), C.byref(f))
    assert f.src_call == buf.value == b"AB1CD    " and f.crc_ok == 0


def lsf_cases(n_random=300, seed=3):
    """LSFs for the field-extraction tests: built ones (SURVEY 8c callsign KATs, broadcast, the largest base-40 word,
    every type-word field exercised) and random bytes (CRC bad, addresses above 40**9)."""
    import ctypes as C
    from m17_sdr_amd import _lib
    from tests import oracle
    L = _lib.load()
    rng = np.random.default_rng(seed)
    out = []
    adds = (0xFFFFFFFFFFFF, 0x00102C8DA29F, 0x0000009FDD51, 0, 40 ** 9 - 1, 1, 39, 40)
    for i, dst in enumerate(adds):
        for src in adds[1:4]:
            b = np.zeros(30, np.uint8)
            tw = int(rng.integers(0, 1 << 16)) if i else ((5 << 7) | (1 << 1) | 1)
            L.m17gen_build_lsf(C.c_uint64(dst), C.c_uint64(src), tw, oracle.vp(rng.integers(0, 256, 14).astype(np.uint8)), oracle.vp(b))
            out.append(b)
    n_built = len(out)
    out += [rng.integers(0, 256, 30).astype(np.uint8) for _ in range(n_random)]
    return np.ascontiguousarray(np.stack(out)), n_built


def test_host_lsf_parser_matches_the_oracle_restatement():
    """m17gpu_parse_lsf (host) against m17o_parse_lsf, the oracle's restatement of parse_lsf (m17_rx_parse.cpp:52-70),
    m17_decode_call (m17_bit_utils.cpp:209-226) and m17_upack_type (:245-254): all 64 bytes of the struct."""
    from m17_sdr_amd import _lib
    from tests import oracle
    L = _lib.load()
    lsf, n_built = lsf_cases()
    got = np.zeros((len(lsf), 64), np.uint8)
    want = np.full((len(lsf), 64), 0xEE, np.uint8)
    for i in range(len(lsf)):
        assert L.m17gpu_parse_lsf(oracle.vp(lsf[i]), oracle.vp(got[i])) == 0
        oracle.L().m17o_parse_lsf(oracle.vp(lsf[i]), oracle.vp(want[i]))
    np.testing.assert_array_equal(got, want)
    assert want[:n_built, 58].all() and not want[n_built:, 58].all()       # crc_ok: built LSFs pass, random ones fail
    assert bytes(want[0, 16:25]) == b"BROADCAST" and bytes(want[0, 26:35]) == b"G4GUO/P  "


def test_traffic_json_comes_from_the_newest_profile_set():
    """bench.py reads roofline.traffic from profiles/traffic.json (PMC passes cannot run inside the bench process): the
    file must have been rebuilt from the newest FETCH_SIZE / WRITE_SIZE passes under profiles/, and name them."""
    import glob
    import json
    import re
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    prefixes = sorted({re.match(r"(r\d\d_[a-z])_pmc_FETCH_SIZE_full\.txt", os.path.basename(f)).group(1)
                       for f in glob.glob(os.path.join(ROOT, "profiles", "r??_?_pmc_FETCH_SIZE_full.txt"))})
    assert prefixes, "no PMC passes under profiles/"
    assert tj.get("_profile_prefix") == prefixes[-1], (tj.get("_profile_prefix"), prefixes[-1])
    assert prefixes[-1] in tj["_comment"]
    for wl in ("full", "frontend"):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            assert os.path.exists(os.path.join(ROOT, "profiles", f"{prefixes[-1]}_pmc_{c}_{wl}.txt"))
    assert tj["full:16384x12"] > 196608 * 7744 and tj["frontend:1024x50"] > 51200 * 8448
    assert tj["full:16384x16"] > 262144 * 7744 and tj["frontend:16384x16"] > 262144 * 8448     # the bench's default step since round 5
