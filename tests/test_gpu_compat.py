"""GPU test of the reference-signature shim (libm17compat.so): the C++ functions
m17_rx_parse.cpp calls are driven through their mangled names and must return what
the oracle returns."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    """The caller-side harness (tests/compat/rx_frame_harness.cpp) defines m17_rx_symbols / m17_rx_lock,
    which libm17compat.so references weakly -- as the reference's m17_rx_frame.cpp would.  It must be in
    the global symbol scope BEFORE the shim is loaded so the shim binds to it, hence RTLD_GLOBAL and a
    module fixture every test of this file depends on."""
    import torch
    assert torch.cuda.is_available()
    import m17_sdr_amd as m
    m.lib()
    torch.zeros(1, device="cuda")
    H = C.CDLL(os.path.join(ROOT, "tests", "compat", "libm17compat_harness.so"), mode=C.RTLD_GLOBAL)
    H.harness_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return H


def test_block_entry_m17_dsp_rx_feeds_the_callers_framer(harness):
    """_Z10m17_dsp_rxP6scmplxi, the entry m17_tx_rx.cpp:164-165 calls once per 40 ms block: 24 blocks
    through front end + timing recovery on the GPU, symbols handed to the caller's m17_rx_symbols, the
    loop threshold taken from the caller's m17_rx_lock() (scripted: locked for blocks 6..17).  The oracle
    runs the same stages with the same lock flag (m17_dsp.cpp:461-476, m17_rx_sync.cpp:92-95)."""
    import m17_sdr_amd as m
    nblk, lock_from, lock_until = 24, 6, 18
    iq = m.generate_channel(0x4D313777, nblk, n_stream_frames=16, delay=777, ebn0_db=14.0)[0]
    got = np.zeros(nblk * 193 + 8, np.float32)
    counts = np.zeros(nblk, np.int32)
    n = harness.harness_run(oracle.vp(iq), nblk, lock_from, lock_until, oracle.vp(got), got.size, oracle.vp(counts), 1)
    assert n > 0
    ch = oracle.Channels(1)
    flock = ch.buf.view(np.int32)[:, 41]                 # m_flock of the oracle's channel state (tests/oracle.py)
    want, wn = [], []
    for b in range(nblk):
        d, _, _ = oracle.frontend(iq[b], ch.buf[0])
        flock[0] = 1 if lock_from <= b < lock_until else 0
        out = np.zeros(200, np.float32)
        k = oracle.L().m17o_rx_sync_samples(oracle.vp(ch.buf[0]), oracle.vp(d), oracle.vp(out[4:]), 384)
        wn.append(k)
        want.append(out[4:4 + k].copy())
    np.testing.assert_array_equal(counts, wn)
    np.testing.assert_array_equal(got[:n].view(np.uint32), np.concatenate(want).view(np.uint32))
    assert n == sum(wn)          # (a block yields 183..193 symbols: a channel on the 39/0 branch boundary slips many times per block)
    # a block of another length is refused without aborting the process (the reference has no error path)
    dsp_rx = getattr(C.CDLL(os.path.join(ROOT, "m17_sdr_amd", "libm17compat.so")), "_Z10m17_dsp_rxP6scmplxi")
    dsp_rx(oracle.vp(iq), 960)


def test_shim_functions_match_oracle(harness):
    import m17_sdr_amd as m
    S = C.CDLL(os.path.join(ROOT, "m17_sdr_amd", "libm17compat.so"))
    for init in ("_Z12m17_crc_initv", "_Z13m17_init_convv", "_Z21m17_init_de_correlatev", "_Z12m17_dsp_initv",
                 "_Z14m17_golay_initv", "_Z16m17_rx_sync_initv"):
        getattr(S, init)()
    rng = np.random.default_rng(5)
    vit = getattr(S, "_Z18m17_viterbi_decodePfPhi")
    for length in (488, 296, 420):
        soft = rng.normal(0, 1, length).astype(np.float32)
        soft[::9] = 0
        out = np.zeros(length // 2, np.uint8)
        assert vit(oracle.vp(soft), oracle.vp(out), length) == length // 2
        np.testing.assert_array_equal(out, oracle.viterbi(soft))
    dem = getattr(S, "_Z19m17_dsp_demap_framePfS_")
    sym = (rng.normal(0, 1, 192) * 0.04).astype(np.float32)
    sb = np.zeros(368, np.float32)
    dem(oracle.vp(sym), oracle.vp(sb))
    np.testing.assert_array_equal(sb.view(np.uint32), oracle.demap(sym).view(np.uint32))
    gol = getattr(S, "_Z17m_17_golay_decodejRt")
    od, oo = C.c_uint16(), C.c_uint16()
    for _ in range(50):
        w = int(rng.integers(0, 1 << 24))
        assert gol(C.c_uint32(w), C.byref(od)) == oracle.L().m17o_golay_decode(w, C.byref(oo))
        assert od.value == oo.value
    # timing recovery alone, lock flag from the (absent) external framer = unlocked
    sig = m.generate_channel(0x4D313701, 4, n_stream_frames=2)[0]
    ch = oracle.Channels(1)
    syn = getattr(S, "_Z19m17_rx_sync_samplesPfS_i")
    for b in range(4):
        d, _, _ = oracle.frontend(sig[b], ch.buf[0])
        want = np.zeros(400, np.float32)
        n_want = oracle.L().m17o_rx_sync_samples(oracle.vp(ch.buf[0]), oracle.vp(d), oracle.vp(want), 384)
        got = np.zeros(400, np.float32)
        n_got = syn(oracle.vp(d), oracle.vp(got), 384)
        assert n_got == n_want
        np.testing.assert_array_equal(got[:n_got].view(np.uint32), want[:n_want].view(np.uint32))
    crc = getattr(S, "_Z20m17_crc_array_encodePhi")
    crc.restype = C.c_uint16
    assert crc(b"123456789", 9) == 0x772B
