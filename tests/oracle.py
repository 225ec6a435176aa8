"""ctypes wrapper of the CPU oracle (oracle/libm17oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "libm17oracle.so")


class Rec(C.Structure):
    _fields_ = [("type", C.c_uint8), ("votes", C.c_uint8), ("golay_errs", C.c_uint8),
                ("frame_errors", C.c_uint8), ("flags", C.c_uint16), ("fn", C.c_uint16),
                ("variance", C.c_float), ("block", C.c_uint32), ("sym_pos", C.c_uint16),
                ("rsv0", C.c_uint16), ("data", C.c_uint8 * 32), ("rsv", C.c_uint8 * 12)]


REC_DTYPE = np.dtype([("type", "u1"), ("votes", "u1"), ("golay_errs", "u1"), ("frame_errors", "u1"),
                      ("flags", "<u2"), ("fn", "<u2"), ("variance", "<f4"), ("block", "<u4"),
                      ("sym_pos", "<u2"), ("rsv0", "<u2"), ("data", "u1", (32,)), ("rsv", "u1", (12,))])
assert REC_DTYPE.itemsize == 64

_L = None


def L():
    global _L
    if _L is None:
        lib = C.CDLL(PATH)
        lib.m17o_init()
        lib.m17o_crc.restype = C.c_uint16
        lib.m17o_golay_encode.restype = C.c_uint32
        lib.m17o_encode_call.restype = C.c_uint64
        for n in ("m17o_tab_mf", "m17o_tab_md"):
            getattr(lib, n).restype = C.POINTER(C.c_float)
        for n in ("m17o_tab_golay_enc", "m17o_tab_golay_err", "m17o_tab_crc"):
            getattr(lib, n).restype = C.POINTER(C.c_uint16)
        lib.m17o_tab_derand.restype = C.POINTER(C.c_uint8)
        _L = lib
    return _L


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def chan_size():
    return L().m17o_sizeof_chan()


class Channels:
    """C oracle channel states."""

    def __init__(self, n):
        self.n = n
        self.size = chan_size()
        self.buf = np.zeros((n, self.size), np.uint8)
        for c in range(n):
            L().m17o_chan_reset(vp(self.buf[c]))

    def rx_blocks(self, iq, mode=1, cap=None, want_syms=True, nthreads=8, net=False, stream_ids=None, dst_override=0):
        """iq int16 [C, nblk, 1920, 2] -> dict(recs[C,cap], counts[C], syms, nsyms[, net[C,cap,56]]).
        net=True attaches the network sink of decode_stream_frame (m17_net_new_rx_data, m17_net.cpp:53-74)."""
        Cn, nblk = iq.shape[0], iq.shape[1]
        assert Cn == self.n and iq.dtype == np.int16 and iq.flags.c_contiguous
        cap = cap or (2 * nblk + 2)
        recs = np.zeros((Cn, cap), REC_DTYPE)
        counts = np.zeros((Cn,), np.int32)
        syms = np.zeros((Cn, nblk * 193 + 8), np.float32) if want_syms else None
        nsyms = np.zeros((Cn, nblk), np.int32) if want_syms else None
        netbuf = np.zeros((Cn, cap, 56), np.uint8) if net else None
        sids = np.ascontiguousarray(stream_ids, np.uint16) if stream_ids is not None else None
        L().m17o_rx_blocks_net(vp(self.buf), Cn, nblk, vp(iq), vp(recs), cap, vp(counts),
                               vp(syms) if want_syms else None, vp(nsyms) if want_syms else None,
                               int(mode), int(nthreads), vp(netbuf) if net else None,
                               vp(sids) if sids is not None else None, C.c_uint64(int(dst_override)))
        return {"recs": recs, "counts": counts, "syms": syms, "nsyms": nsyms, "net": netbuf}

    def set_afc(self, on=True):
        for c in range(self.n):
            L().m17o_set_afc(vp(self.buf[c]), int(bool(on)))

    def afc(self):
        """(m_afc_delta [n] float32, NCO phase [n] float64)"""
        d = np.zeros(self.n, np.float32); a = np.zeros(self.n, np.float64)
        for c in range(self.n):
            dd, aa = C.c_float(), C.c_double()
            L().m17o_get_afc(vp(self.buf[c]), C.byref(dd), C.byref(aa))
            d[c], a[c] = dd.value, aa.value
        return d, a

    def field(self, name):
        """Selected state fields as arrays (layout of m17o_chan in oracle/m17_oracle.h)."""
        i32 = self.buf.view(np.int32)
        f32 = self.buf.view(np.float32)
        off = {"disc_count": 0, "z": 1, "m_clk": 5, "m_thr": 6, "m_index": 7, "sum": 8, "dif": 9,
               "m_buff": 10, "m_flock": 41, "m_fclk": 42, "m_frame_errors": 43, "m_sync": 44, "m_f_sym": 52}
        if name == "z":
            return f32[:, 1:5]
        if name in ("sum", "dif"):
            return f32[:, off[name]]
        if name == "m_buff":
            return f32[:, 10:41]
        if name == "m_sync":
            return f32[:, 44:52]
        if name == "m_f_sym":
            return f32[:, 52:244]
        if name == "m_lsf":
            return self.buf[:, 244 * 4: 244 * 4 + 60].reshape(self.n, 2, 30)
        if name == "counters":       # g_errors, n_frames, in_frame, frame_id_epoch
            base = 244 * 4 + 60 + 4 + 800 + 4
            return self.buf[:, base: base + 16].view(np.uint32)
        return i32[:, off[name]]


def frontend(iq_block, state=None):
    """One channel-block: int16 [1920,2] -> (d[384], d_raw[384], offset)."""
    st = state if state is not None else Channels(1).buf[0]
    d = np.zeros(384, np.float32)
    raw = np.zeros(384, np.float32)
    off = C.c_float()
    L().m17o_frontend(vp(st), vp(np.ascontiguousarray(iq_block)), vp(d), vp(raw), C.byref(off))
    return d, raw, off.value


def viterbi(soft):
    soft = np.ascontiguousarray(soft, np.float32)
    out = np.zeros(len(soft) // 2, np.uint8)
    L().m17o_viterbi_decode(vp(soft), vp(out), len(soft))
    return out


def demap(sym):
    sym = np.ascontiguousarray(sym, np.float32)
    out = np.zeros(368, np.float32)
    L().m17o_demap_frame(vp(sym), vp(out))
    return out


# ---- transmit side (oracle/m17_oracle.c "Transmit side"): the checker of the product's signal source ----
def tx_build_lsf(dst, src, type_word, meta=None):
    meta = np.zeros(14, np.uint8) if meta is None else np.ascontiguousarray(meta, np.uint8)
    lsf = np.zeros(30, np.uint8)
    assert L().m17o_build_lsf(C.c_uint64(int(dst)), C.c_uint64(int(src)), C.c_uint16(int(type_word)), vp(meta), vp(lsf)) == 30
    return lsf


def tx_lsf_frame(lsf, reference_quirks=0):
    d = np.zeros(192, np.uint8)
    assert L().m17o_lsf_frame_dibits(vp(np.ascontiguousarray(lsf, np.uint8)), vp(d), int(reference_quirks)) == 192
    return d


def tx_stream_frame(lsf, lich_count, fn, payload):
    d = np.zeros(192, np.uint8)
    assert L().m17o_stream_frame_dibits(vp(np.ascontiguousarray(lsf, np.uint8)), int(lich_count), C.c_uint16(int(fn)),
                                        vp(np.ascontiguousarray(payload, np.uint8)), vp(d)) == 192
    return d


def tx_packet_frame(payload, length, eof, nf, reference_quirks=0):
    d = np.zeros(192, np.uint8)
    buf = np.zeros(32, np.uint8)
    buf[:len(payload)] = payload
    n = L().m17o_packet_frame_dibits(vp(buf), int(length), int(eof), int(nf), vp(d), int(reference_quirks))
    return d if n == 192 else None


def tx_preamble():
    d = np.zeros(192, np.uint8)
    assert L().m17o_preamble_dibits(vp(d)) == 192
    return d


def tx_eot():
    d = np.zeros(192, np.uint8)
    assert L().m17o_eot_dibits(vp(d)) == 192
    return d


class Modulator:
    """m17_mod_init + m17_mod_dibits / m17_mod_carrier (m17_modulate.cpp:22-92) at 10 samples per symbol."""

    def __init__(self):
        self.buf = np.zeros(L().m17o_sizeof_mod(), np.uint8)
        L().m17o_mod_init(vp(self.buf))

    def modulate(self, dibits, stages=False):
        """dibits uint8 (0..3, 255 = carrier) -> iq int16 [10 n, 2] (, sums float32 [10 n], phases float32 [10 n])"""
        dibits = np.ascontiguousarray(dibits, np.uint8)
        n = len(dibits)
        iq = np.zeros((10 * n, 2), np.int16)
        sums = np.zeros(10 * n, np.float32) if stages else None
        ph = np.zeros(10 * n, np.float32) if stages else None
        assert L().m17o_modulate(vp(self.buf), vp(dibits), n, vp(iq), vp(sums) if stages else None, vp(ph) if stages else None) == 10 * n
        return (iq, sums, ph) if stages else iq


def tx_stream_schedule(lsf, payloads, n_stream_frames, nslots):
    """The dibit schedule of one channel as the reference's transmit thread sends a voice stream
    (m17_tx_rx.cpp:95-98: carrier, two preambles, link setup; then stream frames; m17_send_eot), repeated:
    uint8 [nslots, 192].  payloads [>= frames sent][16]; the LICH counter and FN restart with every transmission
    (m17_send_link_setup_frame, m17_tx_routines.cpp:273-275)."""
    out = np.zeros((nslots, 192), np.uint8)
    period = 5 + n_stream_frames
    sent = 0
    for g in range(nslots):
        slot = g % period
        if slot == 0:
            out[g] = 255
        elif slot <= 2:
            out[g] = tx_preamble()
        elif slot == 3:
            out[g] = tx_lsf_frame(lsf)
        elif slot == 4 + n_stream_frames:
            out[g] = tx_eot()
        else:
            f = slot - 4
            out[g] = tx_stream_frame(lsf, f % 6, f, payloads[sent % len(payloads)])
            sent += 1
    return out
