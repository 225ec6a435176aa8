"""Host-side handle around the C-ABI: owns an m17gpu context, passes torch
tensors' device pointers and the current HIP stream through.  Mirrors the
reference call structure: Receiver.rx_blocks == batched m17_dsp_rx
(m17_dsp.cpp:461-476), Receiver.viterbi_decode == m17_viterbi_decode
(m17_conv.cpp:148-168), and so on.  Errors from the library raise RuntimeError;
nothing here falls back to a CPU implementation."""
import ctypes as C
import numpy as np

from . import _lib


def lib():
    return _lib.load()


def _check(rc, what):
    if rc != 0:
        msg = lib().m17gpu_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Receiver:
    """C independent 48 kHz M17 channels resident on one GPU."""

    def __init__(self, n_channels, max_blocks, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("m17_sdr_amd.Receiver needs a HIP device (no CPU fallback)")
        self.C, self.max_blocks, self.device = int(n_channels), int(max_blocks), int(device)
        self.rec_cap_max = 2 * self.max_blocks + 2
        # torch's HIP context on that device must exist first; the process-wide current device is not touched
        # (the library selects the context's device inside every call and restores the caller's)
        torch.zeros(1, device=f"cuda:{self.device}")
        self._ctx = C.c_void_p()
        _check(lib().m17gpu_create(C.byref(self._ctx), self.C, self.max_blocks, self.device), "m17gpu_create")

    def _stream(self):
        """torch's current stream ON THIS RECEIVER'S DEVICE (not of whatever device is current)."""
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t, dtype, shape=None, name="tensor"):
        """Device / dtype / layout checks of a tensor handed to the C-ABI (a wrong one is an
        out-of-bounds device access, not an exception)."""
        import torch
        if t is None:
            return
        # exceptions, not asserts: `python -O` must not strip the only guard in front of the device pointers
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device.index == self.device):
            raise TypeError(f"{name} must be a tensor on cuda:{self.device}")
        if t.dtype != dtype or not t.is_contiguous():
            raise TypeError(f"{name} must be contiguous {dtype}")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            lib().m17gpu_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _check(lib().m17gpu_reset(self._ctx, self._stream()), "m17gpu_reset")

    # ---- hot path -------------------------------------------------------
    def alloc_outputs(self, nblk, rec_cap=None, want_syms=False):
        import torch
        dev = f"cuda:{self.device}"
        rec_cap = self.rec_cap_max if rec_cap is None else rec_cap
        out = {
            "recs": torch.zeros((self.C, rec_cap, 64), dtype=torch.uint8, device=dev),
            "counts": torch.zeros((self.C,), dtype=torch.int32, device=dev),
            "syms": None, "nsyms": None, "rec_cap": rec_cap, "nblk": int(nblk), "C": self.C,
        }
        if want_syms:
            out["syms"] = torch.zeros((self.C, _lib.sym_stride(nblk)), dtype=torch.float32, device=dev)
            out["nsyms"] = torch.zeros((self.C, nblk), dtype=torch.int32, device=dev)
        return out

    def rx_blocks(self, iq, mode, out):
        """iq: int16 cuda tensor [C, nblk, 1920, 2]; out: dict from alloc_outputs."""
        import torch
        if not isinstance(iq, torch.Tensor) or iq.dim() != 4:
            raise ValueError("iq must be a [C, nblk, 1920, 2] int16 tensor")
        nblk = int(iq.shape[1])
        self._chk(iq, torch.int16, (self.C, nblk, 1920, 2), "iq")
        self._chk_out(out, nblk)
        net = getattr(self, "_net", None)
        if int(mode) == 1 and net is not None and int(net.shape[1]) != int(out["rec_cap"]):
            raise ValueError(f"the network sink was attached with rec_cap {int(net.shape[1])}; these outputs have {int(out['rec_cap'])}")
        _check(lib().m17gpu_rx_blocks(self._ctx, _ptr(iq), nblk, int(mode), _ptr(out["recs"]),
                                      int(out["rec_cap"]), _ptr(out["counts"]), _ptr(out["syms"]),
                                      _ptr(out["nsyms"]), self._stream()), "m17gpu_rx_blocks")
        return out

    def _chk_out(self, out, nblk):
        """Outputs must have been allocated for this receiver and this block count: the symbol rows
        are nblk*193+8 floats apart and nsyms is [C, nblk]."""
        import torch
        if out.get("C") != self.C or out.get("nblk") != nblk:
            raise ValueError(f"outputs were allocated for C={out.get('C')}, nblk={out.get('nblk')}; "
                             f"this call has C={self.C}, nblk={nblk}")
        cap = int(out["rec_cap"])
        self._chk(out["recs"], torch.uint8, (self.C, cap, 64), "out['recs']")
        self._chk(out["counts"], torch.int32, (self.C,), "out['counts']")
        self._chk(out["syms"], torch.float32, (self.C, _lib.sym_stride(nblk)), "out['syms']")
        self._chk(out["nsyms"], torch.int32, (self.C, nblk), "out['nsyms']")

    # ---- stage entry points ----------------------------------------------
    def frontend(self, iq):
        import torch
        nblk = int(iq.shape[1])
        self._chk(iq, torch.int16, (self.C, nblk, 1920, 2), "iq")
        disc = torch.empty((self.C, nblk, 384), dtype=torch.float32, device=iq.device)
        offs = torch.empty((self.C, nblk), dtype=torch.float32, device=iq.device)
        _check(lib().m17gpu_frontend(self._ctx, _ptr(iq), nblk, _ptr(disc), _ptr(offs), self._stream()),
               "m17gpu_frontend")
        return disc, offs

    def sync_frame(self, disc, out):
        import torch
        nblk = int(disc.shape[1])
        self._chk(disc, torch.float32, (self.C, nblk, 384), "disc")
        self._chk_out(out, nblk)
        _check(lib().m17gpu_sync_frame(self._ctx, _ptr(disc), nblk, _ptr(out["recs"]), int(out["rec_cap"]),
                                       _ptr(out["counts"]), _ptr(out["syms"]), _ptr(out["nsyms"]), self._stream()),
               "m17gpu_sync_frame")
        return out

    def sync_samples(self, disc, lock):
        """m17_rx_sync_samples alone (m17_rx_sync.cpp:77-99) under an external framer's lock flag:
        disc [C, nblk, 384] DC-free -> (syms [C, nblk*193+8], nsyms [C, nblk])."""
        import torch
        nblk = int(disc.shape[1])
        self._chk(disc, torch.float32, (self.C, nblk, 384), "disc")
        syms = torch.zeros((self.C, _lib.sym_stride(nblk)), dtype=torch.float32, device=disc.device)
        nsyms = torch.zeros((self.C, nblk), dtype=torch.int32, device=disc.device)
        _check(lib().m17gpu_sync_samples(self._ctx, _ptr(disc), nblk, int(bool(lock)), _ptr(syms), _ptr(nsyms),
                                         self._stream()), "m17gpu_sync_samples")
        return syms, nsyms

    def gen_batch(self, nblk, n_stream_frames=40, ebn0_db=200.0, base_seed=0x4D313700, first_channel=0,
                  noise_cutoff_hz=0.0, stages=False):
        """GPU-side signal source (stream mode): the same signal as generate_batch, made on the device.
        Returns dict(iq[C,nblk,1920,2] int16 cuda, lsf[C,30], payload[C,F,16], nframes[C]) of cuda tensors; with
        stages=True also dibits[C,nblk+1,192] uint8 and phase[C,nblk*1920] float32 (m17gpu_gen_batch_stages)."""
        import torch
        dev = f"cuda:{self.device}"
        max_frames = nblk + 2
        iq = torch.empty((self.C, nblk, 1920, 2), dtype=torch.int16, device=dev)
        lsf = torch.zeros((self.C, 30), dtype=torch.uint8, device=dev)
        pl = torch.zeros((self.C, max_frames, 16), dtype=torch.uint8, device=dev)
        nf = torch.zeros((self.C,), dtype=torch.int32, device=dev)
        out = {"iq": iq, "lsf": lsf, "payload": pl, "nframes": nf}
        if stages:
            out["dibits"] = torch.zeros((self.C, nblk + 1, 192), dtype=torch.uint8, device=dev)
            out["phase"] = torch.zeros((self.C, nblk * 1920), dtype=torch.float32, device=dev)
            _check(lib().m17gpu_gen_batch_stages(self._ctx, base_seed, first_channel, nblk, n_stream_frames, ebn0_db,
                                                 noise_cutoff_hz, _ptr(iq), _ptr(lsf), _ptr(pl), max_frames, _ptr(nf),
                                                 _ptr(out["dibits"]), _ptr(out["phase"]), self._stream()), "m17gpu_gen_batch_stages")
            return out
        _check(lib().m17gpu_gen_batch(self._ctx, base_seed, first_channel, nblk, n_stream_frames, ebn0_db,
                                      noise_cutoff_hz, _ptr(iq), _ptr(lsf), _ptr(pl), max_frames, _ptr(nf), self._stream()),
               "m17gpu_gen_batch")
        return out

    def pluto_decimate(self, wide):
        """wide: int16 cuda tensor [C, n_in, 2] at 384 kHz -> [C, n_in/8, 2] at 48 kHz (radio.cpp:18-40)."""
        import torch
        n_in = int(wide.shape[1])
        self._chk(wide, torch.int16, (self.C, n_in, 2), "wide")
        out = torch.empty((self.C, n_in // 8, 2), dtype=torch.int16, device=wide.device)
        _check(lib().m17gpu_pluto_decimate(self._ctx, _ptr(wide), n_in, _ptr(out), self._stream()), "m17gpu_pluto_decimate")
        return out

    def viterbi_decode(self, soft):
        import torch
        n, length = int(soft.shape[0]), int(soft.shape[1])
        self._chk(soft, torch.float32, (n, length), "soft")
        bits = torch.empty((n, length // 2), dtype=torch.uint8, device=soft.device)
        _check(lib().m17gpu_viterbi_decode(self._ctx, _ptr(soft), _ptr(bits), length, n, self._stream()),
               "m17gpu_viterbi_decode")
        return bits

    def demap_frame(self, sym):
        import torch
        n = int(sym.shape[0])
        self._chk(sym, torch.float32, (n, 192), "sym")
        soft = torch.empty((n, 368), dtype=torch.float32, device=sym.device)
        _check(lib().m17gpu_demap_frame(self._ctx, _ptr(sym), _ptr(soft), n, self._stream()), "m17gpu_demap_frame")
        return soft

    def decode_frames(self, sym, types):
        import torch
        n = int(sym.shape[0])
        self._chk(sym, torch.float32, (n, 192), "sym")
        self._chk(types, torch.uint8, (n,), "types")
        recs = torch.zeros((n, 64), dtype=torch.uint8, device=sym.device)
        _check(lib().m17gpu_decode_frames(self._ctx, _ptr(sym), _ptr(types), _ptr(recs), n, self._stream()),
               "m17gpu_decode_frames")
        return recs

    def golay_decode(self, words):
        import torch
        n = int(words.shape[0])
        self._chk(words, torch.int32, (n,), "words")
        out = torch.empty((n,), dtype=torch.int16, device=words.device)
        _check(lib().m17gpu_golay_decode(self._ctx, _ptr(words), _ptr(out), n, self._stream()), "m17gpu_golay_decode")
        return out

    # ---- record compaction for the multi-GPU gather ---------------
    def pack_records(self, out, packed=None, offsets=None):
        """The valid records of a step, channel-major: (packed uint8 [C*rec_cap, 64] -- rows [0, offsets[C]) are valid --,
        offsets int32 [C+1]); device tensors, nothing is read back (m17gpu_pack_records)."""
        import torch
        cap = int(out["rec_cap"])
        dev = out["recs"].device
        if packed is None:
            packed = torch.empty((self.C * cap, 64), dtype=torch.uint8, device=dev)
        if offsets is None:
            offsets = torch.empty((self.C + 1,), dtype=torch.int32, device=dev)
        self._chk(out["recs"], torch.uint8, (self.C, cap, 64), "out['recs']")
        self._chk(offsets, torch.int32, (self.C + 1,), "offsets")
        if packed.dtype != torch.uint8 or packed.dim() != 2 or packed.shape[1] != 64 or not packed.is_contiguous():
            raise ValueError("packed must be a contiguous uint8 tensor [rows, 64]")
        _check(lib().m17gpu_pack_records(self._ctx, _ptr(out["recs"]), cap, _ptr(out["counts"]), _ptr(packed),
                                         int(packed.shape[0]), _ptr(offsets), self._stream()), "m17gpu_pack_records")
        return packed, offsets

    def unpack_records(self, packed, offsets, rec_cap):
        """Packed rows + offsets [n+1] back into (recs [n, rec_cap, 64], counts [n]) -- the layout rx_blocks writes."""
        import torch
        n = int(offsets.shape[0]) - 1
        recs = torch.empty((n, int(rec_cap), 64), dtype=torch.uint8, device=packed.device)
        counts = torch.empty((n,), dtype=torch.int32, device=packed.device)
        _check(lib().m17gpu_unpack_records(self._ctx, _ptr(packed), _ptr(offsets), n, _ptr(recs), int(rec_cap), _ptr(counts),
                                           self._stream()), "m17gpu_unpack_records")
        return recs, counts

    # ---- output wire format on the device -----------------------
    def set_net_output(self, rec_cap=None, stream_ids=None, dst_override=0):
        """Attach the network sink: returns the uint8 tensor [C, rec_cap, 56] that rx_blocks(mode 1) fills with the
        54-byte M17-over-IP frame of every DELIVERED record (m17_net.cpp:25-74).  rec_cap must be the one of the
        outputs passed to rx_blocks.  stream_ids: optional uint16-valued int16/uint16 tensor [C]."""
        import torch
        rec_cap = self.rec_cap_max if rec_cap is None else int(rec_cap)
        net = torch.zeros((self.C, rec_cap, 56), dtype=torch.uint8, device=f"cuda:{self.device}")
        if stream_ids is not None:
            self._chk(stream_ids, torch.int16, (self.C,), "stream_ids")
        # the library first: a refused call (rec_cap out of range, ...) must leave the sink that IS attached -- and the
        # tensor the context still points at -- alive
        _check(lib().m17gpu_set_net_output(self._ctx, _ptr(net), rec_cap, _ptr(stream_ids), int(dst_override)),
               "m17gpu_set_net_output")
        self._net, self._sids = net, stream_ids                      # keep them alive while attached
        return net

    def clear_net_output(self):
        _check(lib().m17gpu_set_net_output(self._ctx, C.c_void_p(0), 0, C.c_void_p(0), 0), "m17gpu_set_net_output")
        self._net = self._sids = None

    def parse_lsf_batch(self, lsf):
        """lsf: uint8 cuda tensor [n, 30] -> uint8 tensor [n, 64] of m17gpu_lsf_fields structs."""
        import torch
        n = int(lsf.shape[0])
        self._chk(lsf, torch.uint8, (n, 30), "lsf")
        out = torch.zeros((n, 64), dtype=torch.uint8, device=lsf.device)
        _check(lib().m17gpu_parse_lsf_batch(self._ctx, _ptr(lsf), _ptr(out), n, self._stream()), "m17gpu_parse_lsf_batch")
        return out

    def set_option(self, name, value):
        _check(lib().m17gpu_set_option(self._ctx, name.encode(), int(value)), "m17gpu_set_option")

    # ---- measurement ---------------------------------------------------------
    def set_profiling(self, on):
        _check(lib().m17gpu_set_profiling(self._ctx, int(bool(on))), "m17gpu_set_profiling")

    def kernel_ms(self):
        """Average ms per launch of (k_frontend, k_sync_frame, k_worklist + k_decode, k_bookkeeping), number of calls."""
        ms = (C.c_float * 4)()
        n = C.c_int()
        _check(lib().m17gpu_get_kernel_ms(self._ctx, ms, C.byref(n)), "m17gpu_get_kernel_ms")
        return list(ms), n.value

    def call_ms(self):
        """(stage ms [4], average ms of a whole rx_blocks call on the caller's stream, calls averaged)."""
        ms = (C.c_float * 4)()
        call = C.c_float()
        n = C.c_int()
        _check(lib().m17gpu_get_call_ms(self._ctx, ms, C.byref(call), C.byref(n)), "m17gpu_get_call_ms")
        return list(ms), call.value, n.value

    def selftest(self):
        """Mismatch counts of the exhaustive exact-arithmetic self test (must be all zero)."""
        bad = (C.c_uint * 4)()
        _check(lib().m17gpu_selftest(self._ctx, bad), "m17gpu_selftest")
        return list(bad)

    # ---- state -------------------------------------------------------------
    def lsf(self):
        a = np.zeros((self.C, 2, 30), np.uint8)
        _check(lib().m17gpu_get_lsf(self._ctx, a.ctypes.data_as(C.c_void_p)), "m17gpu_get_lsf")
        return a

    def counters(self):
        a = np.zeros((self.C, 4), np.uint32)
        _check(lib().m17gpu_get_counters(self._ctx, a.ctypes.data_as(C.c_void_p)), "m17gpu_get_counters")
        return a

    def afc_delta(self):
        a = np.zeros((self.C,), np.float32)
        _check(lib().m17gpu_get_afc(self._ctx, a.ctypes.data_as(C.c_void_p)), "m17gpu_get_afc")
        return a

    def timing_state(self):
        """Timing-loop / framer control state per channel under the reference's names (m17_rx_sync.cpp:6-11,78,
        m17_rx_frame.cpp:16-18, m17_dsp.cpp:196): dict of arrays [C, ...]; m_buff is [C, 31] with column 0 zero (the
        library keeps m_buff[1 .. 30]: element 0 leaves the window with the next input)."""
        ai = np.zeros((self.C, 6), np.int32)
        af = np.zeros((self.C, 36), np.float32)
        _check(lib().m17gpu_get_timing_state(self._ctx, ai.ctypes.data_as(C.c_void_p), af.ctypes.data_as(C.c_void_p)),
               "m17gpu_get_timing_state")
        buff = np.zeros((self.C, 31), np.float32)
        buff[:, 1:] = af[:, 6:36]
        return {"m_clk": ai[:, 0], "m_thr": ai[:, 1], "m_index": ai[:, 2], "m_flock": ai[:, 3], "m_fclk": ai[:, 4],
                "m_frame_errors": ai[:, 5], "sum": af[:, 0].copy(), "dif": af[:, 1].copy(), "z": af[:, 2:6].copy(), "m_buff": buff}

    def last_path(self):
        """What the last rx_blocks call ran: dict(fir=1|4|5, plain_slots, book=0|1|2) (m17gpu_get_last_path)."""
        a = (C.c_int * 4)()
        _check(lib().m17gpu_get_last_path(self._ctx, a), "m17gpu_get_last_path")
        return {"fir": a[0], "plain_slots": a[1], "book": a[2]}

    def lock(self):
        a = np.zeros((self.C,), np.uint8)
        _check(lib().m17gpu_get_lock(self._ctx, a.ctypes.data_as(C.c_void_p)), "m17gpu_get_lock")
        return a


def generate_channel(seed, nblk, n_stream_frames=40, delay=0, ebn0_db=200.0, packet_mode=0, max_frames=None,
                     noise_cutoff_hz=0.0):
    """Host signal source for one channel: (iq[nblk,1920,2] int16, lsf[30], payloads[n,16], n)."""
    max_frames = max_frames or (nblk + 2)
    iq = np.zeros((nblk, 1920, 2), np.int16)
    lsf = np.zeros(30, np.uint8)
    pl = np.zeros((max_frames, 16), np.uint8)
    p = _lib.GenParams(seed, n_stream_frames, delay, ebn0_db, packet_mode, noise_cutoff_hz)
    n = lib().m17gen_channel(C.byref(p), nblk, iq.ctypes.data_as(C.c_void_p), lsf.ctypes.data_as(C.c_void_p),
                             pl.ctypes.data_as(C.c_void_p), max_frames)
    if n < 0:
        raise RuntimeError(f"m17gen_channel failed ({n})")
    return iq, lsf, pl, n


def generate_batch(n_channels, nblk, n_stream_frames=40, ebn0_db=200.0, base_seed=0x4D313700,
                   first_channel=0, packet_mode=0, nthreads=8, out=None, noise_cutoff_hz=0.0):
    """Host signal source for C channels: dict(iq[C,nblk,1920,2], lsf[C,30], payload[C,F,16], nframes[C])."""
    max_frames = nblk + 2
    iq = out if out is not None else np.zeros((n_channels, nblk, 1920, 2), np.int16)
    lsf = np.zeros((n_channels, 30), np.uint8)
    pl = np.zeros((n_channels, max_frames, 16), np.uint8)
    nf = np.zeros((n_channels,), np.int32)
    rc = lib().m17gen_batch(n_channels, base_seed, first_channel, nblk, n_stream_frames, ebn0_db, noise_cutoff_hz,
                            packet_mode,
                            iq.ctypes.data_as(C.c_void_p), lsf.ctypes.data_as(C.c_void_p),
                            pl.ctypes.data_as(C.c_void_p), max_frames, nf.ctypes.data_as(C.c_void_p), nthreads)
    if rc != 0:
        raise RuntimeError(f"m17gen_batch failed ({rc})")
    return {"iq": iq, "lsf": lsf, "payload": pl, "nframes": nf}
