"""ctypes binding of the C-ABI in include/m17gpu.h (libm17gpu.so, built in-tree).

The library is the product; this module only declares its entry points.  There
is no Python or CPU implementation behind it: if the shared object is missing
the import fails loudly, and compute entry points fail with ERR_NO_DEVICE when
no HIP device is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libm17gpu.so")

BLOCK_SAMPLES = 1920
DISC_OUT = 384
FRAME_SYMS = 192
SOFT_BITS = 368

F_SYNC_OK, F_PARSED, F_LICH_OK, F_DELIVERED = 0x1, 0x2, 0x4, 0x8
F_EOT, F_LOST, F_LSF_GATE, F_PKT_VALID, F_AOS = 0x10, 0x20, 0x40, 0x80, 0x100
OK, ERR_NO_DEVICE, ERR_HIP, ERR_ARG, ERR_NOMEM = 0, -1, -2, -3, -4      # M17GPU_* status codes


def sym_stride(nblk):
    return nblk * 193 + 8


class Rec(C.Structure):
    """m17gpu_rec, 64 bytes."""
    _fields_ = [("type", C.c_uint8), ("votes", C.c_uint8), ("golay_errs", C.c_uint8),
                ("frame_errors", C.c_uint8), ("flags", C.c_uint16), ("fn", C.c_uint16),
                ("variance", C.c_float), ("block", C.c_uint32), ("sym_pos", C.c_uint16),
                ("rsv0", C.c_uint16), ("data", C.c_uint8 * 32), ("rsv", C.c_uint8 * 12)]


class GenParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_stream_frames", C.c_int32), ("delay_samples", C.c_int32),
                ("ebn0_db", C.c_float), ("packet_mode", C.c_int32), ("noise_cutoff_hz", C.c_float)]


# name -> (restype, argtypes); every symbol include/m17gpu.h declares
_vp, _i, _u64 = C.c_void_p, C.c_int, C.c_uint64
SIGNATURES = {
    "m17gpu_create": (_i, [C.POINTER(_vp), _i, _i, _i]),
    "m17gpu_destroy": (None, [_vp]),
    "m17gpu_reset": (_i, [_vp, _vp]),
    "m17gpu_last_error": (C.c_char_p, []),
    "m17gpu_device_count": (_i, []),
    "m17gpu_channels": (_i, [_vp]),
    "m17gpu_rx_blocks": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "m17gpu_frontend": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "m17gpu_sync_frame": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "m17gpu_pluto_decimate": (_i, [_vp, _vp, _i, _vp, _vp]),
    "m17gpu_sync_samples": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "m17gpu_viterbi_decode": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "m17gpu_demap_frame": (_i, [_vp, _vp, _vp, _i, _vp]),
    "m17gpu_decode_frames": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "m17gpu_golay_decode": (_i, [_vp, _vp, _vp, _i, _vp]),
    "m17gpu_set_option": (_i, [_vp, C.c_char_p, _i]),
    "m17gpu_set_profiling": (_i, [_vp, _i]),
    "m17gpu_get_kernel_ms": (_i, [_vp, _vp, _vp]),
    "m17gpu_get_call_ms": (_i, [_vp, _vp, _vp, _vp]),
    "m17gpu_selftest": (_i, [_vp, _vp]),
    "m17gpu_get_lsf": (_i, [_vp, _vp]),
    "m17gpu_get_counters": (_i, [_vp, _vp]),
    "m17gpu_get_lock": (_i, [_vp, _vp]),
    "m17gpu_get_afc": (_i, [_vp, _vp]),
    "m17gpu_get_timing_state": (_i, [_vp, _vp, _vp]),
    "m17gpu_get_last_path": (_i, [_vp, _vp]),
    "m17gpu_get_taps": (_i, [_vp, _vp]),
    "m17gpu_get_golay_tables": (_i, [_vp, _vp]),
    "m17gpu_get_constant": (_i, [C.c_char_p, _vp, _i]),
    "m17gpu_format_net_frame": (_i, [C.c_uint16, _vp, C.c_uint16, _vp, _u64, _vp]),
    "m17gpu_parse_lsf": (_i, [_vp, _vp]),
    "m17gpu_pack_records": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "m17gpu_unpack_records": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "m17gpu_shard_gather_packed": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "m17gpu_shard_set_library": (_i, [C.c_char_p]),
    "m17gpu_set_net_output": (_i, [_vp, _vp, _i, _vp, _u64]),
    "m17gpu_shard_range": (None, [_i, _i, _i, _vp, _vp]),
    "m17gpu_shard_scatter_iq": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "m17gpu_shard_gather_records": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "m17gpu_parse_lsf_batch": (_i, [_vp, _vp, _vp, _i, _vp]),
    "m17gen_channel": (_i, [C.POINTER(GenParams), _i, _vp, _vp, _vp, _i]),
    "m17gen_batch": (_i, [_i, _u64, _i, _i, _i, C.c_float, C.c_float, _i, _vp, _vp, _vp, _i, _vp, _i]),
    "m17gen_stream_frame_dibits": (_i, [_vp, _i, C.c_uint16, _vp, _vp]),
    "m17gen_lsf_frame_dibits": (_i, [_vp, _vp]),
    "m17gen_packet_frame_dibits": (_i, [_vp, _i, _i, _i, _vp]),
    "m17gen_build_lsf": (_i, [_u64, _u64, C.c_uint16, _vp, _vp]),
    "m17gen_encode_call": (_u64, [C.c_char_p]),
    "m17gen_modulate": (_i, [_vp, _i, _vp, _i]),
    "m17gpu_gen_batch": (_i, [_vp, _u64, _i, _i, _i, C.c_float, C.c_float, _vp, _vp, _vp, _i, _vp, _vp]),
    "m17gpu_gen_batch_stages": (_i, [_vp, _u64, _i, _i, _i, C.c_float, C.c_float, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Load libm17gpu.so (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C m17_sdr_amd/csrc` (hipcc, gfx950).  There is no CPU fallback.")
    # PyTorch ships its own libamdhip64; the Python host layer hands torch device pointers to
    # this library, so both must sit on ONE HIP runtime: let torch load its copy first (the
    # SONAME then resolves to it).  Loaded the other way round the process ends up with two
    # runtimes and m17gpu_create sees no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
