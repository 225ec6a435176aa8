"""m17_sdr_amd -- MI355X-native batched M17 receive chain.

The product is the HIP/C++ shared library `libm17gpu.so` (C-ABI in
include/m17gpu.h).  This package is a thin host-side convenience layer for
tests and benchmarks: it loads the library and moves torch device pointers and
streams across the boundary.  No computation happens in Python.
"""
from . import _lib
from ._lib import (Rec, GenParams, BLOCK_SAMPLES, DISC_OUT, FRAME_SYMS, SOFT_BITS, sym_stride,
                   F_SYNC_OK, F_PARSED, F_LICH_OK, F_DELIVERED, F_EOT, F_LOST, F_LSF_GATE,
                   F_PKT_VALID, F_AOS, OK, ERR_NO_DEVICE, ERR_HIP, ERR_ARG, ERR_NOMEM)
from .api import Receiver, generate_batch, generate_channel, lib

__all__ = ["Receiver", "generate_batch", "generate_channel", "lib", "Rec", "GenParams",
           "BLOCK_SAMPLES", "DISC_OUT", "FRAME_SYMS", "SOFT_BITS", "sym_stride"]
