"""imageflow_amd -- MI355X (gfx950) implementation of imageflow's pixel hot path.

The product is libimageflow_hip.so (C ABI in include/imageflow_hip.h, sources in imageflow_amd/csrc).
This package is the host-side mirror of the reference's Rust interface for the same path
(imageflow_core::graphics::{scaling, weights, color, blend, bitmaps, color_matrix, copy_rect, flip, transpose},
codecs::{mozjpeg_decoder, mozjpeg} and the flow nodes in front of them) used by tests and bench.py;
PyTorch supplies device memory, streams and torch.distributed only.
"""
from .errors import ErrorKind, FlowError  # noqa: F401


def trim_cache(keep_device_bytes=0, keep_host_bytes=0):
    """Give the library's recycled device / pinned blocks back to the driver (ifhip_cache_trim): call it
    next to torch.cuda.empty_cache() or after a torch out-of-memory error.  -> (device, host) bytes released."""
    from . import _native
    return _native.trim_cache(keep_device_bytes, keep_host_bytes)
