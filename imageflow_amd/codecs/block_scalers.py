"""Host mirror of imageflow's 8x8 -> NxN spatial block scalers (c_components/lib/codecs_jpeg_idct_fast.h:17-43):
flow_scale_spatial[_srgb]_NxN over arrays of blocks, computed by libimageflow_hip.so."""
import ctypes as C

import numpy as np

from .. import _native


def _bind():
    L = _native.lib()
    if not getattr(L, "_bs_bound", False):
        L.ifhip_block_scaler_tables.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.ifhip_scale_spatial_blocks.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.ifhip_scale_spatial_plane_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                       C.c_void_p, C.c_uint32, C.c_void_p]
        L._bs_bound = True
    return L


def tables(n):
    L = _bind()
    w = np.zeros((7, 8), np.int8)
    d = np.zeros(7, np.uint8)
    s2l = np.zeros(256, np.uint16)
    l2s = np.zeros(4096, np.uint8)
    _native.check(L.ifhip_block_scaler_tables(n, w.ctypes.data, d.ctypes.data, s2l.ctypes.data, l2s.ctypes.data))
    return w[:n], d[:n], s2l, l2s


def flow_scale_spatial(blocks, n, srgb):
    """blocks: uint8 [k][64] -> uint8 [k][n][n] (host buffers, GPU compute)."""
    L = _bind()
    blocks = np.ascontiguousarray(blocks, np.uint8).reshape(-1, 64)
    out = np.zeros((blocks.shape[0], n, n), np.uint8)
    _native.check(L.ifhip_scale_spatial_blocks(blocks.ctypes.data, blocks.shape[0], n, int(bool(srgb)), out.ctypes.data))
    return out


def flow_scale_spatial_plane(plane, n, srgb):
    """plane: uint8 cuda tensor [8*bh, 8*bw] -> uint8 cuda tensor [n*bh, n*bw]."""
    import torch
    L = _bind()
    ph, pw = plane.shape
    out = torch.zeros((ph // 8 * n, pw // 8 * n), dtype=torch.uint8, device=plane.device)
    stream = torch.cuda.current_stream(plane.device).cuda_stream
    with torch.cuda.device(plane.device):
        _native.check(L.ifhip_scale_spatial_plane_device(plane.data_ptr(), plane.stride(0), pw // 8, ph // 8, n,
                                                         int(bool(srgb)), out.data_ptr(), out.stride(0), C.c_void_p(stream)))
    return out
