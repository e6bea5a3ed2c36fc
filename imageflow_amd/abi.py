"""ctypes binding of the libimageflow C-ABI subset that libimageflow_hip.so re-exports (include/imageflow_abi_subset.h,
csrc/abi_shim.cpp) -- written the way a language binding over bindings/headers/imageflow_default.h is written:
create a context, add buffers, send a JSON job, read the response, fetch the outputs.  Host logic only."""
import ctypes as C
import json
import struct

import numpy as np

from . import _native

ABI_MAJOR, ABI_MINOR = 3, 2
RAW_MAGIC = b"IFBGRA1\x00"


def _bind():
    L = _native.lib()
    if getattr(L, "_abi_bound", False):
        return L
    vp, u8pp, szp = C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)
    L.imageflow_abi_compatible.argtypes = [C.c_uint32, C.c_uint32]
    L.imageflow_abi_compatible.restype = C.c_bool
    L.imageflow_abi_version_major.restype = C.c_uint32
    L.imageflow_abi_version_minor.restype = C.c_uint32
    L.imageflow_context_create.argtypes = [C.c_uint32, C.c_uint32]
    L.imageflow_context_create.restype = vp
    L.imageflow_context_destroy.argtypes = [vp]
    L.imageflow_context_destroy.restype = None
    for name in ("has_error", "error_recoverable", "error_try_clear", "begin_terminate"):
        f = getattr(L, "imageflow_context_" + name)
        f.argtypes, f.restype = [vp], C.c_bool
    for name in ("error_code", "error_as_exit_code", "error_as_http_code"):
        f = getattr(L, "imageflow_context_" + name)
        f.argtypes, f.restype = [vp], C.c_int32
    L.imageflow_context_error_write_to_buffer.argtypes = [vp, C.c_char_p, C.c_size_t, szp]
    L.imageflow_context_error_write_to_buffer.restype = C.c_bool
    L.imageflow_context_send_json.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
    L.imageflow_context_send_json.restype = vp
    L.imageflow_json_response_read.argtypes = [vp, vp, C.POINTER(C.c_int64), u8pp, szp]
    L.imageflow_json_response_read.restype = C.c_bool
    L.imageflow_json_response_destroy.argtypes = [vp, vp]
    L.imageflow_json_response_destroy.restype = C.c_bool
    L.imageflow_context_add_input_buffer.argtypes = [vp, C.c_int32, C.c_char_p, C.c_size_t, C.c_int]
    L.imageflow_context_add_input_buffer.restype = C.c_bool
    L.imageflow_context_add_output_buffer.argtypes = [vp, C.c_int32]
    L.imageflow_context_add_output_buffer.restype = C.c_bool
    L.imageflow_context_get_output_buffer_by_id.argtypes = [vp, C.c_int32, u8pp, szp]
    L.imageflow_context_get_output_buffer_by_id.restype = C.c_bool
    L.imageflow_context_take_output_buffer.argtypes = [vp, C.c_int32, u8pp, szp]
    L.imageflow_context_take_output_buffer.restype = C.c_bool
    L.imageflow_buffer_free.argtypes = [C.POINTER(C.c_uint8), C.c_size_t]
    L.imageflow_buffer_free.restype = C.c_bool
    L.imageflow_context_request_cancellation.argtypes = [vp]
    L.imageflow_context_request_cancellation.restype = None
    L.imageflow_context_print_and_exit_if_error.argtypes = [vp]
    L.imageflow_context_print_and_exit_if_error.restype = C.c_bool
    L.ifhip_shim_request_cancellation_after_n_polls.argtypes = [vp, C.c_int64]
    L.ifhip_shim_request_cancellation_after_n_polls.restype = None
    L.ifhip_shim_cancellation_polls_remaining.argtypes = [vp]
    L.ifhip_shim_cancellation_polls_remaining.restype = C.c_int64
    L.ifhip_shim_fused_decode_resamples.argtypes = [vp]
    L.ifhip_shim_fused_decode_resamples.restype = C.c_int64
    L.ifhip_shim_device_coded_files.argtypes = [vp]
    L.ifhip_shim_device_coded_files.restype = C.c_int64
    L.ifhip_shim_coalesced_decodes.argtypes = [vp]
    L.ifhip_shim_coalesced_decodes.restype = C.c_int64
    L.ifhip_shim_spread_contexts.argtypes = [C.c_int]
    L.ifhip_shim_spread_contexts.restype = None
    L.ifhip_shim_context_set_device.argtypes = [vp, C.c_int]
    L.ifhip_shim_context_set_device.restype = C.c_bool
    L.ifhip_shim_context_device.argtypes = [vp]
    L.ifhip_shim_context_device.restype = C.c_int
    L.imageflow_context_memory_allocate.argtypes = [vp, C.c_size_t, C.c_char_p, C.c_int32]
    L.imageflow_context_memory_allocate.restype = vp
    L.imageflow_context_memory_free.argtypes = [vp, vp, C.c_char_p, C.c_int32]
    L.imageflow_context_memory_free.restype = C.c_bool
    L._abi_bound = True
    return L


def pack_raw_bgra(rows, w, h, alpha_meaningful=True):
    """EXTENSION container the shim's decode accepts: rows = uint8 [h][stride]."""
    rows = np.ascontiguousarray(rows, np.uint8)
    return RAW_MAGIC + struct.pack("<IIII", w, h, rows.shape[1], int(alpha_meaningful)) + rows.tobytes()


def unpack_raw_bgra(buf):
    """-> (uint8 [h][stride], w, h, alpha_meaningful)"""
    assert buf[:8] == RAW_MAGIC, buf[:8]
    w, h, stride, alpha = struct.unpack("<IIII", buf[8:24])
    return np.frombuffer(buf, np.uint8, h * stride, 24).reshape(h, stride), w, h, bool(alpha)


class Context:
    def __init__(self):
        self.L = _bind()
        self.p = self.L.imageflow_context_create(ABI_MAJOR, ABI_MINOR)
        if not self.p:
            raise RuntimeError("imageflow_context_create failed")
        self._keep = []

    def close(self):
        if self.p:
            self.L.imageflow_context_destroy(self.p)
            self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def add_input_buffer(self, io_id, data: bytes, outlives_context=True):
        self._keep.append(data)
        return self.L.imageflow_context_add_input_buffer(self.p, io_id, data, len(data), 1 if outlives_context else 0)

    def add_output_buffer(self, io_id):
        return self.L.imageflow_context_add_output_buffer(self.p, io_id)

    def send_json(self, method, message):
        """-> (http status, parsed response or None)"""
        body = message if isinstance(message, (bytes, bytearray)) else json.dumps(message).encode()
        r = self.L.imageflow_context_send_json(self.p, method.encode(), body, len(body))
        if not r:
            return None, None
        status, buf, n = C.c_int64(), C.POINTER(C.c_uint8)(), C.c_size_t()
        assert self.L.imageflow_json_response_read(self.p, r, C.byref(status), C.byref(buf), C.byref(n))
        text = C.string_at(buf, n.value).decode()
        try:
            parsed = json.loads(text)
        except ValueError:
            parsed = text
        self.L.imageflow_json_response_destroy(self.p, r)
        return status.value, parsed

    def get_output_buffer(self, io_id):
        buf, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        if not self.L.imageflow_context_get_output_buffer_by_id(self.p, io_id, C.byref(buf), C.byref(n)):
            return None
        return C.string_at(buf, n.value)

    def take_output_buffer(self, io_id):
        """imageflow_context_take_output_buffer + imageflow_buffer_free: the bytes, owned by the caller; None on error."""
        buf, n = C.POINTER(C.c_uint8)(), C.c_size_t()
        if not self.L.imageflow_context_take_output_buffer(self.p, io_id, C.byref(buf), C.byref(n)):
            return None
        data = C.string_at(buf, n.value)
        assert self.L.imageflow_buffer_free(buf, n.value)
        return data

    def set_device(self, ordinal):
        """Bind the context's jobs to a device ordinal (-1: the calling thread's current device); False + a context error otherwise."""
        return self.L.ifhip_shim_context_set_device(self.p, ordinal)

    @property
    def device(self):
        return self.L.ifhip_shim_context_device(self.p)

    def request_cancellation(self):
        self.L.imageflow_context_request_cancellation(self.p)

    def has_error(self):
        return self.L.imageflow_context_has_error(self.p)

    def error_code(self):
        return self.L.imageflow_context_error_code(self.p)

    def error_message(self, cap=2048):
        b = C.create_string_buffer(cap)
        n = C.c_size_t()
        whole = self.L.imageflow_context_error_write_to_buffer(self.p, b, cap, C.byref(n))
        return b.value.decode("utf-8", "replace"), whole
