"""ctypes binding of libimageflow_hip.so.  Fails loudly when the library is missing: there is no CPU path."""
import ctypes as C
import os

from .errors import FlowError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IFHIP_LIB") or os.path.join(_HERE, "lib", "libimageflow_hip.so")
_lib = None

u8p, u32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -m imageflow_amd.build` (hipcc, gfx950). "
                          "imageflow_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64.so.7
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.ifhip_last_error_message.restype = C.c_char_p
    L.ifhip_version.restype = C.c_char_p
    L.ifhip_stride_for_width.restype = C.c_uint32
    L.ifhip_stride_for_width.argtypes = [C.c_uint32]
    L.ifhip_populate_weights.argtypes = [C.c_int, C.c_int, C.c_float, C.c_double, C.c_uint32, C.c_uint32,
                                         u32p, u32p, f32p, C.c_uint32, u32p]
    L.ifhip_table_srgb_to_floatspace.argtypes = [C.c_int, f32p]
    L.ifhip_table_linear_to_srgb.argtypes = [u8p]
    L.ifhip_scale_and_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                         C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_int, C.c_float, C.c_int, C.c_int, C.c_uint32]
    L.ifhip_resample_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_int, C.c_float]
    L.ifhip_resample_plan_destroy.argtypes = [C.c_void_p]
    L.ifhip_resample_plan_destroy.restype = None
    L.ifhip_resample_plan_kernel_kind.argtypes = [C.c_void_p, C.c_int]
    _batch = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_uint32,
              C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
              C.c_int, C.c_int, C.c_uint32]
    L.ifhip_scale_and_render_batch_device.argtypes = _batch + [C.c_void_p, C.c_int, C.c_void_p]
    L.ifhip_time_scale_and_render_batch_device.argtypes = _batch + [C.c_int, C.c_void_p, C.c_int, f32p]
    L.ifhip_measure_copy_bandwidth.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.ifhip_measure_read_bandwidth.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.ifhip_apply_matte.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    L.ifhip_apply_matte_batch_device.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_int, C.c_uint32, C.c_void_p]
    _lib = L
    if not hasattr(L, "ifhip_debug_set"):          # (a library build of an earlier round, loaded through IFHIP_LIB for an A/B run)
        return L
    L.ifhip_debug_set.argtypes = [C.c_char_p, C.c_char_p]
    # development convenience of THIS binding (the library itself reads no environment variable): IFHIP_<SWITCH>=v in the
    # environment of a tools/ run becomes ifhip_debug_set("<switch>", v) once, at load
    for k, v in os.environ.items():
        if k.startswith("IFHIP_") and k not in ("IFHIP_LIB", "IFHIP_BUILD_IF_STALE"):
            L.ifhip_debug_set(k[6:].lower().encode(), v.encode())
    return L


def debug_set(key, value):
    """ifhip_debug_set: a development switch of the library (tests, tools/); value None unsets."""
    check(lib().ifhip_debug_set(key.encode(), None if value is None else str(value).encode()))


def set_cu_budget(compute_units):
    """ifhip_set_cu_budget: the CUs this process's resample launches plan for (0: all 256) -- what a host that overlaps a few
    workgroups of other work (an RCCL gather) with them leaves."""
    L = lib()
    L.ifhip_set_cu_budget.argtypes = [C.c_uint32]
    check(L.ifhip_set_cu_budget(int(compute_units)))


def trim_cache(keep_device_bytes=0, keep_host_bytes=0):
    """ifhip_cache_trim: give the library's recycled device / pinned blocks back to the driver (what a torch program calls
    next to torch.cuda.empty_cache() or after an out-of-memory error; blocks in use are not touched).  -> (device, host) bytes released."""
    d, h = C.c_size_t(0), C.c_size_t(0)
    L = lib()
    L.ifhip_cache_trim.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    check(L.ifhip_cache_trim(keep_device_bytes, keep_host_bytes, C.byref(d), C.byref(h)))
    return d.value, h.value


class CacheStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "device_hits", "device_driver_allocs", "device_driver_frees", "device_oom_flushes", "device_wide_syncs",
        "device_bytes_cached", "device_bytes_live", "device_blocks_live", "device_limit_bytes",
        "host_hits", "host_driver_allocs", "host_driver_frees",
        "host_bytes_cached", "host_bytes_live", "host_blocks_live", "host_limit_bytes")]


def cache_stats():
    """ifhip_cache_stats as a dict."""
    st = CacheStats()
    L = lib()
    L.ifhip_cache_stats.argtypes = [C.POINTER(CacheStats)]
    check(L.ifhip_cache_stats(C.byref(st)))
    return {n: getattr(st, n) for n, _ in CacheStats._fields_}


def set_cache_limits(device_bytes, host_bytes):
    L = lib()
    L.ifhip_cache_set_limits.argtypes = [C.c_size_t, C.c_size_t]
    check(L.ifhip_cache_set_limits(device_bytes, host_bytes))


def check(rc):
    if rc != 0:
        msg = lib().ifhip_last_error_message()
        raise FlowError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
