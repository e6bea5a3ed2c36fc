"""CPU-side checks of the PRODUCT library (no GPU): it loads, exports every symbol the header declares, and its
host-built tables (weights, LUTs, stride rule) equal the golden vectors and the oracle bit for bit."""
import ctypes
import gzip
import json
import os
import re

import numpy as np
import pytest

from imageflow_amd import _native
from imageflow_amd.errors import ErrorKind, FlowError
from imageflow_amd.graphics import color as Cc
from imageflow_amd.graphics import weights as W
from imageflow_amd.graphics.bitmaps import color32, get_stride
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "imageflow_hip.h")).read()
    names = set(re.findall(r"IFHIP_API\s+[\w\s\*]+?\b(ifhip_\w+)\s*\(", hdr))
    assert len(names) >= 15
    L = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_abi_header_declares_outer_abi_symbols_if_present():
    p = os.path.join(ROOT, "include", "imageflow_abi_subset.h")
    if not os.path.exists(p):
        pytest.skip("outer ABI subset not built yet")
    hdr = open(p).read()
    names = set(re.findall(r"\b(imageflow_\w+)\s*\(", hdr))
    L = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_version_and_error_message():
    L = _native.lib()
    assert b"gfx950" in L.ifhip_version()
    with pytest.raises(FlowError) as e:
        W.populate_weights(99, 10, 10)
    assert e.value.kind == ErrorKind.InvalidArgument


def test_stride_rule():
    for w in (1, 15, 16, 17, 200, 3840, 7680):
        assert get_stride(w) == O.stride_for_width(w) == ((w * 4 + 63) // 64) * 64
    assert get_stride(3840) == 15360 and get_stride(200) == 832 and get_stride(400) == 1600


def test_color32_parsing():
    assert color32("FFFFFFFF") == 0xFFFFFFFF
    assert color32("#FF0000") == 0xFFFF0000
    assert color32("0f08") == 0x8800FF00


def test_weights_match_oracle_bitwise_all_filters():
    for fid in range(1, 32):
        for (i, o) in ((3840, 200), (2160, 113), (7, 3), (100, 250), (50, 50), (1, 1), (5, 17)):
            try:
                l, c, w = O.weights(fid, o, i)
            except RuntimeError:
                with pytest.raises(FlowError):
                    W.populate_weights(fid, o, i)
                continue
            p = W.populate_weights(fid, o, i)
            assert np.array_equal(p.left_pixel, l) and np.array_equal(p.count, c), (fid, i, o)
            assert np.array_equal(p.weights.view(np.uint32), w.view(np.uint32)), (fid, i, o)


def test_weights_sharpen_and_lobe_modes_match_oracle():
    for fid in (2, 6, 13, 24):
        for mode, val, ks in ((2, 15.0, 1.0), (2, 50.0, 1.1), (1, 0.05, 1.0), (1, 0.0, 0.8), (0, 0.0, 1.2)):
            try:
                l, c, w = O.weights(fid, 33, 100, mode, val, ks)
            except RuntimeError:
                with pytest.raises(FlowError):
                    W.populate_weights(fid, 33, 100, mode, val, ks)
                continue
            p = W.populate_weights(fid, 33, 100, mode, val, ks)
            assert np.array_equal(p.weights.view(np.uint32), w.view(np.uint32))


def test_weights_against_golden_directly(golden_dir):
    """The product's own table builder against the reference's golden file (not via the oracle)."""
    g = json.loads(gzip.open(os.path.join(golden_dir, "weights_golden.json.gz")).read())
    bad = 0
    for fid, frm, to, rows in g["plain"]:
        p = W.populate_weights(fid, to, frm)
        off = 0
        for n, row in zip(p.count, rows):
            got = ["%.6f" % float(v) for v in p.weights[off:off + n]]
            off += n
            if len(got) != len(row) or any(a != b and float(a) != float(b) for a, b in zip(got, row)):
                bad += 1
    assert bad == 0


def test_colour_tables_match_reference_and_oracle(golden_dir):
    ref = np.frombuffer(open(os.path.join(golden_dir, "linear_to_srgb_lut.bin"), "rb").read(), np.uint8)
    assert np.array_equal(Cc.linear_to_srgb_table(), ref)
    s2l, s2f, l2s = O.tables()
    assert np.array_equal(Cc.srgb_to_floatspace_table(1).view(np.uint32), s2l.view(np.uint32))
    assert np.array_equal(Cc.srgb_to_floatspace_table(0).view(np.uint32), s2f.view(np.uint32))
    # pin libm powf across hosts: committed table from the build container
    pinned = np.frombuffer(open(os.path.join(golden_dir, "srgb_to_linear_f32.bin"), "rb").read(), np.float32)
    assert np.array_equal(pinned.view(np.uint32), s2l.view(np.uint32))


def test_threshold_form_of_the_l2s_table_is_equivalent():
    """The fused kernel encodes linear->sRGB by upper_bound over 256 thresholds; that must reproduce the table."""
    import ctypes as C
    thr = np.zeros(256, np.uint16)
    _native.lib().ifhip_table_linear_to_srgb_thresholds.argtypes = [C.c_void_p]
    assert _native.lib().ifhip_table_linear_to_srgb_thresholds(thr.ctypes.data) == 0
    table = Cc.linear_to_srgb_table()
    assert np.all(np.diff(table.astype(int)) >= 0)
    rebuilt = np.searchsorted(thr.astype(np.int64), np.arange(16384), side="right")
    assert np.array_equal(rebuilt.astype(np.uint8), table)


def test_no_cpu_fallback_without_gpu():
    """Without a GPU every compute entry point must refuse loudly (never fall back to a CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render_host
    src = np.zeros((4, 64), np.uint8)
    dst = np.zeros((2, 64), np.uint8)
    with pytest.raises(FlowError) as e:
        scale_and_render_host(src, 4, 4, 64, False, dst, 2, 2, 64, ScaleAndRenderParams(0, 0, 2, 2))
    assert e.value.kind in (ErrorKind.GpuUnavailable, ErrorKind.GpuError)
    # argument validation still comes first and uses the reference's error kind (scaling.rs:24-29)
    with pytest.raises(FlowError) as e:
        scale_and_render_host(src, 4, 4, 64, False, dst, 2, 2, 64, ScaleAndRenderParams(1, 0, 2, 2))
    assert e.value.kind == ErrorKind.InvalidArgument


def test_generated_rust_bindings_cover_the_header():
    """bindings/hip_interop.rs (tools/gen_rust_bindings.py) names exactly the functions include/imageflow_hip.h declares."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "imageflow_hip.h")).read()
    rs = open(os.path.join(root, "bindings", "hip_interop.rs")).read()
    declared = set(re.findall(r"IFHIP_API\s+[^;{(]*?\b(ifhip_\w+)\s*\(", header))
    bound = set(re.findall(r"pub fn (ifhip_\w+)\(", rs))
    assert declared == bound and len(bound) > 40


def test_header_is_plain_c_and_links_with_c_linkage(tmp_path):
    """tests/c_abi_smoke.c: a C99 program (-pedantic) including include/imageflow_hip.h and linking libimageflow_hip.so,
    calling host-only entry points."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "imageflow_amd", "lib")
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi_smoke.c"), "-o", exe, "-L", libdir, "-limageflow_hip",
                    f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "c abi ok" in out.stdout, out.stdout + out.stderr


def test_the_library_reads_no_environment_variable():
    """A drop-in .so whose kernel choice depends on the host's environment is not a product: the development switches
    exist only behind ifhip_debug_set (tests, tools/).  No IFHIP_* name is left in the binary, getenv is not imported, and
    a switch set through the entry point is seen by the code that asks for it (the decode-table pool of the scan report)."""
    import subprocess
    from imageflow_amd import _native
    path = _native.LIB_PATH
    names = subprocess.run(["strings", path], capture_output=True, text=True, check=True).stdout
    assert "IFHIP_" not in names
    undefined = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    L = _native.lib()
    assert L.ifhip_debug_set(None, b"1") != 0 and L.ifhip_debug_set(b"", b"1") != 0
    _native.debug_set("no_such_switch", "1")
    _native.debug_set("no_such_switch", None)


def test_cu_budget_is_validated_on_the_host():
    """ifhip_set_cu_budget (the CUs the resample launches plan for while a host overlaps other work with them): host-side state,
    0 = all, anything above the device's 256 CUs is refused with the reference's InvalidArgument kind."""
    from imageflow_amd import _native
    from imageflow_amd.errors import ErrorKind, FlowError
    for ok in (0, 1, 248, 256, 0):
        _native.set_cu_budget(ok)
    with pytest.raises(FlowError) as e:
        _native.set_cu_budget(257)
    assert e.value.kind == ErrorKind.InvalidArgument
    _native.set_cu_budget(0)
