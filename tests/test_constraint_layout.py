"""The layout arithmetic behind the `constrain` and `watermark` nodes (csrc/layout.cpp = imageflow_riapi::ir4::process_constraint,
ir4/layout.rs:334-412 over sizing.rs) -- host code, no GPU.  Pinned to the known answers the reference's own tests hold
(imageflow_riapi/src/sizing_tests.rs:719-762, ir4/layout.rs:819-949) and to a second restatement in Python
(tools/fuzz_shim_chains.py) on random constraints."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

MODES = ["distort", "within", "fit", "larger_than", "within_crop", "fit_crop", "aspect_crop", "within_pad", "fit_pad"]


def pc(mode, sw, sh, w, h, gravity=None):
    from imageflow_amd import _native
    L = _native.lib()
    L.ifhip_shim_process_constraint.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_float,
                                                C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int)]
    crop, sc, pad, cv, fl = (C.c_uint32 * 4)(), (C.c_int32 * 2)(), (C.c_uint32 * 4)(), (C.c_int32 * 2)(), C.c_int()
    rc = L.ifhip_shim_process_constraint(mode.encode(), sw, sh, -1 if w is None else w, -1 if h is None else h, 0 if gravity is None else 1,
                                         *(gravity or (50.0, 50.0)), crop, sc, pad, cv, C.byref(fl))
    if rc:
        return rc, None, None, None, None
    return 0, list(crop) if fl.value & 1 else None, tuple(sc), list(pad) if fl.value & 2 else None, tuple(cv)


def test_known_answers_of_the_reference():
    # sizing_tests.rs:752-762 test_rounding_99: 1200x400 scaled into 100x33 is 100x33, not 99x33
    assert pc("fit", 1200, 400, 100, 33)[2] == (100, 33)
    # sizing_tests.rs:738-745 test_crop_aspect: 638x423 to the aspect of 200x133 keeps 636x423 of the source
    rc, crop, scale, pad, canvas = pc("aspect_crop", 638, 423, 200, 133)
    assert rc == 0 and (crop[2] - crop[0], crop[3] - crop[1]) == (636, 423) and scale == (636, 423) and pad is None
    # ir4/layout.rs:899-949 test_scale: 5104x3380, w=2560 h=1696 mode=max -> 2560x1695
    assert pc("within", 5104, 3380, 2560, 1696) == (0, None, (2560, 1695), None, (2560, 1695))
    # ir4/layout.rs:819-856 test_crop_and_scale: 768x433, w=100 h=200 mode=crop scale=both -> 217 columns of the source, 100x200.
    # (That test goes through Ir4Layout::align, whose centre is an integer division: x1 = 275; process_constraint aligns with
    # gravity1d's f32 round (:673-683): (768 - 217) * 0.5 = 275.5 -> 276.)
    rc, crop, scale, pad, canvas = pc("fit_crop", 768, 433, 100, 200)
    assert rc == 0 and crop == [276, 0, 493, 433] and scale == (100, 200) and canvas == (100, 200) and pad is None
    # sizing.rs:258-260 test_box_of through the modes that use it: 8x8 into 4x8 (inner) = 4x4, 32x32 over 4x8 (outer) = 8x8
    assert pc("fit", 8, 8, 4, 8)[2] == (4, 4)
    assert pc("fit_crop", 32, 32, 4, 8)[1:3] == ([8, 0, 24, 32], (4, 8))


def test_modes_do_what_their_names_say():
    # within never up-scales, fit does; one side given keeps the ratio; no side given keeps the size
    assert pc("within", 100, 50, 400, 400)[2] == (100, 50) and pc("fit", 100, 50, 400, 400)[2] == (400, 200)
    assert pc("larger_than", 100, 50, 400, 400)[2] == (400, 200) and pc("larger_than", 1000, 500, 400, 400)[2] == (1000, 500)
    assert pc("within", 1000, 500, 100, None)[2] == (100, 50) and pc("distort", 100, 50, 200, None)[2] == (200, 100)
    assert pc("fit_crop", 640, 480, None, None) == (0, None, (640, 480), None, (640, 480))
    assert pc("distort", 100, 50, 30, 70)[2] == (30, 70)
    # pad modes: the image keeps its ratio inside a canvas of the target; the gravity places it
    assert pc("fit_pad", 100, 50, 200, 200) == (0, None, (200, 100), [0, 50, 0, 50], (200, 200))
    assert pc("fit_pad", 100, 50, 200, 200, (0.0, 0.0))[3] == [0, 0, 0, 100] and pc("fit_pad", 100, 50, 200, 200, (0.0, 100.0))[3] == [0, 100, 0, 0]
    assert pc("within_pad", 100, 50, 200, 200) == (0, None, (100, 50), None, (100, 50))       # smaller: "reverts to normal" (layout.rs:200-203)
    # crop modes: gravity picks the part of the source
    assert pc("fit_crop", 200, 100, 50, 50, (0.0, 50.0))[1] == [0, 0, 100, 100] and pc("fit_crop", 200, 100, 50, 50, (100.0, 50.0))[1] == [100, 0, 200, 100]
    assert pc("within_crop", 100, 100, 200, 50) == (0, [0, 25, 100, 75], (100, 50), None, (100, 50))      # larger in one side only: crop to the intersection
    assert pc("nonsense", 10, 10, 5, 5)[0] == 2


def test_the_two_restatements_agree_on_random_constraints():
    import fuzz_shim_chains as F
    rng = np.random.default_rng(5)
    for i in range(30000):
        sw, sh = int(rng.integers(1, 500)), int(rng.integers(1, 500))
        w = None if rng.random() < 0.15 else int(rng.integers(1, 700))
        h = None if rng.random() < 0.15 else int(rng.integers(1, 700))
        g = None if rng.random() < 0.5 else (float(rng.integers(-10, 120)), float(rng.integers(-10, 120)))
        mode = MODES[int(rng.integers(0, 9))]
        got = pc(mode, sw, sh, w, h, g)
        try:
            crop, scale, pad, canvas = F.process_constraint(mode, sw, sh, w, h, g)
            exp = (0, crop, tuple(scale), pad, tuple(canvas))
        except F.LayoutError:
            exp = (1, None, None, None, None)
        assert got == exp, (mode, sw, sh, w, h, g)
        if got[0] == 0:                                   # what every result must satisfy (sizing_tests.rs:711-716 add_defaults)
            _, crop, scale, pad, canvas = got
            assert canvas[0] >= scale[0] and canvas[1] >= scale[1]
            if crop:
                assert 0 <= crop[0] < crop[2] <= sw and 0 <= crop[1] < crop[3] <= sh
            if pad:
                assert (pad[0] + pad[2] + scale[0], pad[1] + pad[3] + scale[1]) == canvas
            if mode.startswith("within") and w is not None and h is not None:
                assert scale[0] <= max(sw, w) and scale[1] <= max(sh, h)
