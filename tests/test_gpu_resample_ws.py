"""The wave-specialised resample kernel (csrc/resample_ws.hip) against the CPU oracle, bit for bit.  It is NOT the product's
choice -- measured slower than the one-role kernel on every BASELINE shape (profiles/r5_ws_*) -- and runs only
when the development switch `ws` is 1; these tests keep the structure the round-4 review asked for honest and runnable.

The kernel splits a workgroup into V waves (stream, convert, vertical pass, publish rows into an LDS ring) and H waves
(horizontal pass, encode, store), synchronised by LDS counters instead of a workgroup barrier.  The arithmetic is the
one-role kernel's -- same taps, same order -- so every case below asserts the oracle's BGRA8 bytes AND its f32 working
values (0 ULP).  What the cases vary is the schedule: ring depth (1 slot = fully serialised hand-over), H waves per
workgroup (1 = every unit of every row through one wave), bands, column strips, several frames per workgroup with idle
slots, two- and four-column groups, both working spaces, a sub-rectangle of the canvas.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.graphics.bitmaps import BitmapCompositing  # noqa: E402
from imageflow_amd.graphics.color import WorkingFloatspace  # noqa: E402
from imageflow_amd.graphics.weights import Filter  # noqa: E402
from tests.test_gpu_resample import run_case  # noqa: E402

# (in_w, in_h, out_w, out_h, frames): the BASELINE shapes' horizontal geometry at reduced heights
MODERATE = [
    (3840, 216, 1600, 90, 2),      # cfg3 level 0: two column strips
    (1600, 90, 1200, 68, 3),       # level 1: two-column groups, 7 V waves
    (1600, 90, 800, 45, 2),        # level 2
    (1200, 68, 400, 23, 3),        # level 3: 5 V waves
    (1920, 108, 800, 45, 3),       # cfg4 resize: 8 V waves
    (480, 135, 200, 57, 11),       # cfg1 resize: 4 frames per workgroup, idle slots in the last one
    (333, 100, 250, 75, 5),        # ragged
    (800, 60, 333, 25, 7),
]


@pytest.mark.parametrize("case", MODERATE)
def test_ws_equals_the_oracle(case, debug_switch):
    iw, ih, ow, oh, n = case
    debug_switch("ws", "1")
    p = run_case(iw, ih, ow, oh, n=n, seed=iw + oh)
    assert p.kernel_kind(False) == 0
    run_case(iw, ih, ow, oh, n=1, seed=ow, x=3, y=2, cw=ow + 9, ch=oh + 5, space=WorkingFloatspace.StandardRGB)


def test_ws_is_off_unless_asked_for(debug_switch):
    """Without the switch the one-role kernel runs (same pixels; `trace_launch` names the kernel on stderr)."""
    run_case(1920, 108, 800, 45, n=2, seed=1)


@pytest.mark.parametrize("ring,h_waves", [("1", "1"), ("1", "8"), ("2", "3"), ("4", "2"), ("3", "16")])
def test_ws_schedules(ring, h_waves, debug_switch):
    """Hand-over extremes: one row slot (V waits for H on every row), one H wave, more slots than rows in a band."""
    debug_switch("ws", "1")
    debug_switch("ws_ring", ring)
    debug_switch("ws_h_waves", h_waves)
    run_case(1920, 108, 800, 45, n=2, seed=int(ring) * 10 + int(h_waves))
    run_case(480, 60, 200, 25, n=5, seed=3)
    debug_switch("bands", "7")
    run_case(1600, 200, 1200, 150, n=2, seed=4)


def test_ws_filters_and_rings(debug_switch):
    """Ring sizes 1..5 (vertical ratio / filter window) behind the same horizontal geometry; K = 6 and up stay one-role."""
    debug_switch("ws", "1")
    for (ih, oh, filt) in ((64, 64, Filter.Box), (200, 100, Filter.Box), (120, 64, Filter.Triangle), (300, 100, Filter.Hermite),
                           (400, 180, Filter.Robidoux), (400, 90, Filter.Robidoux), (330, 200, Filter.Lanczos)):
        run_case(640, ih, 300, oh, filt=filt, n=2, seed=ih + oh)


def test_ws_full_size_level0_frame(debug_switch):
    """One full-size frame of cfg3 level 0 (3840x2160 -> 1600x900, 900 rows through the ring), gradient + noise."""
    from tests import util as U
    debug_switch("ws", "1")
    fr = np.concatenate([U.gradient_frames(1, 3840, 2160), U.random_frames(1, 3840, 2160, seed0=5, alpha=True)])
    run_case(3840, 2160, 1600, 900, frames=fr)
