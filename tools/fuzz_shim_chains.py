#!/usr/bin/env python3
"""Differential fuzz of the job interpreter behind the libimageflow C ABI (csrc/abi_shim.cpp) against the Python node mirrors
(imageflow_amd/flow/nodes/*.py) on the GPU (round 6).

Two independent statements of the reference's node semantics exist in this repository: the C++ interpreter a job's JSON
reaches through `imageflow_context_send_json("v1/execute")`, and the Python mirrors of flow/nodes/*.rs that drive the same
`ifhip_*` entry points from tests and bench (each cites the Rust lines it follows; tests/test_node_mirrors.py,
test_gpu_bitmap_ops.py and test_gpu_abi_shim.py pin them to the oracle on hand-picked jobs).  This sweep sends random CHAINS of
nodes -- flips, transpose, rotations, apply_orientation, crop, expand_canvas, fill_rect, region, region_percent, the
colour filters, resample_2d with random hints (filters, sharpen + sharpen_when, resample_when, colour space, background
colour) -- on a random raw frame (alpha meaningful or not) through both and compares the final pixels, size and alpha flag.
A quarter of the cases are two-input GRAPHS: a canvas side (a decoded frame or a create_canvas node, followed by its own
chain) and an input side joined by draw_image_exact (random rect, blend, hints) or copy_rect_to_canvas, then a tail chain;
chains also draw color_matrix_srgb with a random matrix and watermark nodes (second input; fit box, fit mode, gravity,
opacity; sizes by a Python restatement of imageflow_riapi's sizing.rs:118-197).  A third of the sources are baseline JPEG
files (Pillow-written: quality, 4:4:4 / 4:2:2 / 4:2:0, grey) -- in a job they stay coefficients until a node needs pixels
(or are decoded and resampled in one call, or decoded at i/8 under jpeg_downscale_hints); the mirror decodes them with
codecs.mozjpeg_decoder.decode_frames first.  A quarter of the jobs end in a JPEG file (encode preset libjpeg_turbo: quality,
progressive, optimised tables, matte) instead of the raw frame: the file must equal, byte for byte, the one libjpeg-turbo
(Pillow) writes from the mirror's final pixels flattened onto the matte (codecs/mozjpeg.rs:88-94).
What it can find: state that one side carries from node to node and the other does not (alpha_meaningful, the canvas'
compositing mode, matte colours, windows with a foreign stride), and work queued on the wrong stream.  Found in round 6:
fill_rect launched on the null stream behind a colour filter on the job's stream; resample_2d clearing alpha_meaningful
after an opaque matte where the reference keeps the canvas Bgra32.

    python tools/fuzz_shim_chains.py [--seconds 300] [--chains N] [--seed 1] [--threads T] [--out gpurun_out/fuzz_shim.jsonl]

tests/test_gpu_shim_chain_fuzz.py runs a fixed number of chains of this sweep in the GPU suite."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FILTERS = ["robidoux", "robidoux_sharp", "robidoux_fast", "ginseng", "lanczos", "lanczos_2", "cubic", "catmull_rom", "mitchell", "hermite",
           "triangle", "box", "n_cubic", "fastest", "jinc", "cubic_b_spline"]
NAMED = ["sepia", "grayscale_ntsc", "grayscale_ry", "grayscale_flat", "grayscale_bt709", "invert"]
PARAM = ["alpha", "contrast", "saturation", "brightness"]


def rand_color(rng):
    k = int(rng.integers(0, 5))
    if k == 0:
        return "transparent"
    a = [0x00, 0xFF, 0xFF, int(rng.integers(1, 255)), 0x80][int(rng.integers(0, 5))]
    return {"srgb": {"hex": "%02X%02X%02X%02X" % (int(rng.integers(0, 256)), int(rng.integers(0, 256)), int(rng.integers(0, 256)), a)}}


def color32_of(c, color32):
    if c == "transparent":
        return 0
    return color32(c["srgb"]["hex"])


def rand_hints(rng, allow_when=True):
    hints = {}
    if rng.random() < 0.5:
        hints["down_filter"] = FILTERS[int(rng.integers(0, len(FILTERS)))]
    if rng.random() < 0.5:
        hints["up_filter"] = FILTERS[int(rng.integers(0, len(FILTERS)))]
    if rng.random() < 0.5:
        hints["sharpen_percent"] = float(rng.choice([0.0, 10.0, 35.0, 100.0]))
    if rng.random() < 0.4:
        hints["sharpen_when"] = ["downscaling", "upscaling", "size_differs", "always"][int(rng.integers(0, 4))]
    if allow_when and rng.random() < 0.4:
        hints["resample_when"] = ["size_differs", "size_differs_or_sharpening_requested", "always"][int(rng.integers(0, 3))]
    if rng.random() < 0.4:
        hints["scaling_colorspace"] = ["srgb", "linear"][int(rng.integers(0, 2))]
    if rng.random() < 0.4:
        hints["background_color"] = rand_color(rng)
    return hints


def draw_node(rng, w, h, mark=None):
    """-> (json node, new (w, h) estimate) for a frame of w x h; `mark` = (w, h) of the watermark input when the job has one"""
    k = int(rng.integers(0, 19))
    if k == 0:
        return "flip_h", (w, h)
    if k == 1:
        return "flip_v", (w, h)
    if k == 2:
        return "transpose", (h, w)
    if k == 3:
        r = ["rotate_90", "rotate_180", "rotate_270"][int(rng.integers(0, 3))]
        return r, ((w, h) if r == "rotate_180" else (h, w))
    if k == 4:
        f = int(rng.integers(0, 10))
        return {"apply_orientation": {"flag": f}}, ((h, w) if 5 <= f <= 8 else (w, h))
    if k == 5 and w > 1 and h > 1:
        x1, y1 = int(rng.integers(0, w - 1)), int(rng.integers(0, h - 1))
        x2, y2 = int(rng.integers(x1 + 1, w + 1)), int(rng.integers(y1 + 1, h + 1))
        return {"crop": {"x1": x1, "y1": y1, "x2": x2, "y2": y2}}, (x2 - x1, y2 - y1)
    if k == 6:
        l, t, r, b = (int(v) for v in rng.integers(0, 24, 4))
        return {"expand_canvas": {"left": l, "top": t, "right": r, "bottom": b, "color": rand_color(rng)}}, (w + l + r, h + t + b)
    if k == 7:
        x1, y1 = int(rng.integers(0, w)), int(rng.integers(0, h))
        x2, y2 = int(rng.integers(x1 + 1, w + 1)), int(rng.integers(y1 + 1, h + 1))
        if rng.random() < 0.04:
            x2 = x1                                       # an empty rectangle: the node refuses it (clone_crop_fill_expand.rs:114-127)
        if rng.random() < 0.03:
            x2 = w + 1 + int(rng.integers(0, 5))          # outside the frame
        return {"fill_rect": {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "color": rand_color(rng)}}, (w, h)
    if k == 8:
        x1, y1 = int(rng.integers(-30, w + 10)), int(rng.integers(-30, h + 10))
        x2, y2 = x1 + int(rng.integers(1, w + 40)), y1 + int(rng.integers(1, h + 40))
        nw, nh = x2 - x1, y2 - y1
        return {"region": {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "background_color": rand_color(rng)}}, (nw, nh)
    if k == 9:
        x1, y1 = float(rng.integers(-20, 80)), float(rng.integers(-20, 80))
        x2, y2 = x1 + float(rng.integers(5, 90)) + 0.5 * int(rng.integers(0, 2)), y1 + float(rng.integers(5, 90))
        return {"region_percent": {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "background_color": rand_color(rng)}}, None
    if k == 10:
        return {"color_filter_srgb": NAMED[int(rng.integers(0, len(NAMED)))]}, (w, h)
    if k == 11:
        name = PARAM[int(rng.integers(0, len(PARAM)))]
        return {"color_filter_srgb": {name: float(np.float32(rng.uniform(-0.9, 0.9) if name != "alpha" else rng.uniform(0.05, 1.0)))}}, (w, h)
    if k == 12:
        m = np.eye(5, dtype=np.float32) + (rng.uniform(-0.6, 0.6, (5, 5)) * (rng.random((5, 5)) < 0.4)).astype(np.float32)
        return {"color_matrix_srgb": {"matrix": [[float(v) for v in row] for row in m]}}, (w, h)
    if k == 13 and mark is not None:
        wm = {"io_id": 2}
        fb = int(rng.integers(0, 3))
        if fb == 1:
            wm["fit_box"] = {"image_percentage": {"x1": float(rng.integers(-5, 60)), "y1": float(rng.integers(-5, 60)),
                                                  "x2": float(rng.integers(40, 110)), "y2": float(rng.integers(40, 110))}}
        elif fb == 2:
            wm["fit_box"] = {"image_margins": {"left": int(rng.integers(0, 1 + w // 2)), "top": int(rng.integers(0, 1 + h // 2)),
                                               "right": int(rng.integers(0, 1 + w // 2)), "bottom": int(rng.integers(0, 1 + h // 2))}}
        if rng.random() < 0.7:
            wm["fit_mode"] = ["within", "fit", "distort", "within_crop", "fit_crop"][int(rng.integers(0, 5))]
        if rng.random() < 0.6:
            wm["gravity"] = {"percentage": {"x": float(rng.integers(-10, 120)), "y": float(rng.integers(-10, 120))}}
        if rng.random() < 0.6:
            wm["opacity"] = float(np.float32(rng.uniform(-0.1, 1.2)))
        if rng.random() < 0.15:
            wm["min_canvas_width"] = int(rng.integers(0, 200))
        if rng.random() < 0.15:
            wm["min_canvas_height"] = int(rng.integers(0, 150))
        return {"watermark": wm}, (w, h)
    if k in (14, 15):
        mode = ["distort", "within", "fit", "larger_than", "within_crop", "fit_crop", "aspect_crop", "within_pad", "fit_pad"][int(rng.integers(0, 9))]
        c = {"mode": mode}
        if rng.random() < 0.85:
            c["w"] = max(1, int(w * rng.uniform(0.2, 2.0)))
        if rng.random() < 0.85:
            c["h"] = max(1, int(h * rng.uniform(0.2, 2.0)))
        if rng.random() < 0.5:
            c["hints"] = rand_hints(rng)
        if rng.random() < 0.4:
            c["gravity"] = {"percentage": {"x": float(rng.integers(-10, 120)), "y": float(rng.integers(-10, 120))}} if rng.random() < 0.8 else "center"
        if rng.random() < 0.4:
            c["canvas_color"] = rand_color(rng)
        return {"constrain": c}, None               # (the size is the layout engine's: the chain ends here)
    # resample_2d (more likely than any other node)
    same = rng.random() < 0.2
    ow = w if same else max(1, int(w * rng.uniform(0.2, 2.2)))
    oh = h if same else max(1, int(h * rng.uniform(0.2, 2.2)))
    return {"resample_2d": {"w": ow, "h": oh, "hints": rand_hints(rng)}}, (ow, oh)


def draw_chain(rng, w, h, max_nodes, mark=None):
    nodes = []
    for _ in range(max_nodes):
        node, size = draw_node(rng, w, h, mark)
        nodes.append(node)
        if size is None or size[0] < 1 or size[1] < 1 or size[0] * size[1] > 400_000:
            return nodes, None
        w, h = size
    return nodes, (w, h)


# ---- imageflow_riapi restated for the constrain / watermark nodes: sizing.rs (AspectRatio :35-222, Layout :304-450) and
# ---- ir4/layout.rs (step programs :160-283, process_constraint :334-412, gravity :673-698)
class LayoutError(ValueError):
    pass


def _rround(v):                                     # f64::round / f32::round: half away from zero
    import math
    return math.copysign(math.floor(abs(v) + 0.5), v)


def _create(w, h):
    if w < 1 or h < 1:
        raise LayoutError(f"InvalidDimensions {w}x{h}")
    return (int(w), int(h))


def _proportional(sw, sh, basis, basis_is_width, target=None):
    ratio = sw / sh
    snap_amount = 1.0 - 2.220446049250313e-16
    if target is not None:
        if basis_is_width:                          # rounding_loss_based_on_target_height(target.h)  (:100-115)
            snap_amount = abs(target[1] - _rround(sw * (target[1] / sh)) / ratio)
        else:                                       # rounding_loss_based_on_target_width(target.w)   (:81-97)
            snap_amount = abs(target[0] - _rround(sh * (target[0] / sw)) * ratio)
    snap_a = sh if basis_is_width else sw
    snap_b = snap_a if target is None else (target[1] if basis_is_width else target[0])
    f = basis / ratio if basis_is_width else ratio * basis
    da, db = f - snap_a, f - snap_b
    if abs(da) <= snap_amount and abs(da) <= abs(db):
        v = snap_a
    elif abs(db) <= snap_amount:
        v = snap_b
    else:
        v = int(_rround(f))
    if v < 0:
        raise LayoutError("ValueScalingFailed")
    return max(v, 1)


def _box_of(self, target, inner):                   # AspectRatio::box_of (:185-193)
    self_wider = self[0] / self[1] > target[0] / target[1]          # target.aspect_wider_than(self)
    if self_wider == inner:
        return _create(target[0], _proportional(self[0], self[1], target[0], True, target))
    return _create(_proportional(self[0], self[1], target[1], False, target), target[1])


def _inner_box(sw, sh, tw, th):
    return _box_of((sw, sh), (tw, th), True)


def _gravity1d(pct, inner, outer):                  # ir4/layout.rs:673-683
    f32 = np.float32
    if (outer < inner and inner < 1) or outer < 1:
        raise LayoutError("gravity")
    v = int(_rround(float(f32(outer - inner) * (f32(min(max(pct, 0.0), 100.0)) / f32(100)))))
    return max(0, min(v, outer - inner))


def process_constraint(mode, sw, sh, w, h, gravity=None):
    """-> (crop [x1, y1, x2, y2] or None, scale_to (w, h), pad [l, t, r, b] or None, canvas (w, h)); w / h None: not given"""
    initial = _create(sw, sh)
    some_w, some_h = w is not None and w >= 1, h is not None and h >= 1
    if some_w and some_h:
        target = _create(w, h)
    elif some_w:
        target = _create(w, _proportional(sw, sh, w, True))
    elif some_h:
        target = _create(_proportional(sw, sh, h, False), h)
    else:
        target = initial
    fit, scale = {"distort": ("stretch", "both"), "within": ("max", "down"), "fit": ("max", "both"), "larger_than": ("max", "up"),
                  "within_crop": ("crop", "down"), "fit_crop": ("crop", "both"), "aspect_crop": ("aspect", "down"),
                  "within_pad": ("pad", "down"), "fit_pad": ("pad", "both")}[mode]
    if w is None and h is None:
        fit = "max"
    L = {"source": initial, "canvas": initial, "image": initial}

    def cmp():
        return ((L["canvas"][0] > target[0]) - (L["canvas"][0] < target[0]), (L["canvas"][1] > target[1]) - (L["canvas"][1] < target[1]))

    def distort_with(s, old, new):
        return _create(s[0] * new[0] // old[0], s[1] * new[1] // old[1])

    def scale_canvas(inner):
        nc = _box_of(L["canvas"], target, inner)
        L["image"] = distort_with(L["image"], L["canvas"], nc)
        L["canvas"] = nc

    def crop(t):
        if t[0] > L["canvas"][0] or t[1] > L["canvas"][1]:
            raise LayoutError("ImpossibleCrop")
        ni = _create(min(L["image"][0], t[0]), min(L["image"][1], t[1]))
        L["source"] = _box_of(ni, L["source"], True)
        L["image"], L["canvas"] = ni, t
    gate = scale == "both" or (scale == "down" and 1 in cmp()) or (scale == "up" and 1 not in cmp())
    if fit == "max":
        if gate:
            scale_canvas(True)
    elif fit == "pad":
        if gate:
            scale_canvas(True)
            if L["canvas"][0] > target[0] or L["canvas"][1] > target[1]:
                raise LayoutError("ImpossiblePad")
            L["canvas"] = target
    elif fit == "stretch":
        if gate:
            L["image"] = distort_with(L["image"], L["canvas"], target)
            L["canvas"] = target
    elif fit == "crop":
        if scale == "both":
            scale_canvas(False)
            crop(target)
        else:
            if -1 not in cmp():
                scale_canvas(False)
                crop(target)
            if cmp() in ((1, -1), (-1, 1)):
                crop(_create(min(L["image"][0], target[0]), min(L["image"][1], target[1])))
    else:
        crop(_box_of(target, L["canvas"], True))
    gx, gy = gravity if gravity is not None else (50.0, 50.0)
    src = L["source"]
    cx, cy = _gravity1d(gx, src[0], initial[0]), _gravity1d(gy, src[1], initial[1])
    out_crop = [cx, cy, cx + src[0], cy + src[1]] if (cx > 0 or cy > 0 or src != initial) else None
    img, cv = L["image"], L["canvas"]
    left, top = _gravity1d(gx, img[0], cv[0]), _gravity1d(gy, img[1], cv[1])
    right, bottom = cv[0] - img[0] - left, cv[1] - img[1] - top
    pad = [left, top, right, bottom] if max(left, top, right, bottom) > 0 else None
    if pad and min(pad) < 0:
        raise LayoutError("negative padding")
    return out_crop, img, pad, cv


def watermark_placement(cw, ch, mw, mh, wm):
    """-> None (the node disappears) or (x, y, w, h, crop of the mark or None); flow/nodes/watermark.rs:11-86, :109-150"""
    f32 = np.float32
    if not (wm.get("min_canvas_width", 0) < cw and wm.get("min_canvas_height", 0) < ch):
        return None
    box = (0, 0, cw, ch)
    fb = wm.get("fit_box")
    if fb and "image_percentage" in fb:
        p = fb["image_percentage"]
        box = tuple(int(_rround(float(f32(min(max(p[k], 0.0), 100.0)) / f32(100) * f32(s)))) for k, s in (("x1", cw), ("y1", ch), ("x2", cw), ("y2", ch)))
        if not (box[0] < box[2] and box[1] < box[3]):
            return None
    elif fb:
        m = fb["image_margins"]
        if not (m["left"] + m["right"] < cw and m["top"] + m["bottom"] < ch):
            return None
        box = (m["left"], m["top"], cw - m["right"], ch - m["bottom"])
    bw, bh = box[2] - box[0], box[3] - box[1]
    g = wm.get("gravity", {"percentage": {"x": 50.0, "y": 50.0}})["percentage"]
    crop, (tw, th), _, _ = process_constraint(wm.get("fit_mode", "within"), mw, mh, bw, bh, (g["x"], g["y"]) if "gravity" in wm else None)

    def g1(pct, inner, outer):                      # WatermarkDef::gravity1d (:60-67): no clamp of the result
        if (outer < inner and inner < 1) or outer < 1:
            raise LayoutError("Watermark fit_box does not work")
        return int(_rround(float(f32(outer - inner) * (f32(min(max(pct, 0.0), 100.0)) / f32(100)))))
    return g1(g["x"], tw, bw) + box[0], g1(g["y"], th, bh) + box[1], tw, th, crop


def hints_of(hints, M):
    CC, RF, SR, CO, Bm, color32, JSON_FILTER_NAMES, WorkingFloatspace = M[:8]
    return SR.ResampleHints(sharpen_percent=hints.get("sharpen_percent"),
                            down_filter=JSON_FILTER_NAMES[hints["down_filter"]] if "down_filter" in hints else None,
                            up_filter=JSON_FILTER_NAMES[hints["up_filter"]] if "up_filter" in hints else None,
                            scaling_colorspace=(WorkingFloatspace.StandardRGB if hints["scaling_colorspace"] == "srgb" else WorkingFloatspace.LinearRGB)
                            if "scaling_colorspace" in hints else None,
                            sharpen_when=SR.SharpenWhen(hints["sharpen_when"]) if "sharpen_when" in hints else None,
                            resample_when=hints.get("resample_when"))


def canvas_of(M, n, w, h, device, color, bgra32):
    """CreateCanvasDef::execute (create_canvas.rs:77-103): only the enum value Transparent makes a ReplaceSelf canvas; any srgb
    colour -- also one whose alpha is 0 -- is a BlendWithMatte canvas (pre-filled unless that colour is transparent,
    bitmaps.rs:829-837)"""
    Bm, color32 = M[4], M[5]
    enum_transparent = color in (None, "transparent")
    c32 = 0 if enum_transparent else color32(color["srgb"]["hex"])
    return Bm.Bitmap.create_u8(n, w, h, device, alpha_meaningful=bgra32,
                               compose=Bm.BitmapCompositing.ReplaceSelf if enum_transparent else Bm.BitmapCompositing.BlendWithMatte, matte=c32), c32


def mirror_apply(b, node, M, mark=None):
    """the node on the Python mirrors -> the resulting Bitmap"""
    CC, RF, SR, CO, Bm, color32, JSON_FILTER_NAMES, WorkingFloatspace = M[:8]
    if isinstance(node, str):
        return getattr(RF, node)(b)
    (name, p), = node.items()
    if name == "apply_orientation":
        return RF.apply_orientation(b, p["flag"])
    if name == "crop":
        return CC.crop(b, p["x1"], p["y1"], p["x2"], p["y2"])
    if name == "expand_canvas":
        return CC.expand_canvas(b, p["left"], p["top"], p["right"], p["bottom"], color32_of(p["color"], color32), p["color"] == "transparent")
    if name == "fill_rect":
        return CC.fill_rect(b, p["x1"], p["y1"], p["x2"], p["y2"], color32_of(p["color"], color32))
    if name == "region":
        return CC.region(b, p["x1"], p["y1"], p["x2"], p["y2"], color32_of(p["background_color"], color32), p["background_color"] == "transparent")
    if name == "region_percent":
        return CC.region_percent(b, p["x1"], p["y1"], p["x2"], p["y2"], color32_of(p["background_color"], color32), p["background_color"] == "transparent")
    if name == "color_filter_srgb":
        if isinstance(p, str):
            CO.color_filter_srgb(b, p)
        else:
            (fname, v), = p.items()
            CO.color_filter_srgb(b, fname, v)
        return b
    if name == "color_matrix_srgb":
        CO.color_matrix_srgb(b, np.asarray(p["matrix"], dtype=np.float32))
        return b
    if name == "watermark":
        WM = M[8]
        mark_src, mw, mh = mark
        place = watermark_placement(b.w, b.h, mw, mh, p)
        if place is None:
            return b
        x, y, w, h, mcrop = place
        if x < 0 or y < 0:
            raise M[9](M[10].InvalidArgument, "Watermark fit_box does not work")
        m = Bm.Bitmap.from_numpy(mark_src[None].copy(), mw, mh, mark_src.shape[1], b.data.device, alpha_meaningful=True)
        WM.draw_watermark(b, m, x, y, w, h, opacity=p.get("opacity"), crop=mcrop)
        return b
    if name == "constrain":
        # ConstrainDef::expand (constrain.rs:41-98): [Crop] -> Resample2D (canvas_color over hints.background_color) -> [ExpandCanvas]
        g = p["gravity"]["percentage"] if isinstance(p.get("gravity"), dict) else None
        crop, (sw, sh), pad, _ = process_constraint(p["mode"], b.w, b.h, p.get("w"), p.get("h"), (g["x"], g["y"]) if g else None)
        if crop:
            b = CC.crop(b, *crop)
        hints = dict(p.get("hints") or {})
        if p.get("canvas_color") is not None:
            hints["background_color"] = p["canvas_color"]
        b = mirror_apply(b, {"resample_2d": {"w": sw, "h": sh, "hints": hints}}, M)
        if pad:
            cc = p.get("canvas_color")
            b = CC.expand_canvas(b, pad[0], pad[1], pad[2], pad[3], color32_of(cc, color32) if cc is not None else 0, cc is None or cc == "transparent")
        return b
    if name == "resample_2d":
        # Scale2dDef / the Resample2D expansion, scale_render.rs:30-120, then DrawImageExact :139-201
        w, h, hints = p["w"], p["h"], p.get("hints") or {}
        when = hints.get("resample_when", "size_differs_or_sharpening_requested")
        size_differs = w != b.w or h != b.h
        downscaling = w < b.w or h < b.h
        upscaling = w > b.w or h > b.h
        bg = hints.get("background_color")
        apply_matte = b.alpha_meaningful and bg is not None and bg != "transparent"
        raw = hints.get("sharpen_percent", 0.0) or 0.0
        sw = hints.get("sharpen_when", "always")
        sharpen = raw if (sw == "always" or (sw == "downscaling" and downscaling) or (sw == "upscaling" and upscaling)
                          or (sw == "size_differs" and size_differs)) else 0.0
        resample = when == "always" or (when == "size_differs" and size_differs) or \
            (when == "size_differs_or_sharpening_requested" and (size_differs or sharpen != 0.0))
        if not (resample or apply_matte):
            return b
        canvas, c32 = canvas_of(M, b.n, w, h, b.data.device, bg, b.alpha_meaningful)
        rh = hints_of({k: v for k, v in hints.items() if k != "resample_when"}, M)
        rh.sharpen_percent = sharpen
        transparent_bg = (c32 >> 24) == 0                     # Color::is_transparent(): by the colour's alpha (imageflow_types lib.rs:852)
        SR.render(canvas, b, 0, 0, w, h, rh, SR.CompositingMode.Overwrite if transparent_bg else None)
        return canvas
    raise ValueError(name)


def mirror_join(canvas, inp, node, M):
    CC, SR = M[0], M[2]
    (name, p), = node.items()
    if name == "draw_image_exact":
        blend = {"compose": SR.CompositingMode.Compose, "overwrite": SR.CompositingMode.Overwrite, None: None}[p.get("blend")]
        SR.render(canvas, inp, p["x"], p["y"], p["w"], p["h"], hints_of(p.get("hints") or {}, M), blend)
        return canvas
    if name == "copy_rect_to_canvas":
        return CC.copy_rect_to_canvas(inp, canvas, p["from_x"], p["from_y"], p["w"], p["h"], p["x"], p["y"])
    raise ValueError(name)


def rand_jpeg(rng, w, h):
    spec = {"quality": int(rng.integers(20, 98)), "subsampling": int(rng.integers(0, 3)), "grey": bool(rng.random() < 0.15),
            "smooth": bool(rng.random() < 0.5)}
    if rng.random() < 0.08:
        # a DAMAGED scan (a few bytes of the entropy-coded data overwritten): the file decodes to something or is refused -- the
        # same way in a job (where it may sit in an entropy batch with other jobs' files, which must not notice) as alone
        spec["damage"] = [int(rng.integers(0, 1 << 30)), int(rng.integers(1, 4))]
    elif rng.random() < 0.2:
        # a PROGRESSIVE file (what the reference's own mozjpeg preset writes): the job's scans are decoded on the host
        # (csrc/jpeg_read.cpp); the mirror gets the baseline twin libjpeg-turbo writes from the same pixels -- the same coefficients
        spec["progressive"] = True
    if rng.random() < 0.5:                              # s::DecoderCommand::JpegDownscaleHints on the decode node
        spec["hints"] = {"width": max(1, int(w * rng.uniform(0.05, 1.2))), "height": max(1, int(h * rng.uniform(0.05, 1.2)))}
        if rng.random() < 0.6:
            spec["hints"]["scale_luma_spatially"] = bool(rng.integers(0, 2))
        if rng.random() < 0.6:
            spec["hints"]["gamma_correct_for_srgb_during_spatial_luma_scaling"] = bool(rng.integers(0, 2))
    return spec


def decode_node(io_id, spec):
    if spec and "hints" in spec:
        return {"decode": {"io_id": io_id, "commands": [{"jpeg_downscale_hints": spec["hints"]}]}}
    return {"decode": {"io_id": io_id}}


def hinted_size(w, h, spec):
    """MzDec::apply_downscaling with the hints as tell_decoder maps them (mozjpeg_decoder.rs:72-77, :588-618) -> (scale_num, w, h)"""
    hw, hh = (spec["hints"]["width"], spec["hints"]["height"]) if spec and "hints" in spec else (0, 0)
    if hw > 0 and hh > 0 and (w > hw or h > hh):
        for i in (1, 2, 3, 4, 5, 6):
            nw, nh = -(-w * i // 8), -(-h * i // 8)
            if nw >= hw and nh >= hh:
                return i, nw, nh
    return 8, w, h


def jpeg_bytes(src, w, h, spec, progressive=False):
    """the frame as a baseline (or progressive) JPEG file (Pillow = libjpeg-turbo); smooth: a low-pass of the noise, so that
    files with long zero runs and small DC differences appear too"""
    import io

    from PIL import Image, ImageFilter
    rgb = np.ascontiguousarray(src[:, :4 * w].reshape(h, w, 4)[:, :, 2::-1])
    im = Image.fromarray(rgb, "RGB")
    if spec["smooth"]:
        im = im.filter(ImageFilter.GaussianBlur(2.0))
    if spec["grey"]:
        im = im.convert("L")
    f = io.BytesIO()
    kw = {} if spec["grey"] else {"subsampling": spec["subsampling"]}
    im.save(f, "JPEG", quality=spec["quality"], progressive=progressive, **kw)
    data = f.getvalue()
    if "damage" in spec:
        sos = data.find(b"\xff\xda")
        first = sos + 2 + int.from_bytes(data[sos + 2:sos + 4], "big")          # the first byte of entropy-coded data
        if first < len(data) - 3:
            r = np.random.default_rng(spec["damage"][0])
            b = bytearray(data)
            for _ in range(spec["damage"][1]):
                b[int(r.integers(first, len(data) - 2))] = int(r.integers(0, 256))
            data = bytes(b)
    return data


def rand_encode(rng):
    enc = {}
    if rng.random() < 0.8:
        enc["quality"] = int(rng.integers(1, 101)) if rng.random() < 0.93 else int(rng.choice([0, 101, 150, 255, 256, 300, -1, -5, -200]))
    if rng.random() < 0.3:
        enc["progressive"] = bool(rng.integers(0, 2))
    if rng.random() < 0.3:
        enc["optimize_huffman_coding"] = bool(rng.integers(0, 2))
    if rng.random() < 0.4:
        enc["matte"] = rand_color(rng)
    return enc


JPEG_POOL = [(64, 48, 2), (97, 61, 2), (160, 120, 0), (33, 100, 1)]       # (w, h, subsampling): geometries concurrent jobs share


def draw_case(rng, pool_jpeg=False):
    """-> the case as plain JSON data: sizes, seeds, node lists.  pool_jpeg: most JPEG sources take one of four geometries,
    so that the decode coalescer finds files of concurrent jobs it can put into one entropy batch"""
    w, h = int(rng.integers(1, 180)), int(rng.integers(1, 130))
    pooled = None
    if pool_jpeg and rng.random() < 0.7:
        pooled = JPEG_POOL[int(rng.integers(0, len(JPEG_POOL)))]
    case = {"size": [w, h], "alpha": bool(rng.integers(0, 2)), "seed": int(rng.integers(0, 1 << 30)),
            "mark": [int(rng.integers(1, 70)), int(rng.integers(1, 50)), int(rng.integers(0, 1 << 30))]}
    if rng.random() < (0.6 if pool_jpeg else 0.33):
        if pooled:
            w, h = pooled[:2]
            case["size"] = [w, h]
        case["jpeg"] = rand_jpeg(rng, w, h)
        if pooled:
            case["jpeg"]["subsampling"], case["jpeg"]["grey"] = pooled[2], False
        case["alpha"] = False
        _, w, h = hinted_size(w, h, case["jpeg"])
    if rng.random() < 0.12:
        # a TREE (the shape of export_4_sizes): source -> trunk -> a parent several branches read, each with its own chain and
        # output.  The parent is shared: MutProtect puts a Clone before a mutating first node (definitions.rs:320-341); the
        # nodes that would change a shared parent in place in the reference (fill_rect, a watermark drawn onto it, a JPEG
        # encoder's matte) are kept off the branches' first position / the parent's own output -- there the reference's result
        # depends on the order the engine happens to run the siblings in.
        case["tree"] = True
        case["nodes"], size = draw_chain(rng, w, h, int(rng.integers(0, 3)))
        while case["nodes"] and isinstance(case["nodes"][-1], dict) and "constrain" in case["nodes"][-1]:
            case["nodes"].pop()                           # (a constrain ends a chain without a size estimate: keep the trunk sized)
            size = None
        if size is None:
            case["nodes"], size = [], (w, h)
        case["parent_output"] = bool(rng.random() < 0.5)
        case["branches"] = []
        for _ in range(int(rng.integers(2, 4))):
            while True:
                chain, _ = draw_chain(rng, size[0], size[1], int(rng.integers(0, 4)), mark=case["mark"][:2])
                first = chain[0] if chain else None
                if not (isinstance(first, dict) and ("fill_rect" in first or "watermark" in first)):
                    break
            br = {"chain": chain}
            if chain and rng.random() < 0.4:
                br["encode"] = rand_encode(rng)
            case["branches"].append(br)
        return case
    if rng.random() < 0.25:
        case["encode"] = rand_encode(rng)
    if rng.random() >= 0.25:
        case["nodes"], _ = draw_chain(rng, w, h, int(rng.integers(1, 7)), mark=case["mark"][:2])
        return case
    # two inputs: the canvas side (decode of input 0 or create_canvas) and the input side (decode of input 1)
    if rng.random() < 0.4:
        cw, ch = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        case["canvas_node"] = {"create_canvas": {"w": cw, "h": ch, "format": ["bgra_32", "bgr_32"][int(rng.integers(0, 2))], "color": rand_color(rng)}}
    else:
        cw, ch = w, h
    case["canvas_chain"], csize = draw_chain(rng, cw, ch, int(rng.integers(0, 3)))
    iw, ih = int(rng.integers(1, 180)), int(rng.integers(1, 130))
    case["input"] = {"size": [iw, ih], "alpha": bool(rng.integers(0, 2)), "seed": int(rng.integers(0, 1 << 30))}
    if rng.random() < 0.33:
        case["input"]["jpeg"] = rand_jpeg(rng, iw, ih)
        case["input"]["alpha"] = False
        _, iw, ih = hinted_size(iw, ih, case["input"]["jpeg"])
    case["input_chain"], isize = draw_chain(rng, iw, ih, int(rng.integers(0, 3)))
    cw, ch = csize or (cw, ch)
    iw, ih = isize or (iw, ih)
    if rng.random() < 0.6:
        rw, rh = int(rng.integers(1, cw + 1 + (cw // 8))), int(rng.integers(1, ch + 1 + (ch // 8)))     # now and then too big
        x, y = int(rng.integers(0, max(1, cw - rw + 2))), int(rng.integers(0, max(1, ch - rh + 2)))
        d = {"x": x, "y": y, "w": rw, "h": rh, "hints": {k: v for k, v in rand_hints(rng, allow_when=rng.random() < 0.2).items() if k != "background_color"}}
        if rng.random() < 0.7:
            d["blend"] = ["compose", "overwrite"][int(rng.integers(0, 2))]
        case["join"] = {"draw_image_exact": d}
    else:
        rw, rh = int(rng.integers(1, min(iw, cw) + 2)), int(rng.integers(1, min(ih, ch) + 2))
        case["join"] = {"copy_rect_to_canvas": {"from_x": int(rng.integers(0, max(1, iw - rw + 2))), "from_y": int(rng.integers(0, max(1, ih - rh + 2))),
                                                 "w": rw, "h": rh, "x": int(rng.integers(0, max(1, cw - rw + 2))), "y": int(rng.integers(0, max(1, ch - rh + 2)))}}
    case["tail"], _ = draw_chain(rng, cw, ch, int(rng.integers(0, 3)))
    return case


def job_of(case):
    """the JSON of v1/execute for a case"""
    if case.get("tree"):
        nodes, edges = {"0": decode_node(0, case.get("jpeg"))}, []
        for n in case["nodes"]:
            nodes[str(len(nodes))] = n
            edges.append({"from": len(nodes) - 2, "to": len(nodes) - 1, "kind": "input"})
        parent = len(nodes) - 1
        if case["parent_output"]:
            nodes[str(len(nodes))] = {"encode": {"io_id": 9, "preset": "gif"}}
            edges.append({"from": parent, "to": len(nodes) - 1, "kind": "input"})
        for k, br in enumerate(case["branches"]):
            prev = parent
            for n in br["chain"] + [{"encode": {"io_id": 10 + k, "preset": {"libjpeg_turbo": br["encode"]} if "encode" in br else "gif"}}]:
                nodes[str(len(nodes))] = n
                edges.append({"from": prev, "to": len(nodes) - 1, "kind": "input"})
                prev = len(nodes) - 1
        return {"framewise": {"graph": {"nodes": nodes, "edges": edges}}}
    enc = {"encode": {"io_id": 9, "preset": {"libjpeg_turbo": case["encode"]} if "encode" in case else "gif"}}
    if "join" not in case:
        return {"framewise": {"steps": [decode_node(0, case.get("jpeg"))] + case["nodes"] + [enc]}}
    nodes, edges = {}, []

    def chain(first, rest):
        nodes[str(len(nodes))] = first
        for n in rest:
            nodes[str(len(nodes))] = n
            edges.append({"from": len(nodes) - 2, "to": len(nodes) - 1, "kind": "input"})
        return len(nodes) - 1
    c_end = chain(case.get("canvas_node") or decode_node(0, case.get("jpeg")), case["canvas_chain"])
    i_end = chain(decode_node(1, case["input"].get("jpeg")), case["input_chain"])
    nodes[str(len(nodes))] = case["join"]
    j = len(nodes) - 1
    edges += [{"from": i_end, "to": j, "kind": "input"}, {"from": c_end, "to": j, "kind": "canvas"}]
    for n in case["tail"] + [enc]:
        nodes[str(len(nodes))] = n
        edges.append({"from": len(nodes) - 2, "to": len(nodes) - 1, "kind": "input"})
    return {"framewise": {"graph": {"nodes": nodes, "edges": edges}}}


def case_inputs(case, E):
    """the source frames and files of a case"""
    U = E[5]
    w, h = case["size"]
    src = U.random_frames(1, w, h, seed0=case["seed"], alpha=True)[0]
    mw, mh, mseed = case["mark"]
    inp = {"src": src, "mark": (U.random_frames(1, mw, mh, seed0=mseed, alpha=True)[0], mw, mh),
           "file0": jpeg_bytes(src, w, h, case["jpeg"]) if "jpeg" in case else None, "isrc": None, "file1": None}
    inp["job_file0"] = jpeg_bytes(src, w, h, case["jpeg"], True) if case.get("jpeg", {}).get("progressive") else inp["file0"]
    if "input" in case:
        iw, ih = case["input"]["size"]
        inp["isrc"] = U.random_frames(1, iw, ih, seed0=case["input"]["seed"], alpha=True)[0]
        inp["file1"] = jpeg_bytes(inp["isrc"], iw, ih, case["input"]["jpeg"]) if "jpeg" in case["input"] else None
    inp["job_file1"] = jpeg_bytes(inp["isrc"], iw, ih, case["input"]["jpeg"], True) if case.get("input", {}).get("jpeg", {}).get("progressive") else inp["file1"]
    return inp


COUNTS = {"coalesced_decodes": 0, "fused_decode_resamples": 0, "device_coded_files": 0}     # the contexts' diagnostics, summed


def run_shim(case, inp, E):
    """the job through the C ABI -> (error text or None, the output: file bytes, or (pixels, w, h, alpha flag))"""
    Context, pack_raw_bgra, unpack_raw_bgra = E[1:4]
    w, h = case["size"]
    mark_src, mw, mh = inp["mark"]
    with Context() as c:
        c.add_input_buffer(0, inp["job_file0"] if inp["file0"] is not None else pack_raw_bgra(inp["src"], w, h, alpha_meaningful=case["alpha"]))
        if "input" in case:
            iw, ih = case["input"]["size"]
            c.add_input_buffer(1, inp["job_file1"] if inp["file1"] is not None else pack_raw_bgra(inp["isrc"], iw, ih, alpha_meaningful=case["input"]["alpha"]))
        c.add_input_buffer(2, pack_raw_bgra(mark_src, mw, mh, alpha_meaningful=True))
        outs = ([9] if case["parent_output"] else []) + [10 + k for k in range(len(case["branches"]))] if case.get("tree") else [9]
        for o in outs:
            c.add_output_buffer(o)
        status, r = c.send_json("v1/execute", job_of(case))
        COUNTS["coalesced_decodes"] += int(c.L.ifhip_shim_coalesced_decodes(c.p))          # (+= of an int under the GIL)
        COUNTS["fused_decode_resamples"] += int(c.L.ifhip_shim_fused_decode_resamples(c.p))
        COUNTS["device_coded_files"] += int(c.L.ifhip_shim_device_coded_files(c.p))
        if status != 200:
            return f"{status}: {c.error_message()[:160]}", None
        def out(o, is_file):
            if is_file:
                return bytes(c.get_output_buffer(o))
            rows, gw, gh, galpha = unpack_raw_bgra(c.get_output_buffer(o))
            return (rows[:, :4 * gw].copy(), gw, gh, galpha)
        if case.get("tree"):
            return None, [out(o, o >= 10 and "encode" in case["branches"][o - 10]) for o in outs]
        return None, out(9, "encode" in case)


def jpeg_file_of(b, enc, E):
    """MozjpegEncoder::write_frame (mozjpeg.rs:88-94): the frame flattened onto the matte (default white), then the file
    libjpeg-turbo (Pillow) writes from those pixels at 4:2:0"""
    import io

    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 24
    matte = enc.get("matte")
    E[8].apply_matte(b, 0xFFFFFFFF if matte is None else color32_of(matte, E[6][5]))
    E[0].cuda.synchronize()
    out = b.to_numpy()[0]
    rgb = np.ascontiguousarray(out[:, :4 * b.w].reshape(b.h, b.w, 4)[:, :, 2::-1])
    f = io.BytesIO()
    # Option<i32> -> `q as u8` (codecs/auto.rs:201: wraps) -> min(100, ..) (mozjpeg.rs:71) -> jpeg_set_quality (<= 0 is 1)
    quality = max(1, min(100, enc.get("quality", 75) & 0xFF))
    Image.fromarray(rgb).save(f, "JPEG", quality=quality, subsampling="4:2:0", optimize=bool(enc.get("optimize_huffman_coding", False)),
                              progressive=bool(enc.get("progressive", False)))
    return f.getvalue()


def run_mirror(case, inp, E):
    """the same nodes on the Python mirrors -> (error text or None, the expected output)"""
    torch, FlowError, M = E[0], E[4], E[6]
    Bm = M[4]
    w, h = case["size"]
    try:
        def frame(s, fw, fh, alpha, file=None, spec=None):
            if file is not None:
                scale, _, _ = hinted_size(fw, fh, spec)
                hi = spec.get("hints", {})
                spatial = scale < 8 and hi.get("scale_luma_spatially", False)
                return E[7].decode_frames([file], "cuda:0", scale_num=scale, luma_spatial=spatial,
                                          luma_srgb=spatial and hi.get("gamma_correct_for_srgb_during_spatial_luma_scaling", False))
            return Bm.Bitmap.from_numpy(s[None].copy(), fw, fh, s.shape[1], "cuda:0", alpha_meaningful=alpha)
        def finish(b, enc):
            if enc is None:
                torch.cuda.synchronize()
                out = b.to_numpy()[0]
                return (out[:, :4 * b.w].copy(), b.w, b.h, bool(b.alpha_meaningful))
            return jpeg_file_of(b, enc, E)
        if case.get("tree"):
            parent = frame(inp["src"], w, h, case["alpha"], inp["file0"], case.get("jpeg"))
            for node in case["nodes"]:
                parent = mirror_apply(parent, node, M, inp["mark"])
            results = [finish(parent, None)] if case["parent_output"] else []
            for br in case["branches"]:
                b = parent
                for node in br["chain"]:
                    name = node if isinstance(node, str) else next(iter(node))
                    # the node reads the shared parent (it is the first, or the ones before it disappeared and were snapped
                    # together): MutProtect puts a Clone before a mutating node whose parent has other children
                    # (definitions.rs:334-337); fill_rect and watermark are not MutProtect nodes -- the interpreter copies for
                    # them too, the reference would change the siblings' input
                    if b is parent and name in ("flip_h", "flip_v", "rotate_90", "rotate_180", "apply_orientation", "color_filter_srgb", "color_matrix_srgb",
                                                "crop", "region", "region_percent", "fill_rect", "watermark"):
                        b = M[0].clone(parent)
                    if b is parent and name == "constrain":              # its Crop, when the layout has one, is a MutProtect node too
                        g = node["constrain"]["gravity"]["percentage"] if isinstance(node["constrain"].get("gravity"), dict) else None
                        try:
                            if process_constraint(node["constrain"]["mode"], b.w, b.h, node["constrain"].get("w"), node["constrain"].get("h"),
                                                  (g["x"], g["y"]) if g else None)[0]:
                                b = M[0].clone(parent)
                        except LayoutError:
                            pass
                    b = mirror_apply(b, node, M, inp["mark"])
                # (a JPEG encoder flattens the frame it is given in place: a shared one is copied first, as the interpreter does)
                results.append(finish(M[0].clone(b) if "encode" in br and b is parent else b, br.get("encode")))
            return None, results
        if "join" not in case:
            b = frame(inp["src"], w, h, case["alpha"], inp["file0"], case.get("jpeg"))
            for node in case["nodes"]:
                b = mirror_apply(b, node, M, inp["mark"])
        else:
            if "canvas_node" in case:
                p = case["canvas_node"]["create_canvas"]
                cv, _ = canvas_of(M, 1, p["w"], p["h"], "cuda:0", p["color"], p["format"] == "bgra_32")
            else:
                cv = frame(inp["src"], w, h, case["alpha"], inp["file0"], case.get("jpeg"))
            for node in case["canvas_chain"]:
                cv = mirror_apply(cv, node, M)
            iw, ih = case["input"]["size"]
            ib = frame(inp["isrc"], iw, ih, case["input"]["alpha"], inp["file1"], case["input"].get("jpeg"))
            for node in case["input_chain"]:
                ib = mirror_apply(ib, node, M)
            b = mirror_join(cv, ib, case["join"], M)
            for node in case["tail"]:
                b = mirror_apply(b, node, M)
        return None, finish(b, case.get("encode"))
    except (FlowError, ValueError) as e:
        return str(e)[:160], None


def compare(case, shim, mirror):
    """-> the record of a case from the two sides' (error, output)"""
    (shim_err, got), (mir_err, exp) = shim, mirror
    rec = dict(case)
    if shim_err or mir_err:
        rec["shim_error"], rec["mirror_error"] = shim_err, mir_err
        rec["ok"] = bool(shim_err and mir_err)
        rec["refused"] = True
        return rec
    if case.get("tree"):
        rec["ok"] = True
        for k, (g, e) in enumerate(zip(got, exp)):
            same = g == e if isinstance(g, bytes) else (g[1:] == e[1:] and np.array_equal(g[0], e[0]))
            if not same:
                rec["ok"] = False
                rec.setdefault("outputs_differ", []).append([k, "file" if isinstance(g, bytes) else [list(g[1:]), list(e[1:])]])
        return rec
    if "encode" in case:
        rec["ok"] = got == exp
        if not rec["ok"]:
            rec["file_bytes"] = [len(got), len(exp)]
            rec["first_differing_byte"] = next((i for i in range(min(len(got), len(exp))) if got[i] != exp[i]), min(len(got), len(exp)))
        return rec
    same_meta = got[1:] == exp[1:]
    same_px = same_meta and np.array_equal(got[0], exp[0])
    rec["ok"] = bool(same_px)
    if not same_px:
        rec["shim"] = [got[1], got[2], got[3]]
        rec["mirror"] = [exp[1], exp[2], exp[3]]
        if same_meta:
            d = np.argwhere(got[0] != exp[0])
            rec["bytes_differ"] = int(len(d))
            rec["first"] = [int(v) for v in d[0]] + [int(got[0][tuple(d[0])]), int(exp[0][tuple(d[0])])]
    return rec


def run_case(case, E):
    """-> the record of one case: both sides run, compared"""
    inp = case_inputs(case, E)
    return compare(case, run_shim(case, inp, E), run_mirror(case, inp, E))


def environment():
    import torch
    from imageflow_amd.abi import Context, pack_raw_bgra, unpack_raw_bgra
    from imageflow_amd.errors import ErrorKind, FlowError
    from imageflow_amd.flow.nodes import clone_crop_fill_expand as CC, color as CO, rotate_flip_transpose as RF, scale_render as SR, watermark as WM
    from imageflow_amd.graphics import bitmaps as Bm
    from imageflow_amd.graphics.bitmaps import color32
    from imageflow_amd.graphics.color import WorkingFloatspace
    from imageflow_amd.graphics.weights import JSON_FILTER_NAMES
    from tests import util as U
    M = (CC, RF, SR, CO, Bm, color32, JSON_FILTER_NAMES, WorkingFloatspace, WM, FlowError, ErrorKind)
    from imageflow_amd.codecs import mozjpeg_decoder as MD
    from imageflow_amd.graphics import blend as BL
    return (torch, Context, pack_raw_bgra, unpack_raw_bgra, FlowError, U, M, MD, BL)


def sweep(seed, seconds=None, chains=None, out=None, threads=1):
    """run cases until `seconds` have passed or `chains` cases are done -> (summary, the failing records).  threads > 1: the
    mirrors run one case after the other as before, the JOBS of a group of 8 x threads cases run concurrently on `threads`
    host threads (one context per job, as imageflow_abi/src/lib.rs:20-27 prescribes) -- per-thread streams, the block cache's
    stream-ordered release and the decode coalescer (JPEG sources of concurrent jobs share one entropy batch) under load."""
    E = environment()
    rng = np.random.default_rng(seed)
    t_end = time.time() + (seconds if seconds is not None else 1e9)
    done = bad = both_refuse = one_refuses = graphs = jpegs = files = trees = progressive = damaged = 0
    failing = []
    f = open(out, "w") if out else None
    pool = None
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(threads)
    while time.time() < t_end and (chains is None or done < chains):
        group = 1 if pool is None else 8 * threads
        if chains is not None:
            group = min(group, chains - done)
        cases = [draw_case(rng, pool_jpeg=pool is not None) for _ in range(group)]
        if pool is None:
            recs = [run_case(cases[0], E)]
        else:
            inputs = [case_inputs(c, E) for c in cases]
            jobs = [pool.submit(run_shim, c, i, E) for c, i in zip(cases, inputs)]      # the jobs run while the mirrors do
            mirrors = [run_mirror(c, i, E) for c, i in zip(cases, inputs)]
            recs = [compare(c, j.result(), m) for c, j, m in zip(cases, jobs, mirrors)]
        for case, rec in zip(cases, recs):
            rec["case"] = done
            graphs += "join" in case
            jpegs += ("jpeg" in case) + ("jpeg" in case.get("input", {}))
            files += ("encode" in case) + sum("encode" in b for b in case.get("branches", []))
            trees += bool(case.get("tree"))
            damaged += ("damage" in case.get("jpeg", {})) + ("damage" in case.get("input", {}).get("jpeg", {}))
            progressive += bool(case.get("jpeg", {}).get("progressive")) + bool(case.get("input", {}).get("jpeg", {}).get("progressive"))
            if rec.get("refused"):
                both_refuse += rec["ok"]
                one_refuses += not rec["ok"]
            if not rec["ok"]:
                bad += 1
                if len(failing) < 3000:
                    failing.append(rec)
            if f and ((not rec["ok"] and bad <= 3000) or done % 500 == 0):       # every disagreement (the first 3 000), every 500th case
                f.write(json.dumps(rec) + "\n")
                f.flush()
            done += 1
    if pool is not None:
        pool.shutdown()
    summary = {"summary": True, "seed": seed, "threads": threads, "chains": done, "graphs": graphs, "trees": trees, "jpeg_sources": jpegs, "progressive_sources": progressive, "damaged_sources": damaged, "jpeg_outputs": files,
               "disagreements": bad, "both_refuse": both_refuse, "only_one_side_refuses": one_refuses, **COUNTS}
    if f:
        f.write(json.dumps(summary) + "\n")
        f.close()
    return summary, failing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--chains", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", type=int, default=1, help="jobs of a group of cases run concurrently on this many host threads")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_shim.jsonl"))
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    summary, _ = sweep(args.seed, seconds=args.seconds, chains=args.chains, out=args.out, threads=args.threads)
    print(json.dumps(summary))
    sys.exit(1 if summary["disagreements"] else 0)


if __name__ == "__main__":
    main()
