#!/usr/bin/env python3
"""Research tool (GPU): the JPEG pixel stage over samplings x decode scales x luma selector (not only BASELINE config 4's
4:2:0 at 4/8): time of read_frames (coefficient planes -> BGRA at the scaled size) and of read_frames_into (-> a 400-wide
thumbnail in one call), as source megapixels per second -- finds combinations that fall onto a slow path."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from imageflow_amd.codecs.mozjpeg_decoder import JpegPixelStage  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams  # noqa: E402

SAMPLINGS = {"420": (3, (2, 1, 1), (2, 1, 1)), "422": (3, (2, 1, 1), (1, 1, 1)), "440": (3, (1, 1, 1), (2, 1, 1)),
             "444": (3, (1, 1, 1), (1, 1, 1)), "gray": (1, (1,), (1,))}


def main():
    dev = "cuda:0"
    n = 32
    for (w, h) in ((1920, 1080), (1001, 667)):
        for name, (nc, hs, vs) in SAMPLINGS.items():
            for scale_num in (8, 6, 5, 4, 3, 2, 1):
                for spatial in ((False, True) if scale_num < 8 else (False,)):
                    try:
                        st = JpegPixelStage(w, h, nc, hs, vs, n, dev, scale_num=scale_num, luma_spatial=spatial, luma_srgb=spatial)
                        g = torch.Generator(device=dev)
                        g.manual_seed(1)
                        coef = []
                        for c in range(nc):
                            shape = (n, st.blocks_h[c], st.blocks_w[c], 64)
                            t = torch.randint(-30, 31, shape, dtype=torch.int16, device=dev, generator=g)
                            mask = torch.rand(shape, device=dev, generator=g) < 0.15
                            mask[..., 0] = True
                            coef.append(t * mask)
                        qt = torch.randint(1, 40, (n, nc, 64), dtype=torch.int16, device=dev, generator=g)
                        out = Bitmap.create_u8(n, st.out_w, st.out_h, dev)
                        tw = min(400, st.out_w)
                        th = max(1, round(st.out_h * tw / st.out_w))
                        small = Bitmap.create_u8(n, tw, th, dev)
                        info = ScaleAndRenderParams(0, 0, tw, th)

                        def timed(f, reps=5):
                            f()
                            torch.cuda.synchronize()
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            for _ in range(reps):
                                f()
                            e1.record()
                            torch.cuda.synchronize()
                            return e0.elapsed_time(e1) / reps
                        t_dec = timed(lambda: st.read_frames(coef, qt, out))
                        fused = st.read_frames_into(coef, qt, small, info)
                        t_one = timed(lambda: st.read_frames_into(coef, qt, small, info))
                        rec = {"size": [w, h], "sampling": name, "scale_num": scale_num, "luma_spatial_srgb": spatial, "decoded": [st.out_w, st.out_h],
                               "decode_ms": round(t_dec, 4), "decode_source_GPps": round(n * w * h / t_dec / 1e6, 1),
                               "decode_resample_ms": round(t_one, 4), "one_call_fused": bool(fused), "chain_source_GPps": round(n * w * h / t_one / 1e6, 1)}
                    except Exception as e:  # noqa: BLE001
                        rec = {"size": [w, h], "sampling": name, "scale_num": scale_num, "luma_spatial_srgb": spatial, "error": str(e)[:160]}
                    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
