#!/usr/bin/env python3
"""Forward (encode-side) pixel stage only: ms per 32 frames 3840x2160, 4:2:0 and 4:4:4 (A/B helper)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageflow_amd.codecs.mozjpeg import JpegForwardStage, quant_tables_for_quality
from imageflow_amd.graphics.bitmaps import Bitmap
n, w, h, dev = 32, 3840, 2160, "cuda:0"
frames = Bitmap.create_u8(n, w, h, dev)
frames.data.copy_(torch.randint(0, 256, frames.data.shape, dtype=torch.uint8, device=dev))
qt = torch.from_numpy(np.stack([quant_tables_for_quality(90)] * n).view(np.int16)).to(dev)
out = {}
for name, hs, vs in (("420", (2, 1, 1), (2, 1, 1)), ("444", (1, 1, 1), (1, 1, 1))):
    st = JpegForwardStage(w, h, hs, vs, n, dev)
    coef = st.write_frames(frames, qt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        st.write_frames(frames, qt, coef)
    torch.cuda.synchronize()
    out[name] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
print(json.dumps(out))
