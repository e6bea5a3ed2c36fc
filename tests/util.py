"""Shared helpers for parity tests: synthetic frames and oracle drivers."""
import numpy as np

from oracle import oracle as O


def stride_for(w):
    return O.stride_for_width(w)


def gradient_frames(n, w, h, k0=0):
    """bench_graphics.rs:403-414 gradient plus a per-frame offset: B=(x+k)&255, G=(y+k)&255, R=(x+y+k)&255, A=255."""
    st = stride_for(w)
    out = np.zeros((n, h, st), np.uint8)
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    for i in range(n):
        k = k0 + i
        px = out[i, :, : 4 * w].reshape(h, w, 4)
        px[..., 0] = (x + k) & 255
        px[..., 1] = (y + k) & 255
        px[..., 2] = (x + y + k) & 255
        px[..., 3] = 255
    return out


def random_frames(n, w, h, seed0=1000, alpha=True):
    """uniform random bytes, numpy.random.default_rng(seed=1000+k) per frame (worst case for rounding parity)."""
    st = stride_for(w)
    out = np.zeros((n, h, st), np.uint8)
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        out[i] = rng.integers(0, 256, size=(h, st), dtype=np.uint8)
    if not alpha:
        out[:, :, 3:4 * w:4] = 255
    return out


def oracle_render(frames, in_w, in_h, canvas, cw, ch, x, y, w, h, **kw):
    """frames [n, in_h, stride] / canvas [n, ch, cstride] uint8; canvas modified in place. Returns f32 or None."""
    n = frames.shape[0]
    want = kw.pop("want_f32", False)
    f32s = []
    for i in range(n):
        rc, f32 = O.scale_and_render(frames[i], in_w, in_h, canvas[i], cw, ch, x, y, w, h, want_f32=want, **kw)
        assert rc == 0, rc
        f32s.append(f32)
    return np.stack(f32s) if want else None
