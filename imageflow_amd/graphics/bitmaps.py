"""imageflow_core/src/graphics/bitmaps.rs mirror, device-resident.

A `Bitmap` here is a *batch* of n equally shaped BGRA8 frames in HBM (one torch uint8 tensor [n, h*stride]),
rows padded to 64 bytes exactly as Bitmap::create_u8 (bitmaps.rs:784-839, get_stride :712-740).
"""
import enum
from dataclasses import dataclass

import torch

from .. import _native
from ..errors import ErrorKind, FlowError


class BitmapCompositing(enum.IntEnum):   # bitmaps.rs:155-160 / ffi/mod.rs:41-47
    ReplaceSelf = 0
    BlendWithSelf = 1
    BlendWithMatte = 2


def get_stride(w):
    return int(_native.lib().ifhip_stride_for_width(w))


def color32(hex_rrggbbaa):
    """imageflow_helpers/src/colors.rs:36-61,77-117: '[#]RGB|RGBA|RRGGBB|RRGGBBAA' -> Color32 0xAARRGGBB."""
    s = hex_rrggbbaa.lstrip("#")
    if len(s) in (3, 4):
        s = "".join(ch * 2 for ch in s)
    if len(s) == 6:
        s += "FF"
    if len(s) != 8:
        raise FlowError(ErrorKind.InvalidArgument, f"bad colour {hex_rrggbbaa!r}")
    r, g, b, a = (int(s[i:i + 2], 16) for i in (0, 2, 4, 6))
    return (a << 24) | (r << 16) | (g << 8) | b


@dataclass
class Bitmap:
    data: torch.Tensor          # uint8 [n, h*stride] on a cuda device
    w: int
    h: int
    stride: int
    alpha_meaningful: bool = False
    compose: BitmapCompositing = BitmapCompositing.ReplaceSelf
    matte: int = 0              # Color32 when compose == BlendWithMatte

    @property
    def n(self):
        return self.data.shape[0]

    @property
    def image_bytes(self):
        return self.data.stride(0)

    @staticmethod
    def create_u8(n, w, h, device, alpha_meaningful=False, compose=BitmapCompositing.ReplaceSelf, matte=0):
        if w == 0 or h == 0:
            raise FlowError(ErrorKind.InvalidArgument, "Bitmap dimensions cannot be zero")
        stride = get_stride(w)
        data = torch.zeros((n, h * stride), dtype=torch.uint8, device=device)
        b = Bitmap(data, w, h, stride, alpha_meaningful, compose, matte)
        if compose == BitmapCompositing.BlendWithMatte and (matte >> 24) != 0:     # bitmaps.rs:829-837
            px = torch.tensor([matte & 255, (matte >> 8) & 255, (matte >> 16) & 255, matte >> 24],
                              dtype=torch.uint8, device=device)
            rows = data.view(n, h, stride)[:, :, : 4 * w].reshape(n, h, w, 4)
            rows[:] = px
        return b

    @staticmethod
    def from_numpy(frames, w, h, stride, device, **kw):
        """frames: uint8 array [n, h*stride] (or [n, h, stride])."""
        t = torch.from_numpy(frames.reshape(frames.shape[0], -1)).to(device)
        return Bitmap(t, w, h, stride, **kw)

    def to_numpy(self):
        """uint8 [n][h][stride].  A cropped window (flow/nodes crop) holds (h-1)*stride + 4*w bytes per frame: its last
        row is returned zero-padded to the stride."""
        flat = self.data.cpu().numpy()
        want = self.h * self.stride
        if flat.shape[1] == want:
            return flat.reshape(self.n, self.h, self.stride)
        import numpy as np
        if flat.shape[1] > want or flat.shape[1] < (self.h - 1) * self.stride + 4 * self.w:
            raise FlowError(ErrorKind.InvalidState, f"bitmap window holds {flat.shape[1]} bytes per frame, {self.w}x{self.h} "
                                                    f"at stride {self.stride} needs {want}")
        out = np.zeros((self.n, want), np.uint8)
        out[:, :flat.shape[1]] = flat
        return out.reshape(self.n, self.h, self.stride)
