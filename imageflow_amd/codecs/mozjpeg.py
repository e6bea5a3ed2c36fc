"""Host mirror of the pixel half of MozjpegEncoder::write_frame (imageflow_core/src/codecs/mozjpeg.rs:78-160, the
classic preset: Defaults::LibJPEGv6 -> set_fastest_defaults, so no trellis quantisation and no overshoot deringing):
what libjpeg runs between write_scanlines and the entropy coder -- BGRA -> YCbCr, chroma down-sampling with edge
expansion, islow forward DCT, quantisation, dummy blocks -- runs in libimageflow_hip.so on frames that stay in HBM.
The entropy coder runs on the device too for the preset's default (baseline, Annex K tables: JpegEntropyStage), on the
host for its progressive / optimize_coding options (write_jpeg*); evalchroma's sampling decision stays on the host."""
import ctypes as C

import numpy as np
import torch

from .. import _native
from ..graphics.bitmaps import Bitmap

DEFAULT_QUALITY = 90          # mozjpeg.rs:21

# ITU T.81 Annex K.1 / K.2 in natural order: the base tables jpeg_set_quality scales (jcparam.c std_*_quant_tbl)
STD_LUMINANCE_QUANT_TBL = (
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101,
    72, 92, 95, 98, 112, 100, 103, 99)
STD_CHROMINANCE_QUANT_TBL = (
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99)


def quant_tables_for_quality(quality):
    """jpeg_set_quality(cinfo, q, force_baseline = TRUE) (mozjpeg.rs:113-115 -> jcparam.c jpeg_quality_scaling +
    jpeg_add_quant_table): uint16 [3][64] natural order, rows = the table each of Y, Cb, Cr uses."""
    q = min(100, max(1, int(quality)))
    scale = 5000 // q if q < 50 else 200 - 2 * q
    out = np.zeros((3, 64), np.uint16)
    for row, base in ((0, STD_LUMINANCE_QUANT_TBL), (1, STD_CHROMINANCE_QUANT_TBL)):
        t = (np.array(base, np.int64) * scale + 50) // 100
        out[row] = np.clip(t, 1, 255)
    out[2] = out[1]
    return out


def sampling_factors(cb, cr):
    """mozjpeg.rs:141-149: evalchroma's chroma pixel sizes -> per-component (h_samp, v_samp)."""
    mh, mv = max(cb[0], cr[0]), max(cb[1], cr[1])
    sizes = ((1, 1), cb, cr)
    return [mh // s[0] for s in sizes], [mv // s[1] for s in sizes]


def _bind():
    L = _native.lib()
    if getattr(L, "_jpeg_fwd_bound", False):
        return L
    L.ifhip_jpeg_fwd_stage_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
    L.ifhip_jpeg_fwd_stage_destroy.argtypes = [C.c_void_p]
    L.ifhip_jpeg_fwd_stage_destroy.restype = None
    L.ifhip_jpeg_fwd_stage_block_dims.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifhip_jpeg_forward_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_uint32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifhip_jpeg_forward.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
    L._jpeg_fwd_bound = True
    return L


class JpegForwardStage:
    """ifhip_jpeg_fwd_stage: geometry + the down-sampled component planes between the colour and the DCT kernels."""

    def __init__(self, width, height, h_samp, v_samp, max_images, device="cuda:0"):
        L = _bind()
        self.width, self.height = width, height
        self.device = torch.device(device)
        self._h = C.c_void_p()
        hs, vs = np.array(list(h_samp), np.uint8), np.array(list(v_samp), np.uint8)
        with torch.cuda.device(self.device):
            _native.check(L.ifhip_jpeg_fwd_stage_create(C.byref(self._h), width, height, hs.ctypes.data, vs.ctypes.data, max_images))
        bw, bh = np.zeros(3, np.uint32), np.zeros(3, np.uint32)
        _native.check(L.ifhip_jpeg_fwd_stage_block_dims(self._h, bw.ctypes.data, bh.ctypes.data))
        self.blocks_w, self.blocks_h = [int(v) for v in bw], [int(v) for v in bh]

    def write_frames(self, frames: Bitmap, qt, coef=None):
        """frames: n BGRA frames (already matted, mozjpeg.rs:88-94); qt: cuda tensor [n, 3, 64] of uint16 bit patterns.
        Returns int16 cuda tensors [n, bh_c, bw_c, 64]: what the entropy coder consumes."""
        L = _bind()
        n = frames.n
        if coef is None:
            coef = [torch.empty((n, self.blocks_h[c], self.blocks_w[c], 64), dtype=torch.int16, device=self.device) for c in range(3)]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _native.check(L.ifhip_jpeg_forward_batch_device(self._h, frames.data.data_ptr(), frames.image_bytes, frames.stride,
                                                            qt.data_ptr(), n, coef[0].data_ptr(), coef[1].data_ptr(),
                                                            coef[2].data_ptr(), C.c_void_p(stream)))
        return coef

    def __del__(self):
        try:
            if self._h:
                _bind().ifhip_jpeg_fwd_stage_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


ENC_BAD_COEFFICIENT, ENC_SCAN_OVERFLOW, ENC_FILE_OVERFLOW = 1, 2, 4     # include/imageflow_hip.h IFHIP_ENC_*


class JpegEntropyStage:
    """ifhip_jpeg_enc_stage: the device entropy coder (csrc/jpeg_encode.hip) -- what compressor.write_scanlines / finish run
    behind the pixel stage (mozjpeg.rs:155-175) for a baseline file with the Annex K tables: coefficient planes in HBM in,
    complete files in HBM out."""

    def __init__(self, width, height, h_samp, v_samp, blocks_w, blocks_h, max_images, device="cuda:0", scan_capacity=0):
        L = _bind()
        L.ifhip_jpeg_enc_stage_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_uint32, C.c_size_t]
        L.ifhip_jpeg_enc_stage_destroy.argtypes = [C.c_void_p]
        L.ifhip_jpeg_enc_stage_destroy.restype = None
        L.ifhip_jpeg_enc_stage_max_file_bytes.argtypes = [C.c_void_p]
        L.ifhip_jpeg_enc_stage_max_file_bytes.restype = C.c_size_t
        L.ifhip_jpeg_encode_batch_device.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                                     C.c_void_p]
        self.device = torch.device(device)
        self.ncomp = len(list(blocks_w))
        self.max_images = max_images
        self._h = C.c_void_p()
        pad = [1] * (3 - self.ncomp)
        hs, vs = np.array(list(h_samp)[:self.ncomp] + pad, np.uint8), np.array(list(v_samp)[:self.ncomp] + pad, np.uint8)
        bw, bh = np.array(list(blocks_w) + [0] * (3 - self.ncomp), np.uint32), np.array(list(blocks_h) + [0] * (3 - self.ncomp), np.uint32)
        with torch.cuda.device(self.device):
            _native.check(L.ifhip_jpeg_enc_stage_create(C.byref(self._h), width, height, self.ncomp, hs.ctypes.data, vs.ctypes.data,
                                                        bw.ctypes.data, bh.ctypes.data, max_images, scan_capacity))
        self.max_file_bytes = int(L.ifhip_jpeg_enc_stage_max_file_bytes(self._h))

    def encode_device(self, coef, quality, file_pitch=None, files=None):
        """coef: 1 or 3 int16 cuda tensors [n, bh_c, bw_c, 64].  Returns (files [n, file_pitch] uint8, lengths [n] int32,
        status [n] int32), all cuda tensors; nothing is synchronised."""
        L = _bind()
        n = coef[0].shape[0]
        if file_pitch is None:
            file_pitch = files.shape[1] if files is not None else (self.max_file_bytes + 15) // 16 * 16
        if files is None:
            files = torch.empty((n, file_pitch), dtype=torch.uint8, device=self.device)
        lengths = torch.zeros(n, dtype=torch.int32, device=self.device)
        status = torch.zeros(n, dtype=torch.int32, device=self.device)
        ptrs = [c.data_ptr() for c in coef] + [None] * (3 - len(coef))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _native.check(L.ifhip_jpeg_encode_batch_device(self._h, *ptrs, int(quality), n, files.data_ptr(), file_pitch,
                                                           lengths.data_ptr(), status.data_ptr(), C.c_void_p(stream)))
        return files, lengths, status

    def encode(self, coef, quality, file_pitch=None):
        """The n files as bytes (None for a dropped image) and the status words."""
        files, lengths, status = self.encode_device(coef, quality, file_pitch)
        lengths, status = lengths.cpu().numpy(), status.cpu().numpy()
        host = files[:, :max(int(lengths.max()), 1)].cpu().numpy()
        return [host[i, :int(k)].tobytes() if k else None for i, k in enumerate(lengths)], [int(s) for s in status]

    def __del__(self):
        try:
            if self._h:
                _bind().ifhip_jpeg_enc_stage_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


def pack_files_device(files, lengths, out=None):
    """ifhip_pack_files_device: files [n, pitch] uint8 + lengths [n] int32 (as JpegEntropyStage.encode_device leaves them) ->
    (packed uint8 [capacity], offsets int64 [n + 1]): the files back to back, 16-byte aligned starts -- ONE message for the
    job's final gather.  `out`: a buffer to reuse (its size is the capacity; default n * pitch, which always fits).
    Asynchronous on the current stream; offsets[n] is the number of bytes used."""
    L = _bind()
    L.ifhip_pack_files_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    n, pitch = files.shape
    if out is None:
        out = torch.empty(n * pitch, dtype=torch.uint8, device=files.device)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=files.device)
    stream = torch.cuda.current_stream(files.device).cuda_stream
    with torch.cuda.device(files.device):
        _native.check(L.ifhip_pack_files_device(files.data_ptr(), pitch, lengths.data_ptr(), n, out.data_ptr(), out.numel(),
                                                offsets.data_ptr(), C.c_void_p(stream)))
    return out, offsets


def jpeg_forward_host(bgra, width, height, stride, h_samp, v_samp, qt):
    """Host-buffer drop-in (numpy): BGRA rows -> list of int16 arrays [bh_c][bw_c][64]."""
    L = _bind()
    hs, vs = np.array(list(h_samp), np.uint8), np.array(list(v_samp), np.uint8)
    hmax, vmax = int(hs.max()), int(vs.max())
    mw, mh = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    coef = [np.zeros((mh * int(vs[c]), mw * int(hs[c]), 64), np.int16) for c in range(3)]
    src = np.ascontiguousarray(bgra, np.uint8)
    q = np.ascontiguousarray(qt, np.uint16)
    _native.check(L.ifhip_jpeg_forward(src.ctypes.data, width, height, stride, hs.ctypes.data, vs.ctypes.data, q.ctypes.data,
                                       coef[0].ctypes.data, coef[1].ctypes.data, coef[2].ctypes.data))
    return coef


JPEG_OPTIMIZE_HUFFMAN, JPEG_PROGRESSIVE = 1, 2          # include/imageflow_hip.h IFHIP_JPEG_*


def write_jpeg(coef, width, height, h_samp, v_samp, quality, progressive=False, optimize_coding=False):
    """The entropy-coding half (host code of the library, csrc/jpeg_write.cpp): quantised coefficient planes
    [bh_c][bw_c][64] (numpy int16, 1 or 3 of them) -> the bytes libjpeg-turbo would write for them."""
    L = _native.lib()
    L.ifhip_jpeg_write.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                   C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    planes = [np.ascontiguousarray(c, np.int16) for c in coef]
    n = len(planes)
    bw, bh = np.array([p.shape[1] for p in planes] + [0] * (3 - n), np.uint32), np.array([p.shape[0] for p in planes] + [0] * (3 - n), np.uint32)
    hs, vs = np.array(list(h_samp) + [1] * (3 - n), np.uint8), np.array(list(v_samp) + [1] * (3 - n), np.uint8)
    flags = (JPEG_PROGRESSIVE if progressive else 0) | (JPEG_OPTIMIZE_HUFFMAN if optimize_coding else 0)
    ptrs = [p.ctypes.data for p in planes] + [None] * (3 - n)
    size = C.c_size_t(0)
    args = ptrs + [bw.ctypes.data, bh.ctypes.data, n, hs.ctypes.data, vs.ctypes.data, width, height, int(quality), flags]
    # one pass when the guess is large enough (a file is far smaller than its coefficients); the entry point reports the
    # size it needs otherwise
    out = np.empty(max(4096, sum(p.size for p in planes) // 2), np.uint8)
    rc = L.ifhip_jpeg_write(*args, out.ctypes.data, out.size, C.byref(size))
    if rc != 0 and size.value > out.size:
        out = np.empty(size.value, np.uint8)
        rc = L.ifhip_jpeg_write(*args, out.ctypes.data, out.size, C.byref(size))
    _native.check(rc)
    return out[:size.value].tobytes()


def write_jpeg_batch(coef, width, height, h_samp, v_samp, quality, progressive=False, optimize_coding=False, threads=0):
    """n images of one geometry: coef = planes [n][bh_c][bw_c][64] (numpy int16, 1 or 3 of them), coded on `threads` host
    threads (0: one per core).  Returns the n files."""
    L = _native.lib()
    L.ifhip_jpeg_write_batch.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                         C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    planes = [np.ascontiguousarray(c, np.int16) for c in coef]
    ncomp, n = len(planes), planes[0].shape[0]
    bw = np.array([p.shape[2] for p in planes] + [0] * (3 - ncomp), np.uint32)
    bh = np.array([p.shape[1] for p in planes] + [0] * (3 - ncomp), np.uint32)
    hs, vs = np.array(list(h_samp) + [1] * (3 - ncomp), np.uint8), np.array(list(v_samp) + [1] * (3 - ncomp), np.uint8)
    flags = (JPEG_PROGRESSIVE if progressive else 0) | (JPEG_OPTIMIZE_HUFFMAN if optimize_coding else 0)
    ptrs = [p.ctypes.data for p in planes] + [None] * (3 - ncomp)
    total = C.c_size_t(0)
    offsets, lengths = np.zeros(n, np.uintp), np.zeros(n, np.uintp)
    args = ptrs + [bw.ctypes.data, bh.ctypes.data, ncomp, hs.ctypes.data, vs.ctypes.data, width, height, int(quality), flags, n, int(threads)]
    out = np.empty(max(4096, sum(p.size for p in planes) // 2), np.uint8)
    rc = L.ifhip_jpeg_write_batch(*args, out.ctypes.data, out.size, offsets.ctypes.data, lengths.ctypes.data, C.byref(total))
    if rc != 0 and total.value > out.size:
        out = np.empty(total.value, np.uint8)
        rc = L.ifhip_jpeg_write_batch(*args, out.ctypes.data, out.size, offsets.ctypes.data, lengths.ctypes.data, C.byref(total))
    _native.check(rc)
    return [out[int(o):int(o) + int(k)].tobytes() for o, k in zip(offsets, lengths)]


class MozjpegEncoder:
    """MozjpegEncoder::create_classic + write_frame (mozjpeg.rs:60-77, :78-160) over the device stages: apply_matte
    (default white, :88-92), the forward pixel stage at 4:2:0 (the maximum evalchroma may choose, :133) and the file
    writer with the preset's progressive / optimize_coding options.  Returns the file's bytes."""

    def __init__(self, quality=None, progressive=None, optimize_coding=None, matte=0xFFFFFFFF):
        self.quality = min(100, DEFAULT_QUALITY if quality is None else int(quality))
        self.progressive, self.optimize_coding, self.matte = bool(progressive), bool(optimize_coding), matte

    @classmethod
    def create_classic(cls, quality=None, progressive=None, optimize_coding=None, matte=None):
        return cls(quality, progressive, optimize_coding, 0xFFFFFFFF if matte is None else matte)

    def write_frame(self, bitmap: Bitmap, frame=0):
        from ..graphics.blend import apply_matte
        apply_matte(bitmap, self.matte)
        bitmap.alpha_meaningful = False                                       # :94
        hs, vs = sampling_factors((2, 2), (2, 2))
        stage = JpegForwardStage(bitmap.w, bitmap.h, hs, vs, bitmap.n, bitmap.data.device)
        qt = torch.from_numpy(np.stack([quant_tables_for_quality(self.quality)] * bitmap.n).view(np.int16)).to(bitmap.data.device)
        coef = [c[frame].cpu().numpy() for c in stage.write_frames(bitmap, qt)]
        return write_jpeg(coef, bitmap.w, bitmap.h, hs, vs, self.quality, self.progressive, self.optimize_coding)

    def write_frames(self, bitmap: Bitmap, threads=0, device_entropy=True):
        """Every frame of the bitmap: one device launch for the pixel stage; the files coded on the device (the preset's
        default: baseline, Annex K tables) or -- progressive / optimize_coding, or device_entropy=False -- in parallel on
        the host.  The two coders write the same bytes."""
        from ..graphics.blend import apply_matte
        apply_matte(bitmap, self.matte)
        bitmap.alpha_meaningful = False
        hs, vs = sampling_factors((2, 2), (2, 2))
        stage = JpegForwardStage(bitmap.w, bitmap.h, hs, vs, bitmap.n, bitmap.data.device)
        qt = torch.from_numpy(np.stack([quant_tables_for_quality(self.quality)] * bitmap.n).view(np.int16)).to(bitmap.data.device)
        if device_entropy and not self.progressive and not self.optimize_coding:
            coder = JpegEntropyStage(bitmap.w, bitmap.h, hs, vs, stage.blocks_w, stage.blocks_h, bitmap.n, bitmap.data.device)
            files, status = coder.encode(stage.write_frames(bitmap, qt), self.quality)
            if any(status):                                                   # (cannot happen with coefficients of the forward stage)
                raise RuntimeError(f"device entropy coder dropped images: status {status}")
            return files
        coef = [c.cpu().numpy() for c in stage.write_frames(bitmap, qt)]
        return write_jpeg_batch(coef, bitmap.w, bitmap.h, hs, vs, self.quality, self.progressive, self.optimize_coding, threads)
