"""Host mirror of the pixel half of MozJpegDecoder::read_frame (imageflow_core/src/codecs/mozjpeg_decoder.rs:295-420):
what libjpeg does after entropy decoding -- de-quantise, islow IDCT, fancy chroma up-sampling, YCbCr -> BGRA
(out_color_space = JCS_EXT_BGRA) -- runs in libimageflow_hip.so on frames that stay in HBM.
The Huffman pass stays on the host (mozjpeg's jpeg_read_coefficients in the reference integration)."""
import ctypes as C

import numpy as np
import torch

from .. import _native
from ..graphics.bitmaps import Bitmap, get_stride


def _bind():
    L = _native.lib()
    if getattr(L, "_jpeg_bound", False):
        return L
    L.ifhip_jpeg_idct_color.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32]
    L.ifhip_jpeg_stage_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_uint32]
    L.ifhip_jpeg_stage_output_size.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifhip_jpeg_stage_destroy.argtypes = [C.c_void_p]
    L.ifhip_jpeg_stage_destroy.restype = None
    L.ifhip_jpeg_stage_block_dims.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifhip_jpeg_idct_color_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                     C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
    L.ifhip_jpeg_decode_resample_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                          C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    L._jpeg_bound = True
    return L


def apply_downscaling(original_width, original_height, downscale_if_wider_than, or_if_taller_than,
                      downscaled_min_width, downscaled_min_height):
    """MzDec::apply_downscaling (mozjpeg_decoder.rs:588-618): the scale_num/8 libjpeg is asked for, given the decoder
    hints (ffi/c_interop.rs:6-15).  Returns (scale_num, w, h); scale_num == 8 means full-size decode.  Every value this
    can return (1..6, 8) is implemented by the GPU pixel stage."""
    if (downscaled_min_width > 0 and downscaled_min_height > 0
            and (original_width > downscale_if_wider_than or original_height > or_if_taller_than)):
        for i in range(1, 8):
            if i == 7:
                continue                      # "Because 7/8ths is slower than 8/8"
            new_w = -(-original_width * i // 8)
            new_h = -(-original_height * i // 8)
            if new_w >= downscaled_min_width and new_h >= downscaled_min_height:
                return i, new_w, new_h
    return 8, original_width, original_height


def idct_method_for_luma(scaled_size, scale_luma_spatially, gamma_correct_for_srgb):
    """wrap_jpeg_idct_method_selector (codec_jpeg_wrapper.c:274-343): which block routine the luma component gets.
    Returns ("islow", 8) or ("spatial" | "spatial_srgb", n)."""
    if 0 < scaled_size < 8 and scale_luma_spatially:
        return ("spatial_srgb" if gamma_correct_for_srgb else "spatial", scaled_size)
    return ("islow", 8)


class JpegPixelStage:
    """ifhip_jpeg_stage: geometry + the component planes between the IDCT and the colour kernel."""

    def __init__(self, width, height, n_components, h_samp, v_samp, max_images, device="cuda:0", scale_num=8,
                 luma_spatial=False, luma_srgb=False):
        L = _bind()
        self.width, self.height, self.n = width, height, n_components
        self.device = torch.device(device)
        self._h = C.c_void_p()
        hs = np.array(list(h_samp)[:3] + [0] * (3 - len(h_samp)), np.uint8)
        vs = np.array(list(v_samp)[:3] + [0] * (3 - len(v_samp)), np.uint8)
        with torch.cuda.device(self.device):
            _native.check(L.ifhip_jpeg_stage_create(C.byref(self._h), width, height, n_components, hs.ctypes.data,
                                                    vs.ctypes.data, scale_num, int(luma_spatial), int(luma_srgb), max_images))
        ow, oh = C.c_uint32(), C.c_uint32()
        _native.check(L.ifhip_jpeg_stage_output_size(self._h, C.byref(ow), C.byref(oh)))
        self.out_w, self.out_h = ow.value, oh.value
        bw, bh = np.zeros(3, np.uint32), np.zeros(3, np.uint32)
        _native.check(L.ifhip_jpeg_stage_block_dims(self._h, bw.ctypes.data, bh.ctypes.data))
        self.blocks_w, self.blocks_h = [int(v) for v in bw], [int(v) for v in bh]

    def read_frames(self, coef, qt, out: Bitmap = None):
        """coef: list of int16 cuda tensors [n, bh_c, bw_c, 64]; qt: uint16 cuda tensor [n, ncomp, 64] (as int16 storage).
        Returns a Bitmap of n BGRA frames (alpha not meaningful, as MozJpegDecoder::read_frame creates it, :101-123)."""
        L = _bind()
        n = coef[0].shape[0]
        if out is None:
            out = Bitmap.create_u8(n, self.out_w, self.out_h, self.device, alpha_meaningful=False)
        ptr = [coef[c].data_ptr() if c < self.n else None for c in range(3)]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _native.check(L.ifhip_jpeg_idct_color_batch_device(self._h, ptr[0], ptr[1], ptr[2], qt.data_ptr(), n,
                                                               out.data.data_ptr(), out.image_bytes, out.stride,
                                                               C.c_void_p(stream)))
        return out

    def read_frames_into(self, coef, qt, canvas: Bitmap, info, plan=None):
        """read_frame (mozjpeg_decoder.rs:346-362) + DrawImageDef::render's scale_and_render (scale_render.rs:304-313) as one
        device call: the decoded frames are rendered into the (info.x, info.y, info.w, info.h) rect of `canvas`.  Returns
        True when no decoded BGRA frame went through HBM (component planes at output resolution: the resampler reads them and
        converts the colours in its row fetch), False when the call ran the two-step chain through a scratch bitmap."""
        from ..graphics.scaling import plan_for
        L = _bind()
        n = coef[0].shape[0]
        assert canvas.n == n
        plan = plan or plan_for(self.out_w, self.out_h, info.w, info.h, info.interpolation_filter, info.sharpen_percent_goal, self.device)
        ptr = [coef[c].data_ptr() if c < self.n else None for c in range(3)]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        fused = C.c_int(0)
        with torch.cuda.device(self.device):
            _native.check(L.ifhip_jpeg_decode_resample_batch_device(
                self._h, ptr[0], ptr[1], ptr[2], qt.data_ptr(), n, plan.handle, canvas.data.data_ptr(), canvas.image_bytes,
                canvas.w, canvas.h, canvas.stride, info.x, info.y, int(info.scale_in_colorspace), int(canvas.compose),
                int(canvas.matte), C.addressof(fused), C.c_void_p(stream)))
        return bool(fused.value)

    def __del__(self):
        try:
            if self._h:
                _bind().ifhip_jpeg_stage_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


def jpeg_idct_color_host(coef, qt, n_components, h_samp, v_samp, width, height, stride=None, scale_num=8,
                         luma_spatial=False, luma_srgb=False):
    """Host-buffer drop-in (numpy): coef = list of int16 arrays [bh][bw][64], qt = uint16 [ncomp][64] -> BGRA rows."""
    L = _bind()
    ow, oh = (width * scale_num + 7) // 8, (height * scale_num + 7) // 8
    stride = stride or get_stride(ow)
    out = np.zeros((oh, stride), np.uint8)
    hs = np.array(list(h_samp)[:3], np.uint8)
    vs = np.array(list(v_samp)[:3], np.uint8)
    keep = [np.ascontiguousarray(coef[c]) for c in range(n_components)]
    p = [keep[c].ctypes.data if c < n_components else None for c in range(3)]
    q = np.ascontiguousarray(qt[:n_components], np.uint16)
    _native.check(L.ifhip_jpeg_idct_color(p[0], p[1], p[2], q.ctypes.data, n_components, hs.ctypes.data, vs.ctypes.data,
                                          width, height, scale_num, int(luma_spatial), int(luma_srgb), out.ctypes.data, stride))
    return out


# ---------------------------------------------------------------------------------------------------------
# entropy stage on the GPU (csrc/jpeg_entropy.hip): whole baseline files in, coefficient planes out
# ---------------------------------------------------------------------------------------------------------
def _bind_entropy():
    L = _native.lib()
    if getattr(L, "_jpeg_entropy_bound", False):
        return L
    L.ifhip_jpeg_parse_headers.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 9
    L.ifhip_jpeg_entropy_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32]
    L.ifhip_jpeg_entropy_destroy.argtypes = [C.c_void_p]
    L.ifhip_jpeg_entropy_destroy.restype = None
    L.ifhip_jpeg_entropy_info.argtypes = [C.c_void_p] + [C.c_void_p] * 9
    L.ifhip_jpeg_entropy_quant_tables.argtypes = [C.c_void_p, C.c_void_p]
    L.ifhip_jpeg_entropy_decode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ifhip_jpeg_entropy_prepare.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t]
    L.ifhip_jpeg_prepared_destroy.argtypes = [C.c_void_p]
    L.ifhip_jpeg_prepared_destroy.restype = None
    L.ifhip_jpeg_prepared_upload.argtypes = [C.c_void_p, C.c_void_p]
    L.ifhip_jpeg_entropy_create_prepared.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint32]
    L._jpeg_entropy_bound = True
    return L


def get_image_info(data: bytes):
    """The header facts MozJpegDecoder::get_unscaled_image_info needs (mozjpeg_decoder.rs:245-293), parsed on the host
    (no GPU): dict(width, height, ncomp, hs, vs, bw, bh, qt [3][64] natural order, restart_interval)."""
    L = _bind_entropy()
    buf = np.frombuffer(data, np.uint8)
    w, h, n, ri = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_uint32()
    hs, vs = np.zeros(3, np.uint8), np.zeros(3, np.uint8)
    bw, bh = np.zeros(3, np.uint32), np.zeros(3, np.uint32)
    qt = np.zeros((3, 64), np.uint16)
    _native.check(L.ifhip_jpeg_parse_headers(buf.ctypes.data, len(data), C.addressof(w), C.addressof(h), C.addressof(n),
                                             hs.ctypes.data, vs.ctypes.data, bw.ctypes.data, bh.ctypes.data, qt.ctypes.data,
                                             C.addressof(ri)))
    k = n.value
    return dict(width=w.value, height=h.value, ncomp=k, hs=[int(v) for v in hs[:k]], vs=[int(v) for v in vs[:k]],
                bw=[int(v) for v in bw[:k]], bh=[int(v) for v in bh[:k]], qt=qt, restart_interval=ri.value)


class JpegEntropyBatch:
    """ifhip_jpeg_entropy: n baseline files of one geometry, parsed and un-stuffed on the host, Huffman-decoded on the
    GPU by the self-synchronising parallel decoder."""

    def __init__(self, files, device="cuda:0", prepared=False, upload_stream=None):
        """prepared=True: the per-file form a host with one job per thread uses -- ifhip_jpeg_entropy_prepare for every file,
        then (upload_stream: a torch stream, or None for no early upload) ifhip_jpeg_prepared_upload on that stream, then
        ifhip_jpeg_entropy_create_prepared on the handles, which are destroyed as soon as it returns."""
        L = _bind_entropy()
        self.device = torch.device(device)
        self.n = len(files)
        self._keep = [np.frombuffer(f, np.uint8) for f in files]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            if not prepared:
                ptrs = (C.c_void_p * self.n)(*[k.ctypes.data for k in self._keep])
                lens = (C.c_size_t * self.n)(*[len(f) for f in files])
                _native.check(L.ifhip_jpeg_entropy_create(C.byref(self._h), ptrs, lens, self.n))
            else:
                handles = []
                try:
                    for k in self._keep:
                        h = C.c_void_p()
                        _native.check(L.ifhip_jpeg_entropy_prepare(C.byref(h), k.ctypes.data, len(k)))
                        handles.append(h)
                        if upload_stream is not None:
                            _native.check(L.ifhip_jpeg_prepared_upload(h, C.c_void_p(upload_stream.cuda_stream)))
                    arr = (C.c_void_p * self.n)(*[h.value for h in handles])
                    _native.check(L.ifhip_jpeg_entropy_create_prepared(C.byref(self._h), arr, self.n))
                finally:
                    for h in handles:
                        L.ifhip_jpeg_prepared_destroy(h)
        w, h, n, ns, ng = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_uint32(), C.c_uint32()
        hs, vs = np.zeros(3, np.uint8), np.zeros(3, np.uint8)
        bw, bh = np.zeros(3, np.uint32), np.zeros(3, np.uint32)
        _native.check(L.ifhip_jpeg_entropy_info(self._h, C.addressof(w), C.addressof(h), C.addressof(n), hs.ctypes.data,
                                                vs.ctypes.data, bw.ctypes.data, bh.ctypes.data, C.addressof(ns), C.addressof(ng)))
        self.width, self.height, self.ncomp = w.value, h.value, n.value
        self.h_samp, self.v_samp = [int(v) for v in hs[:self.ncomp]], [int(v) for v in vs[:self.ncomp]]
        self.blocks_w, self.blocks_h = [int(v) for v in bw], [int(v) for v in bh]
        self.n_subsequences, self.n_segments = ns.value, ng.value
        qt = np.zeros((self.n, 3, 64), np.uint16)
        _native.check(L.ifhip_jpeg_entropy_quant_tables(self._h, qt.ctypes.data))
        self.qt = qt
        self.rounds = 0

    def read_coefficients(self, coef=None):
        """-> list of int16 cuda tensors [n, bh_c, bw_c, 64] (jpeg_read_coefficients' virtual block arrays)."""
        L = _bind_entropy()
        if coef is None:
            coef = [torch.empty((self.n, max(self.blocks_h[c], 1), max(self.blocks_w[c], 1), 64), dtype=torch.int16, device=self.device)
                    for c in range(3)]
        r = C.c_uint32()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _native.check(L.ifhip_jpeg_entropy_decode_device(self._h, coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                                                             C.addressof(r), C.c_void_p(stream)))
        self.rounds = r.value
        return coef

    def __del__(self):
        try:
            if self._h:
                _bind_entropy().ifhip_jpeg_entropy_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


def decode_frames(files, device="cuda:0", scale_num=8, luma_spatial=False, luma_srgb=False) -> Bitmap:
    """MozJpegDecoder::read_frame for a batch of equally shaped baseline files, entirely on the device:
    entropy decode -> de-quantise + IDCT -> up-sample + colour -> BGRA frames."""
    ent = JpegEntropyBatch(files, device)
    coef = ent.read_coefficients()
    stage = JpegPixelStage(ent.width, ent.height, ent.ncomp, ent.h_samp, ent.v_samp, ent.n, device, scale_num=scale_num,
                           luma_spatial=luma_spatial, luma_srgb=luma_srgb)
    qt = torch.from_numpy(ent.qt[:, :ent.ncomp].copy().view(np.int16)).to(device)
    return stage.read_frames(coef, qt)
