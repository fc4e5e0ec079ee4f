"""ctypes binding of the C-ABI in include/cuttlefish_hip.h (libcuttlefish_hip.so).

This is plumbing: the product is the HIP library.  There is no CPU fallback --
if the library or a HIP device is missing every call raises.
"""
from __future__ import annotations

import ctypes
import enum
import os
from typing import Iterable, Optional, Sequence

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libcuttlefish_hip.so")


class Format(enum.IntEnum):
    """cuttlefish::Texture::Format values (lib/include/cuttlefish/Texture.h:59-130)."""
    R4G4 = 1
    R4G4B4A4 = 2
    B4G4R4A4 = 3
    A4R4G4B4 = 4
    R5G6B5 = 5
    B5G6R5 = 6
    R5G5B5A1 = 7
    B5G5R5A1 = 8
    A1R5G5B5 = 9
    R8 = 10
    R8G8 = 11
    R8G8B8 = 12
    B8G8R8 = 13
    R8G8B8A8 = 14
    B8G8R8A8 = 15
    A8B8G8R8 = 16
    A2R10G10B10 = 17
    A2B10G10R10 = 18
    R16 = 19
    R16G16 = 20
    R16G16B16 = 21
    R16G16B16A16 = 22
    R32 = 23
    R32G32 = 24
    R32G32B32 = 25
    R32G32B32A32 = 26
    B10G11R11_UFloat = 27
    E5B9G9R9_UFloat = 28
    BC1_RGB = 29
    BC1_RGBA = 30
    BC2 = 31
    BC3 = 32
    BC4 = 33
    BC5 = 34
    BC6H = 35
    BC7 = 36
    ETC1 = 37
    ETC2_R8G8B8 = 38
    ETC2_R8G8B8A1 = 39
    ETC2_R8G8B8A8 = 40
    EAC_R11 = 41
    EAC_R11G11 = 42
    ASTC_4x4 = 43
    ASTC_5x4 = 44
    ASTC_5x5 = 45
    ASTC_6x5 = 46
    ASTC_6x6 = 47
    ASTC_8x5 = 48
    ASTC_8x6 = 49
    ASTC_8x8 = 50
    ASTC_10x5 = 51
    ASTC_10x6 = 52
    ASTC_10x8 = 53
    ASTC_10x10 = 54
    ASTC_12x10 = 55
    ASTC_12x12 = 56


class Type(enum.IntEnum):
    """cuttlefish::Texture::Type (Texture.h:135-143)."""
    UNorm = 0
    SNorm = 1
    UInt = 2
    Int = 3
    UFloat = 4
    Float = 5


class Quality(enum.IntEnum):
    """cuttlefish::Texture::Quality (Texture.h:181-188)."""
    Lowest = 0
    Low = 1
    Normal = 2
    High = 3
    Highest = 4


class Alpha(enum.IntEnum):
    """cuttlefish::Texture::Alpha (Texture.h:161-167)."""
    None_ = 0
    Standard = 1
    PreMultiplied = 2
    Encoded = 3


class ColorSpace(enum.IntEnum):
    """cuttlefish::ColorSpace (Color.h:40-44)."""
    Linear = 0
    sRGB = 1


class ResizeFilter(enum.IntEnum):
    """cuttlefish::Image::ResizeFilter (Image.h:79-86)."""
    Box = 0
    Linear = 1
    Cubic = 2
    CatmullRom = 3
    BSpline = 4


class PixelType(enum.IntEnum):
    RGBA8 = 0
    RGBA32F = 1
    RGBA16F = 2


E_INVALID, E_UNSUPPORTED, E_CAPACITY, E_DEVICE, E_NO_DEVICE = -1, -2, -3, -4, -5

# cfhip_consumed_fn: void (*)(void* user, size_t surface_index)
CONSUMED_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t)

EXPORTS = ["cfhip_abi_version", "cfhip_device_count", "cfhip_create", "cfhip_destroy",
           "cfhip_query", "cfhip_encode", "cfhip_encode_multi", "cfhip_encode_multi_ex", "cfhip_encode_device",
           "cfhip_shard_rows",
           "cfhip_last_kernel_ms", "cfhip_last_kernel_name", "cfhip_last_error", "cfhip_pinned_bytes",
           "cfhip_profile_begin", "cfhip_profile_end", "cfhip_generate_mips_device",
           "cfhip_generate_mips3d_device", "cfhip_resize_device", "cfhip_generate_mips_array_device"]


class Params(ctypes.Structure):
    _fields_ = [("format", ctypes.c_int32), ("type", ctypes.c_int32), ("quality", ctypes.c_int32),
                ("alpha", ctypes.c_int32), ("mask_rgba", ctypes.c_uint8 * 4),
                ("color_space", ctypes.c_int32)]


class Surface(ctypes.Structure):
    _fields_ = [("pixels", ctypes.c_void_p), ("pixel_type", ctypes.c_int32),
                ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("row_pitch_bytes", ctypes.c_ssize_t), ("out", ctypes.c_void_p),
                ("out_capacity", ctypes.c_size_t)]


class CfhipError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__("cfhip error %d: %s" % (code, text))
        self.code = code


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libcuttlefish_hip.so.  torch (if importable) is imported first so the
    process holds ONE HIP runtime (torch bundles libamdhip64.so.7 with the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("CFHIP_LIB") or LIB_PATH   # CFHIP_LIB: A/B kernel variants
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the product has no CPU fallback)" % path)
    try:  # pragma: no cover - depends on environment
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    L.cfhip_abi_version.restype = ctypes.c_int
    L.cfhip_device_count.restype = ctypes.c_int
    L.cfhip_create.restype = ctypes.c_void_p
    L.cfhip_create.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_int)]
    L.cfhip_destroy.argtypes = [ctypes.c_void_p]
    L.cfhip_destroy.restype = None
    L.cfhip_query.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3
    L.cfhip_query.restype = ctypes.c_int
    L.cfhip_encode.argtypes = [ctypes.c_void_p, ctypes.POINTER(Surface), ctypes.c_size_t,
                               ctypes.POINTER(Params)]
    L.cfhip_encode.restype = ctypes.c_int
    L.cfhip_encode_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(Surface),
                                     ctypes.c_size_t, ctypes.POINTER(Params)]
    L.cfhip_encode_multi.restype = ctypes.c_int
    L.cfhip_encode_multi_ex.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(Surface),
                                        ctypes.c_size_t, ctypes.POINTER(Params), CONSUMED_FN, ctypes.c_void_p]
    L.cfhip_encode_multi_ex.restype = ctypes.c_int
    L.cfhip_encode_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(Surface), ctypes.c_size_t,
                                      ctypes.POINTER(Params), ctypes.c_void_p]
    L.cfhip_encode_device.restype = ctypes.c_int
    L.cfhip_shard_rows.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_uint32),
                                   ctypes.POINTER(ctypes.c_uint32)]
    L.cfhip_shard_rows.restype = ctypes.c_int
    L.cfhip_last_kernel_ms.argtypes = [ctypes.c_void_p]
    L.cfhip_last_kernel_ms.restype = ctypes.c_float
    L.cfhip_pinned_bytes.argtypes = [ctypes.c_void_p]
    L.cfhip_pinned_bytes.restype = ctypes.c_size_t
    L.cfhip_last_kernel_name.argtypes = [ctypes.c_void_p]
    L.cfhip_last_kernel_name.restype = ctypes.c_char_p
    L.cfhip_profile_begin.argtypes = [ctypes.c_void_p]
    L.cfhip_profile_begin.restype = ctypes.c_int
    L.cfhip_profile_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float),
                                    ctypes.POINTER(ctypes.c_uint32)]
    L.cfhip_profile_end.restype = ctypes.c_int
    L.cfhip_last_error.argtypes = [ctypes.c_void_p]
    L.cfhip_last_error.restype = ctypes.c_char_p
    L.cfhip_generate_mips_device.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
        ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
        ctypes.c_uint32, ctypes.c_void_p]
    L.cfhip_generate_mips_device.restype = ctypes.c_int
    L.cfhip_generate_mips_array_device.argtypes = [
        ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32,
        ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
        ctypes.c_uint32, ctypes.c_void_p]
    L.cfhip_generate_mips_array_device.restype = ctypes.c_int
    L.cfhip_generate_mips3d_device.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
        ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
        ctypes.c_uint32, ctypes.c_void_p]
    L.cfhip_generate_mips3d_device.restype = ctypes.c_int
    L.cfhip_resize_device.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_size_t,
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    L.cfhip_resize_device.restype = ctypes.c_int
    _lib = L
    return L


def make_params(fmt, typ=Type.UNorm, quality=Quality.Normal, alpha=Alpha.Standard,
                color_mask: Sequence[bool] = (True, True, True, True),
                color_space=ColorSpace.Linear) -> Params:
    p = Params()
    p.format, p.type, p.quality = int(fmt), int(typ), int(quality)
    p.alpha, p.color_space = int(alpha), int(color_space)
    for i in range(4):
        p.mask_rgba[i] = 1 if color_mask[i] else 0
    return p


def query(fmt, typ=Type.UNorm):
    """(block_w, block_h, block_bytes) = Texture::blockWidth/Height/Size; raises on the
    (format, type) pairs createConverter rejects (Converter.cpp:339-412)."""
    bw, bh, bs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = load_library().cfhip_query(int(fmt), int(typ), bw, bh, bs)
    if rc != 0:
        raise CfhipError(rc, "illegal (format, type) = (%r, %r)" % (fmt, typ))
    return bw.value, bh.value, bs.value


def payload_size(fmt, typ, width: int, height: int) -> int:
    bw, bh, bs = query(fmt, typ)
    return ((width + bw - 1) // bw) * ((height + bh - 1) // bh) * bs


def shard_rows(block_rows: int, rank: int, world: int):
    a, b = ctypes.c_uint32(), ctypes.c_uint32()
    rc = load_library().cfhip_shard_rows(block_rows, rank, world, a, b)
    if rc != 0:
        raise CfhipError(rc, "bad shard arguments")
    return a.value, b.value


def pixel_type_of(arr: np.ndarray) -> PixelType:
    if arr.dtype == np.uint8:
        return PixelType.RGBA8
    if arr.dtype == np.float32:
        return PixelType.RGBA32F
    if arr.dtype == np.float16:
        return PixelType.RGBA16F
    raise TypeError("pixel dtype %s not accepted at the boundary" % arr.dtype)


class Context:
    """One encoder context = one GPU (one process per GPU in multi-GPU jobs)."""

    def __init__(self, device_id: int = 0):
        self._lib = load_library()
        err = ctypes.c_int(0)
        self._h = self._lib.cfhip_create(device_id, 0, ctypes.byref(err))
        if not self._h:
            raise CfhipError(err.value, self._lib.cfhip_last_error(None).decode())
        self.device_id = device_id

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cfhip_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise CfhipError(rc, self._lib.cfhip_last_error(self._h).decode())

    def encode(self, images: Iterable[np.ndarray], params: Params):
        """Host-buffer path (what HipConverter::process calls): list of (h, w, 4) arrays
        -> list of payload byte arrays."""
        surf, outs, keep = self._host_surfaces(images, params)
        self._check(self._lib.cfhip_encode(self._h, surf, len(outs), ctypes.byref(params)))
        return outs

    def _host_surfaces(self, images, params):
        images = [np.asarray(im) for im in images]
        surf = (Surface * len(images))()
        outs, keep = [], []
        for i, im in enumerate(images):
            if im.ndim != 3 or im.shape[2] != 4:
                raise ValueError("surface %d: expected (h, w, 4)" % i)
            if im.strides[2] != im.itemsize or im.strides[1] != 4 * im.itemsize:
                im = np.ascontiguousarray(im)
            keep.append(im)
            h, w = im.shape[:2]
            out = np.zeros(payload_size(params.format, params.type, w, h), np.uint8)
            outs.append(out)
            surf[i].pixels = im.ctypes.data
            surf[i].pixel_type = int(pixel_type_of(im))
            surf[i].width, surf[i].height = w, h
            surf[i].row_pitch_bytes = im.strides[0]
            surf[i].out = out.ctypes.data
            surf[i].out_capacity = out.nbytes
        return surf, outs, keep

    def encode_multi(self, others: Sequence["Context"], images: Iterable[np.ndarray], params: Params,
                     consumed=None):
        """cfhip_encode_multi: this context plus `others` (one per GPU of this process) share the
        surfaces of the call by block count; same payloads as encode().  consumed(index): called (possibly
        from the contexts' worker threads) once the library has finished reading surface `index`
        (cfhip_encode_multi_ex: the hook through which HipConverter releases each source image)."""
        ctxs = [self] + list(others)
        surf, outs, keep = self._host_surfaces(images, params)
        arr = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
        if consumed is not None:
            cb = CONSUMED_FN(lambda user, i: consumed(int(i)))
            rc = self._lib.cfhip_encode_multi_ex(arr, len(ctxs), surf, len(outs), ctypes.byref(params), cb, None)
        else:
            rc = self._lib.cfhip_encode_multi(arr, len(ctxs), surf, len(outs), ctypes.byref(params))
        if rc != 0:
            for c in ctxs:
                c._check(rc)
        return outs

    def encode_device(self, surfaces: Sequence[dict], params: Params, stream: int = 0):
        """Device-buffer path.  surfaces: dicts with pixels (device ptr int), pixel_type,
        width, height, row_pitch_bytes, out (device ptr int), out_capacity."""
        surf = (Surface * len(surfaces))()
        for i, s in enumerate(surfaces):
            surf[i].pixels = s["pixels"]
            surf[i].pixel_type = int(s["pixel_type"])
            surf[i].width, surf[i].height = s["width"], s["height"]
            surf[i].row_pitch_bytes = s["row_pitch_bytes"]
            surf[i].out = s["out"]
            surf[i].out_capacity = s["out_capacity"]
        self._check(self._lib.cfhip_encode_device(self._h, surf, len(surfaces),
                                                   ctypes.byref(params),
                                                   ctypes.c_void_p(stream) if stream else None))

    def generate_mips_device(self, src: int, pixel_type, width: int, height: int,
                             row_pitch_bytes: int, dst_levels: Sequence[int],
                             color_space=ColorSpace.Linear, filter=0, stream: int = 0):
        """Texture::generateMipmaps on the GPU: level k (k = 1..len(dst_levels)) of a width x height
        texture into dst_levels[k-1] (device pointers, RGBA32F tightly packed), each level resized
        from the previous one in linear space.  filter: ResizeFilter (0 Box, 1 Linear)."""
        n = len(dst_levels) + 1
        arr = (ctypes.c_void_p * max(len(dst_levels), 1))(*[ctypes.c_void_p(int(p)) for p in dst_levels])
        self._check(self._lib.cfhip_generate_mips_device(
            self._h, ctypes.c_void_p(int(src)), int(pixel_type), width, height, row_pitch_bytes,
            int(color_space), int(filter), arr, n, ctypes.c_void_p(stream) if stream else None))

    def generate_mips_array_device(self, srcs: Sequence[int], pixel_type, width: int, height: int,
                                   row_pitch_bytes: int, dst_levels: Sequence[Sequence[int]],
                                   color_space=ColorSpace.Linear, filter=0, stream: int = 0):
        """generate_mips_device for the layers of an array / cube texture in one call: srcs[l] = level 0
        of layer l, dst_levels[l][k-1] receives its level k.  One launch per pass and level for all
        layers; bit-identical to one call per layer."""
        nl = len(srcs)
        if nl == 0 or len(dst_levels) != nl or len({len(d) for d in dst_levels}) != 1:
            raise ValueError("srcs and dst_levels must list the same layers, every layer the same levels")
        per = len(dst_levels[0])
        sarr = (ctypes.c_void_p * nl)(*[ctypes.c_void_p(int(p)) for p in srcs])
        flat = [ctypes.c_void_p(int(p)) for d in dst_levels for p in d]
        darr = (ctypes.c_void_p * max(len(flat), 1))(*flat)
        self._check(self._lib.cfhip_generate_mips_array_device(
            self._h, sarr, nl, int(pixel_type), width, height, row_pitch_bytes, int(color_space), int(filter),
            darr, per + 1, ctypes.c_void_p(stream) if stream else None))

    def generate_mips3d_device(self, src: int, pixel_type, width: int, height: int, depth: int,
                               row_pitch_bytes: int, slice_pitch_bytes: int, dst_levels: Sequence[int],
                               color_space=ColorSpace.Linear, filter=0, stream: int = 0):
        """Texture::generateMipmaps for a 3-D texture on the GPU: level k into dst_levels[k-1] as
        max(1, depth >> k) tightly packed RGBA32F slices."""
        n = len(dst_levels) + 1
        arr = (ctypes.c_void_p * max(len(dst_levels), 1))(*[ctypes.c_void_p(int(p)) for p in dst_levels])
        self._check(self._lib.cfhip_generate_mips3d_device(
            self._h, ctypes.c_void_p(int(src)), int(pixel_type), width, height, depth, row_pitch_bytes,
            slice_pitch_bytes, int(color_space), int(filter), arr, n,
            ctypes.c_void_p(stream) if stream else None))

    def resize_device(self, src: int, pixel_type, width: int, height: int, row_pitch_bytes: int,
                      dst: int, dst_width: int, dst_height: int, color_space=ColorSpace.Linear,
                      filter=0, stream: int = 0):
        """Image::resize on the GPU: src (device pointer) -> dst (device pointer, dst_width x
        dst_height RGBA32F tightly packed), in linear space."""
        self._check(self._lib.cfhip_resize_device(
            self._h, ctypes.c_void_p(int(src)), int(pixel_type), width, height, row_pitch_bytes,
            int(color_space), int(filter), ctypes.c_void_p(int(dst)), dst_width, dst_height,
            ctypes.c_void_p(stream) if stream else None))

    def last_kernel_ms(self) -> float:
        return float(self._lib.cfhip_last_kernel_ms(self._h))

    def profile_begin(self):
        self._check(self._lib.cfhip_profile_begin(self._h))

    def profile_end(self):
        """-> (summed kernel ms, launches) since profile_begin (hipEvents on the launch stream)."""
        ms, n = ctypes.c_float(), ctypes.c_uint32()
        self._check(self._lib.cfhip_profile_end(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def pinned_bytes(self) -> int:
        """page-locked host memory the context holds for its host path (source strip slots + payload landing ring)"""
        return int(self._lib.cfhip_pinned_bytes(self._h))

    def last_kernel_name(self) -> str:
        return self._lib.cfhip_last_kernel_name(self._h).decode()


def device_count() -> int:
    return int(load_library().cfhip_device_count())
