"""ctypes binding of oracle/libcf_oracle.so -- the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libcf_oracle.so")

PIX_RGBA8, PIX_RGBA32F, PIX_RGBA16F = 0, 1, 2


class Params(ctypes.Structure):
    _fields_ = [("format", ctypes.c_int), ("type", ctypes.c_int), ("quality", ctypes.c_int),
                ("alpha", ctypes.c_int), ("mask", ctypes.c_uint8 * 4),
                ("color_space", ctypes.c_int)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR)
            if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = (not os.path.exists(LIB_PATH) or
             any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libcf_oracle.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        L.cfo_encode.restype = ctypes.c_int
        L.cfo_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
                                 ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.POINTER(Params), ctypes.c_uint]
        L.cfo_decode.restype = ctypes.c_int
        L.cfo_decode.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                                 ctypes.c_uint32, ctypes.c_void_p]
        L.cfo_block_info.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3
        _lib = L
    return _lib


def make_params(fmt, typ=0, quality=2, alpha=1, mask=(1, 1, 1, 1), color_space=0) -> Params:
    p = Params()
    p.format, p.type, p.quality, p.alpha, p.color_space = int(fmt), int(typ), int(quality), \
        int(alpha), int(color_space)
    for i in range(4):
        p.mask[i] = 1 if mask[i] else 0
    return p


def block_bytes(fmt) -> int:
    bw, bh, bs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if lib().cfo_block_info(int(fmt), bw, bh, bs) != 0:
        raise ValueError("unsupported format %r" % (fmt,))
    return bs.value


def _pixel_type(img: np.ndarray) -> int:
    if img.dtype == np.uint8:
        return PIX_RGBA8
    if img.dtype == np.float32:
        return PIX_RGBA32F
    if img.dtype == np.float16:
        return PIX_RGBA16F
    raise TypeError(img.dtype)


def encode(img: np.ndarray, fmt, typ=0, quality=2, threads=1, **kw) -> np.ndarray:
    """img: (h, w, 4) uint8 / float32 / float16, C-contiguous rows."""
    assert img.ndim == 3 and img.shape[2] == 4
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    bw, bh, bs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if lib().cfo_block_info(int(fmt), bw, bh, bs) != 0:
        raise ValueError("unsupported format %r" % (fmt,))
    n = ((w + bw.value - 1) // bw.value) * ((h + bh.value - 1) // bh.value) * bs.value
    out = np.zeros(n, np.uint8)
    p = make_params(fmt, typ, quality, **kw)
    rc = lib().cfo_encode(img.ctypes.data, _pixel_type(img), w, h, img.strides[0],
                          out.ctypes.data, out.nbytes, ctypes.byref(p), threads)
    if rc != 0:
        raise RuntimeError("cfo_encode failed: %d" % rc)
    return out


def decode_bc6h(blocks: np.ndarray, width: int, height: int, typ=4) -> np.ndarray:
    """-> (h, w, 3) float16 (typ 4 = UFloat, 5 = Float)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    out = np.zeros((height, width, 3), np.uint16)
    L = lib()
    L.cfo_decode_bc6h_image.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32,
                                        ctypes.c_uint32, ctypes.c_void_p]
    rc = L.cfo_decode_bc6h_image(blocks.ctypes.data, int(typ), width, height, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("cfo_decode_bc6h_image failed: %d" % rc)
    return out.view(np.float16)


def decode_etc(blocks: np.ndarray, fmt, width: int, height: int) -> np.ndarray:
    """ETC1 / ETC2 RGB / RGBA1 / RGBA8 (fmt 37..40) -> (h, w, 4) uint8."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    out = np.zeros((height, width, 4), np.uint8)
    L = lib()
    L.cfo_decode_etc_image.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_void_p]
    if L.cfo_decode_etc_image(int(fmt), blocks.ctypes.data, width, height, out.ctypes.data) != 0:
        raise RuntimeError("cfo_decode_etc_image failed")
    return out


def decode_astc(blocks: np.ndarray, fmt, width: int, height: int):
    """ASTC (fmt 43..56), emitted subset only -> ((h, w, 4) uint8, blocks outside the subset)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    out = np.zeros((height, width, 4), np.uint8)
    L = lib()
    L.cfo_decode_astc_image.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                                        ctypes.c_uint32, ctypes.c_void_p]
    bad = L.cfo_decode_astc_image(int(fmt), blocks.ctypes.data, width, height, out.ctypes.data)
    if bad < 0:
        raise RuntimeError("cfo_decode_astc_image failed")
    return out, bad


def decode_astc_hdr(blocks: np.ndarray, fmt, width: int, height: int):
    """ASTC under the HDR profile -> ((h, w, 4) float16, blocks the decoder does not model)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    out = np.zeros((height, width, 4), np.uint16)
    L = lib()
    L.cfo_decode_astc_image_hdr.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                                            ctypes.c_uint32, ctypes.c_void_p]
    bad = L.cfo_decode_astc_image_hdr(int(fmt), blocks.ctypes.data, width, height, out.ctypes.data)
    if bad < 0:
        raise RuntimeError("cfo_decode_astc_image_hdr failed")
    return out.view(np.float16), bad


def decode_eac(blocks: np.ndarray, fmt, width: int, height: int, typ=0) -> np.ndarray:
    """EAC R11 / RG11 (fmt 41/42) -> (h, w, nch) int32 (0..2047 or -1023..1023)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    nch = 2 if int(fmt) == 42 else 1
    out = np.zeros((height, width, nch), np.int32)
    L = lib()
    L.cfo_decode_eac_image.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    if L.cfo_decode_eac_image(int(fmt), int(typ), blocks.ctypes.data, width, height,
                              out.ctypes.data) != 0:
        raise RuntimeError("cfo_decode_eac_image failed")
    return out


def decode(blocks: np.ndarray, fmt, width: int, height: int, typ=0) -> np.ndarray:
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    out = np.zeros((height, width, 4), np.uint8)
    rc = lib().cfo_decode(int(fmt), int(typ), blocks.ctypes.data, width, height, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("cfo_decode failed: %d" % rc)
    return out


# ---- mip-level resize (oracle/mipgen.c) --------------------------------------------------

FILTER_FALLBACK = 0x100      # CFO_FILTER_FALLBACK: Box / Linear through the in-tree loops of Image.cpp:1393-1505


def resize_rgbaf(img: np.ndarray, width: int, height: int, filter=0, color_space=0) -> np.ndarray:
    """Image::resize of an RGBAF image (h, w, 4) float32 -> (height, width, 4) float32."""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape[:2]
    out = np.zeros((height, width, 4), np.float32)
    L = lib()
    L.cfo_resize_rgbaf.restype = ctypes.c_int
    L.cfo_resize_rgbaf.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p,
                                   ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    rc = L.cfo_resize_rgbaf(img.ctypes.data, w, h, out.ctypes.data, width, height, int(filter),
                            int(color_space))
    if rc != 0:
        raise RuntimeError("cfo_resize_rgbaf failed: %d" % rc)
    return out


def mip_chain(img: np.ndarray, levels: int, filter=0, color_space=0):
    """Texture::generateMipmaps (2-D): [level 0 as RGBAF, level 1, ...], level k from level k-1."""
    if img.dtype == np.uint8:
        base = (img.astype(np.float64) / 255.0).astype(np.float32)     # toDoubleNorm, float store
    else:
        base = img.astype(np.float32)
    h, w = base.shape[:2]
    out = [base]
    for k in range(1, levels):
        out.append(resize_rgbaf(out[-1], max(1, w >> k), max(1, h >> k), filter, color_space))
    return out


def mip_depth_pass(prev: np.ndarray, depth: int, filter=0, color_space=0) -> np.ndarray:
    """generateMips3d (Texture.cpp:103-227): (n_prev, h, w, 4) float32 slices -> (depth, h, w, 4)."""
    prev = np.ascontiguousarray(prev, np.float32)
    n, h, w = prev.shape[:3]
    out = np.zeros((depth, h, w, 4), np.float32)
    L = lib()
    L.cfo_mip_depth_pass.restype = ctypes.c_int
    L.cfo_mip_depth_pass.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                     ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
    rc = L.cfo_mip_depth_pass(prev.ctypes.data, n, w, h, out.ctypes.data, depth, int(filter), int(color_space))
    if rc != 0:
        raise RuntimeError("cfo_mip_depth_pass failed: %d" % rc)
    return out


def mip_chain3d(vol: np.ndarray, levels: int, filter=0, color_space=0):
    """Texture::generateMipmaps, Dim3D branch (Texture.cpp:1345-1440): vol (d, h, w, 4); every slice
    of level k-1 is resized in x, y, then the depth pass."""
    if vol.dtype == np.uint8:
        base = (vol.astype(np.float64) / 255.0).astype(np.float32)
    else:
        base = vol.astype(np.float32)
    d0, h0, w0 = base.shape[:3]
    out = [base]
    for k in range(1, levels):
        w, h, d = max(1, w0 >> k), max(1, h0 >> k), max(1, d0 >> k)
        prev = out[-1]
        resized = np.stack([resize_rgbaf(prev[i], w, h, filter, color_space) for i in range(prev.shape[0])])
        out.append(mip_depth_pass(resized, d, filter, color_space))
    return out


def color_fns():
    L = lib()
    for n in ("cfo_srgb_to_linear", "cfo_linear_to_srgb"):
        getattr(L, n).restype = ctypes.c_double
        getattr(L, n).argtypes = [ctypes.c_double]
    return L.cfo_srgb_to_linear, L.cfo_linear_to_srgb


REF_LIB_PATH = os.path.join(ORACLE_DIR, "_ref", "libcf_ref.so")


def ref_lib():
    """oracle/_ref/libcf_ref.so: the reference's own Color.h compiled (`make -C oracle ref`, only
    where /root/reference is mounted; the built file travels to the GPU box).  None if absent."""
    if not os.path.exists(REF_LIB_PATH):
        return None
    R = ctypes.CDLL(REF_LIB_PATH)
    for n in ("cfref_srgb_to_linear", "cfref_linear_to_srgb"):
        getattr(R, n).restype = ctypes.c_double
        getattr(R, n).argtypes = [ctypes.c_double]
    R.cfref_to_grayscale.restype = ctypes.c_double
    R.cfref_to_grayscale.argtypes = [ctypes.c_double] * 3
    return R


# ---- uncompressed ("standard") converters (oracle/std_pack.c) ------------------------------

def std_pixel_bytes(fmt: int, typ: int) -> int:
    L = lib()
    L.cfo_std_pixel_bytes.restype = ctypes.c_int
    L.cfo_std_pixel_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    return L.cfo_std_pixel_bytes(int(fmt), int(typ))


def as_rgbaf(img: np.ndarray) -> np.ndarray:
    """The reference's RGBAF view of a source image: u8/255 (toDoubleNorm + float store, which for
    8-bit values equals the float division), half -> float exact, float as is."""
    if img.dtype == np.uint8:
        return (img.astype(np.float64)/255.0).astype(np.float32)
    return np.ascontiguousarray(img.astype(np.float32))


def std_pack(img: np.ndarray, fmt: int, typ: int) -> np.ndarray:
    """StandardConverter family: (h, w, 4) image -> h*w*bytes_per_pixel payload bytes."""
    f = np.ascontiguousarray(as_rgbaf(img))
    h, w = f.shape[:2]
    bpp = std_pixel_bytes(fmt, typ)
    if not bpp:
        raise ValueError("illegal (format, type) = (%d, %d)" % (fmt, typ))
    out = np.zeros(h*w*bpp, np.uint8)
    L = lib()
    L.cfo_std_pack.restype = ctypes.c_int
    L.cfo_std_pack.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                               ctypes.c_uint32, ctypes.c_ssize_t, ctypes.c_void_p, ctypes.c_size_t]
    rc = L.cfo_std_pack(int(fmt), int(typ), f.ctypes.data, w, h, f.strides[0], out.ctypes.data, out.nbytes)
    if rc != 0:
        raise RuntimeError("cfo_std_pack failed: %d" % rc)
    return out
