"""ctypes binding of tools/mesa_ref/libmesa_decode.so -- TEST INFRASTRUCTURE.

An independent decoder for every block format on the path: Mesa 23.2.1's software texture
decompression behind an off-screen llvmpipe GL context (tools/mesa_ref/mesa_decode.c).  Present
in this image as a system library (libgl1-mesa-dri); `available()` is False anywhere it is not.
Used to generate tests/golden/mesa_*.npz and to cross-check live encoder output.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "mesa_ref", "mesa_decode.c")
LIB = os.path.join(ROOT, "tools", "mesa_ref", "libmesa_decode.so")
DRIVER = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"

GL_RED, GL_RG, GL_RGB, GL_RGBA = 0x1903, 0x8227, 0x1907, 0x1908
GL_UNSIGNED_BYTE, GL_BYTE, GL_UNSIGNED_SHORT, GL_SHORT, GL_FLOAT, GL_HALF_FLOAT = \
    0x1401, 0x1400, 0x1403, 0x1402, 0x1406, 0x140B

# Texture::Format value -> GL internal format (linear / sRGB where one exists)
GLFMT = {
    29: 0x83F0, 30: 0x83F1, 31: 0x83F2, 32: 0x83F3,           # S3TC DXT1 RGB / RGBA, DXT3, DXT5
    33: 0x8DBB, 34: 0x8DBD,                                   # RGTC1 / RGTC2 (snorm: +1)
    35: 0x8E8F, 36: 0x8E8C,                                   # BPTC unsigned float, BPTC unorm
    37: 0x9274, 38: 0x9274, 39: 0x9276, 40: 0x9278,           # ETC1 (= ETC2 RGB subset), ETC2
    41: 0x9270, 42: 0x9272,                                   # EAC R11 / RG11 (signed: +1)
}
for _i in range(14):
    GLFMT[43 + _i] = 0x93B0 + _i                              # ASTC 4x4 .. 12x12 (sRGB: +0x20)
GL_BPTC_SIGNED_FLOAT = 0x8E8E

_lib = None
_state = None


def available() -> bool:
    global _lib, _state
    if _state is not None:
        return _state
    _state = False
    if not os.path.exists(DRIVER) or not os.path.exists("/usr/include/GL/internal/dri_interface.h") \
            and not os.path.exists(LIB):
        return False
    try:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
            subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", LIB, SRC, "-ldl"])
        L = ctypes.CDLL(LIB)
        L.mesa_init.argtypes = [ctypes.c_char_p]
        L.mesa_decode.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
        L.mesa_version.restype = ctypes.c_char_p
        os.environ.setdefault("LP_NUM_THREADS", "1")
        if L.mesa_init(DRIVER.encode()) != 0:
            return False
        _lib = L
        _state = True
    except Exception:
        _state = False
    return _state


def version() -> str:
    return _lib.mesa_version().decode() if available() else ""


def decode_gl(glfmt: int, payload: np.ndarray, width: int, height: int, rb_format=GL_RGBA,
              rb_type=GL_UNSIGNED_BYTE) -> np.ndarray:
    assert available()
    nch = {GL_RED: 1, GL_RG: 2, GL_RGB: 3, GL_RGBA: 4}[rb_format]
    dt = {GL_UNSIGNED_BYTE: np.uint8, GL_BYTE: np.int8, GL_UNSIGNED_SHORT: np.uint16,
          GL_SHORT: np.int16, GL_FLOAT: np.float32, GL_HALF_FLOAT: np.float16}[rb_type]
    payload = np.ascontiguousarray(payload, np.uint8)
    out = np.zeros((height, width, nch), dt)
    rc = _lib.mesa_decode(glfmt, width, height, payload.ctypes.data, payload.nbytes, rb_format,
                          rb_type, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("GL error 0x%x decoding format 0x%x" % (rc, glfmt))
    return out


def decode(fmt: int, payload: np.ndarray, width: int, height: int, typ: int = 0, srgb=False,
           rb_format=GL_RGBA, rb_type=GL_UNSIGNED_BYTE) -> np.ndarray:
    g = GLFMT[int(fmt)]
    if int(fmt) in (33, 34, 41, 42) and typ == 1:
        g += 1
    if int(fmt) == 35 and typ == 5:
        g = GL_BPTC_SIGNED_FLOAT
    if srgb and int(fmt) >= 43:
        g += 0x20
    return decode_gl(g, payload, width, height, rb_format, rb_type)


def encode(fmt: int, img: np.ndarray, typ: int = 0) -> np.ndarray:
    """Mesa's OWN software encoder (an independent encoder, not ours): (h, w, 4) uint8 RGBA -- or
    float32 for BC6H -- through glTexImage2D with a compressed internal format, payload back through
    glGetCompressedTexImage.  S3TC (29..32), RGTC (33, 34), BPTC (35, 36)."""
    assert available()
    _lib.mesa_encode.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    _lib.mesa_encode.restype = ctypes.c_int
    g = GLFMT[int(fmt)]
    if int(fmt) in (33, 34) and typ == 1:
        g += 1
    if int(fmt) == 35 and typ == 5:
        g = GL_BPTC_SIGNED_FLOAT
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    src_type = GL_FLOAT if img.dtype == np.float32 else GL_UNSIGNED_BYTE
    cap = ((w + 3)//4)*((h + 3)//4)*16
    out = np.zeros(cap, np.uint8)
    rc = _lib.mesa_encode(g, w, h, GL_RGBA, src_type, img.ctypes.data, out.ctypes.data, cap)
    if rc <= 0:
        raise RuntimeError("Mesa could not compress format %d (GL error / not compressed: %d)" % (fmt, rc))
    return out[:rc].copy()
