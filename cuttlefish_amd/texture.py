"""Host-side mirror of the reference interface for the block-encode path.

Mirrors, for this path only, ``cuttlefish::Texture`` as the reference's callers
and tests use it (lib/include/cuttlefish/Texture.h:42-836):

    Texture(width, height)            Texture.h ctor (Dim2D)
    set_image(image, mip, depth)      Texture::setImage        Texture.cpp:1252-1318
    convert(format, type, quality, alpha_type, color_mask, threads)
                                      Texture::convert         Texture.cpp:1536-1561
    converted(), format(), type(), data(mip, depth), data_size(mip, depth)

Same argument meaning and error behaviour: ``convert`` returns False when the
images are incomplete, the (format, type) pair is illegal (isFormatValid /
createConverter returning nullptr, Converter.cpp:339-412) or the texture is sRGB
and the format has no native sRGB variant (Texture::hasNativeSRGB,
Texture.cpp:421-465).  Everything else about Texture (mip generation, saving,
image processing) is out of scope.

The conversion itself is one call into the C-ABI (``cfhip_encode``) for all
surfaces of the texture -- the whole-surface Converter of INTEGRATION.md.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import api
from .api import Alpha, ColorSpace, Format, Quality, Type

_NATIVE_SRGB = {Format.R8G8B8, Format.B8G8R8, Format.R8G8B8A8, Format.B8G8R8A8, Format.A8B8G8R8,
                Format.BC1_RGB, Format.BC1_RGBA, Format.BC2, Format.BC3, Format.BC7,
                Format.ETC2_R8G8B8, Format.ETC2_R8G8B8A1, Format.ETC2_R8G8B8A8} | \
    {Format(v) for v in range(43, 57)}


class Texture:
    allCores = 0xFFFFFFFF  # Texture::allCores; thread count is meaningless on the GPU path

    def __init__(self, width: int, height: int, depth: int = 0, mip_levels: int = 1,
                 color_space: ColorSpace = ColorSpace.Linear, device_id: int = 0):
        if width <= 0 or height <= 0 or mip_levels <= 0:
            raise ValueError("invalid texture dimensions")
        self._w, self._h, self._depth = width, height, depth
        self._mips = mip_levels
        self._color_space = ColorSpace(color_space)
        self._images: List[List[Optional[np.ndarray]]] = [
            [None] * max(depth, 1) for _ in range(mip_levels)]
        self._textures: List[List[np.ndarray]] = []
        self._format: Optional[Format] = None
        self._type: Optional[Type] = None
        self._alpha = Alpha.Standard
        self._mask = (True, True, True, True)
        self._device_id = device_id
        self._ctx: Optional[api.Context] = None

    # -- geometry ---------------------------------------------------------
    def width(self, mip: int = 0) -> int:
        return max(self._w >> mip, 1)

    def height(self, mip: int = 0) -> int:
        return max(self._h >> mip, 1)

    def mip_level_count(self) -> int:
        return self._mips

    def color_space(self) -> ColorSpace:
        return self._color_space

    # -- images -----------------------------------------------------------
    def set_image(self, image: np.ndarray, mip: int = 0, depth: int = 0) -> bool:
        """Texture::setImage: the image must match the mip's size.  The reference
        converts everything to RGBAF; RGBA8 arrays are accepted too because that is
        what toColorBlock (S3tcConverter.cpp:97-111) makes of them anyway."""
        if self._textures:
            return False
        if not (0 <= mip < self._mips) or not (0 <= depth < max(self._depth, 1)):
            return False
        image = np.asarray(image)
        if image.ndim != 3 or image.shape[2] != 4:
            return False
        if image.shape[0] != self.height(mip) or image.shape[1] != self.width(mip):
            return False
        if image.dtype not in (np.uint8, np.float32, np.float16):
            image = image.astype(np.float32)
        self._images[mip][depth] = image
        return True

    def generate_mipmaps(self, filter=api.ResizeFilter.CatmullRom, mip_levels: Optional[int] = None) -> bool:
        """Texture::generateMipmaps (Texture.cpp:1320-1514, 2-D path): every level from the
        previous one through Image::resize in linear space, on the GPU
        (cfhip_generate_mips_device).  Box / Linear: the in-tree fallback arithmetic; Cubic,
        CatmullRom (default, as in the reference) and BSpline: FreeImage's resampler restated
        (FreeImage is absent: parity unpinned).  Levels come back as RGBAF (float32) images."""
        if self._textures or self._depth or any(im is None for im in self._images[0]):
            return False
        w, h = self._w, self._h
        max_levels = max(w, h).bit_length()
        levels = max_levels if mip_levels is None else min(max(int(mip_levels), 1), max_levels)
        try:
            filter = api.ResizeFilter(filter)
        except ValueError:
            return False
        import torch  # device memory + stream: plumbing only
        if self._ctx is None:
            self._ctx = api.Context(self._device_id)
        base = np.ascontiguousarray(self._images[0][0])
        src = torch.from_numpy(base).to("cuda:%d" % self._device_id)
        dsts = [torch.empty((max(1, h >> k), max(1, w >> k), 4), dtype=torch.float32,
                            device=src.device) for k in range(1, levels)]
        self._ctx.generate_mips_device(src.data_ptr(), api.pixel_type_of(base), w, h,
                                       base.strides[0], [d.data_ptr() for d in dsts],
                                       color_space=self._color_space, filter=int(filter))
        self._mips = levels
        self._images = [[base]] + [[d.cpu().numpy()] for d in dsts]
        return True

    def images_complete(self) -> bool:
        return all(im is not None for level in self._images for im in level)

    # -- conversion -------------------------------------------------------
    def convert(self, format: Format, type: Type, quality: Quality = Quality.Normal,
                alpha_type: Alpha = Alpha.Standard,
                color_mask: Sequence[bool] = (True, True, True, True),
                threads: int = allCores) -> bool:
        del threads  # the GPU path has no thread count (PvrtcConverter-style whole surface)
        if not self.images_complete():
            return False
        try:
            format = Format(format)
            type = Type(type)
            api.query(format, type)
        except (ValueError, api.CfhipError):
            return False
        if self._color_space == ColorSpace.sRGB and not (format in _NATIVE_SRGB and
                                                         type == Type.UNorm):
            return False
        if self._ctx is None:
            self._ctx = api.Context(self._device_id)
        params = api.make_params(format, type, quality, alpha_type, color_mask,
                                 self._color_space)
        flat = [im for level in self._images for im in level]
        # the reference converts every image to RGBAF before it reaches a converter
        # (Converter.h:52-56): half-float images are bit-exact sources for BC6H and the
        # uncompressed packers, every other block kernel takes them as floats
        if not (format == Format.BC6H or int(format) < int(Format.BC1_RGB)):
            flat = [im.astype(np.float32) if im.dtype == np.float16 else im for im in flat]
        try:
            outs = self._ctx.encode(flat, params)
        except api.CfhipError as e:
            if e.code == api.E_UNSUPPORTED:
                return False  # createConverter -> nullptr -> convert() returns false
            raise
        it = iter(outs)
        self._textures = [[next(it) for _ in level] for level in self._images]
        # Converter::convert frees each source image once its surface is done (:586)
        self._images = [[None] * len(level) for level in self._images]
        self._format, self._type = format, type
        self._alpha, self._mask = Alpha(alpha_type), tuple(bool(m) for m in color_mask)
        return True

    def converted(self) -> bool:
        return bool(self._textures)

    def format(self) -> Optional[Format]:
        return self._format

    def type(self) -> Optional[Type]:
        return self._type

    def alpha_type(self) -> Alpha:
        return self._alpha

    def color_mask(self):
        return self._mask

    def data(self, mip: int = 0, depth: int = 0) -> Optional[np.ndarray]:
        if not self._textures:
            return None
        return self._textures[mip][depth]

    def data_size(self, mip: int = 0, depth: int = 0) -> int:
        d = self.data(mip, depth)
        return 0 if d is None else int(d.nbytes)
