"""Host-side mirror of the reference interface for the block-encode path.

Mirrors ``cuttlefish::Texture`` as the reference's callers and tests use it
(lib/include/cuttlefish/Texture.h:42-836, lib/src/Texture.cpp):

    Texture(dimension, width, height, depth, mip_levels, color_space)   Texture::initialize :1136-1163
    dimension / width / height / depth / is_array / mip_level_count / face_count       :1170-1232
    set_image(image, [face,] mip, depth) / get_image([face,] mip, depth)              :1234-1318
    generate_mipmaps(filter, mip_levels, custom_mip_images)                          :1320-1514
    images_complete()                                                                :1516-1534
    convert(format, type, quality, alpha_type, color_mask, threads)                  :1536-1561
    converted / format / type / alpha_type / color_mask / data / data_size           :1563-1634
    save(file_name | None, file_type) -> SaveResult (or bytes)                       :1636-1685
    statics: is_format_valid, has_native_srgb, has_alpha, max_mipmap_levels, block_width /
             block_height / block_size, min_width / min_height, file_type,
             adjust_image_value_range                                              :318-1084

Same argument meaning and error behaviour: ``convert`` returns False when the images are
incomplete, the (format, type) pair is illegal (isFormatValid / createConverter returning nullptr,
Converter.cpp:339-412) or the texture is sRGB and the format has no native sRGB variant
(Texture::hasNativeSRGB, Texture.cpp:421-465); ``save`` returns the reference's SaveResult codes.

What runs where: the conversion is ONE call into the C-ABI (``cfhip_encode``) for all surfaces of
the texture -- every mip, array element, 3-D slice and cube face -- the whole-surface Converter of
INTEGRATION.md; mip generation runs on the GPU (``cfhip_generate_mips_device``,
``cfhip_generate_mips3d_device``, ``cfhip_resize_device``); the containers are serialised on the
host (containers.py).  Images are numpy arrays (h, w, 4): the reference's Image class, its loaders
and pixel operations are out of scope, so ``Image(format, w, h)`` of the
reference's tests is ``np.zeros((h, w, 4), np.float32)`` here.  The short form
``Texture(width, height, ...)`` of earlier rounds is kept (a 2-D texture).
"""
from __future__ import annotations

import enum
import io
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import api, containers
from .api import Alpha, ColorSpace, Format, Quality, Type

_NATIVE_SRGB = {Format.R8G8B8, Format.B8G8R8, Format.R8G8B8A8, Format.B8G8R8A8, Format.A8B8G8R8,
                Format.BC1_RGB, Format.BC1_RGBA, Format.BC2, Format.BC3, Format.BC7,
                Format.ETC2_R8G8B8, Format.ETC2_R8G8B8A1, Format.ETC2_R8G8B8A8} | \
    {Format(v) for v in range(43, 57)}


class Dimension(enum.IntEnum):      # Texture::Dimension (Texture.h:48-54)
    Dim1D = 0
    Dim2D = 1
    Dim3D = 2
    Cube = 3


class CubeFace(enum.IntEnum):       # Texture::CubeFace (Texture.h:148-156)
    PosX = 0
    NegX = 1
    PosY = 2
    NegY = 3
    PosZ = 4
    NegZ = 5


class MipReplacement(enum.IntEnum):  # Texture::MipReplacement (Texture.h:172-176)
    Once = 0        # resume with the previous image when going down the mip chain
    Continue = 1    # continue with the new image


class FileType(enum.IntEnum):       # Texture::FileType (Texture.h:193-199)
    Auto = 0
    DDS = 1
    KTX = 2
    PVR = 3


class SaveResult(enum.IntEnum):     # Texture::SaveResult (Texture.h:204-211)
    Success = 0
    Invalid = 1
    UnknownFormat = 2
    Unsupported = 3
    WriteError = 4


class ImageFormat(enum.IntEnum):    # Image::Format (Image.h:54-74): only names an image's ORIGINAL storage here
    Invalid = 0
    Gray8 = 1
    Gray16 = 2
    RGB5 = 3
    RGB565 = 4
    RGB8 = 5
    RGB16 = 6
    RGBF = 7
    RGBA8 = 8
    RGBA16 = 9
    RGBAF = 10
    Int16 = 11
    UInt16 = 12
    Int32 = 13
    UInt32 = 14
    Float = 15
    Double = 16
    Complex = 17


class CustomMipImage:
    """Texture::CustomMipImage (Texture.h:330-395): an image that replaces a generated mip level."""

    def __init__(self, image, replacement: MipReplacement = MipReplacement.Once):
        self.image = image
        self.replacement = MipReplacement(replacement)


def image_index(*args) -> Tuple[int, int, int]:
    """Texture::ImageIndex (Texture.h:246-290): image_index([face,] mip=0, depth=0) -> the
    (face, mip, depth) key of a custom_mip_images dict."""
    if args and isinstance(args[0], CubeFace):
        face, rest = int(args[0]), args[1:]
    else:
        face, rest = 0, args
    mip = int(rest[0]) if len(rest) > 0 else 0
    depth = int(rest[1]) if len(rest) > 1 else 0
    return (face, mip, depth)


def _as_image(image) -> Optional[np.ndarray]:
    """What Image::convert(RGBAF) keeps of an image on this path: an (h, w, 4) array.  uint8 / float16
    arrays stay as they are (the kernels read them as the reference's RGBAF values, toColorBlock
    S3tcConverter.cpp:97-111 / HalfFloat.h), everything else becomes float32."""
    if image is None:
        return None
    image = np.asarray(image)
    if image.ndim != 3 or image.shape[2] != 4 or image.shape[0] == 0 or image.shape[1] == 0:
        return None
    if image.dtype not in (np.uint8, np.float32, np.float16):
        image = image.astype(np.float32)
    return image


class Texture:
    allMipLevels = 0xFFFFFFFF   # Texture::allMipLevels (Texture.h:405)
    allCores = 0xFFFFFFFF       # Texture::allCores (:410); a thread count is meaningless on the GPU path

    # ---- statics (Texture.cpp:318-957) ---------------------------------------------------------
    @staticmethod
    def is_format_valid(format, type, file_type: Optional[FileType] = None) -> bool:
        """Texture::isFormatValid(format, type[, fileType]) (Texture.cpp:318-419)."""
        try:
            format, type = Format(format), Type(type)
            api.query(format, type)
        except (ValueError, api.CfhipError):
            return False
        if file_type is None:
            return True
        file_type = FileType(file_type)
        if file_type == FileType.DDS:
            return (format, type) in containers._DXGI
        if file_type == FileType.KTX:
            return (format, type) in containers._GL or (format, type) in containers._GLU
        if file_type == FileType.PVR:
            return format in containers._PVR_GENERIC or format in containers._PVR_SPECIAL
        return False

    @staticmethod
    def has_native_srgb(format, type) -> bool:
        """Texture::hasNativeSRGB (Texture.cpp:421-465)."""
        try:
            return Format(format) in _NATIVE_SRGB and Type(type) == Type.UNorm
        except ValueError:
            return False

    @staticmethod
    def has_alpha(format) -> bool:
        return containers.has_alpha(format)

    @staticmethod
    def max_mipmap_levels(dimension, width: int, height: int, depth: int = 0) -> int:
        """Texture::maxMipmapLevels (Texture.cpp:514-527): 32 - clz of the largest extent."""
        big = max(int(width), int(height))
        if Dimension(dimension) == Dimension.Dim3D:
            big = max(big, int(depth))
        return int(big).bit_length()

    @staticmethod
    def _block(format, i) -> int:
        try:
            format = Format(format)
        except ValueError:
            return 0
        for t in Type:
            try:
                return api.query(format, t)[i]
            except api.CfhipError:
                continue
        return 0

    @staticmethod
    def block_width(format) -> int:
        return Texture._block(format, 0)

    @staticmethod
    def block_height(format) -> int:
        return Texture._block(format, 1)

    @staticmethod
    def block_size(format) -> int:
        return Texture._block(format, 2)

    @staticmethod
    def min_width(format) -> int:
        """Texture::minWidth (Texture.cpp:775-855): the block width of the formats of this backend."""
        return Texture._block(format, 0)

    @staticmethod
    def min_height(format) -> int:
        return Texture._block(format, 1)

    @staticmethod
    def file_type(file_name: str) -> FileType:
        """Texture::fileType (Texture.cpp:939-957): by extension, case-insensitive."""
        low = str(file_name).lower()
        for ext, ft in ((".dds", FileType.DDS), (".ktx", FileType.KTX), (".pvr", FileType.PVR)):
            if low.endswith(ext):
                return ft
        return FileType.Auto

    @staticmethod
    def adjust_image_value_range(image, type, orig_image_format=ImageFormat.Invalid) -> Optional[np.ndarray]:
        """Texture::adjustImageValueRange (Texture.cpp:959-1084): what the reference's front end does
        to an image that came from an integer file format before a SNorm / UInt / Int conversion --
        SNorm remaps [0, 1] to [-1, 1] (v*2 - 1 in float), UInt scales to the original integer range
        (round(v*max)), Int also offsets by the type's minimum.  Images from float formats, and
        UNorm / UFloat / Float conversions, are returned unchanged.  image: (h, w, C) array; uint8 /
        uint16 arrays are read as v/255, v/65535 and name their own original format."""
        if image is None:
            return None
        image = np.asarray(image)
        fmt = ImageFormat(orig_image_format)
        if fmt == ImageFormat.Invalid:
            if image.dtype == np.uint8:
                fmt = {1: ImageFormat.Gray8, 3: ImageFormat.RGB8}.get(image.shape[-1], ImageFormat.RGBA8)
            elif image.dtype == np.uint16:
                fmt = {1: ImageFormat.Gray16, 3: ImageFormat.RGB16}.get(image.shape[-1], ImageFormat.RGBA16)
            else:
                fmt = ImageFormat.RGBAF
        if image.dtype == np.uint8:
            out = (image.astype(np.float64)/255.0).astype(np.float32)
        elif image.dtype == np.uint16:
            out = (image.astype(np.float64)/65535.0).astype(np.float32)
        else:
            out = image.astype(np.float32)
        type = Type(type)
        integer_origin = fmt in (ImageFormat.Gray8, ImageFormat.Gray16, ImageFormat.RGB5, ImageFormat.RGB565,
                                 ImageFormat.RGB8, ImageFormat.RGB16, ImageFormat.RGBA8, ImageFormat.RGBA16)
        if type not in (Type.SNorm, Type.UInt, Type.Int) or not integer_origin:
            return out if out is not image else out.copy()
        if type == Type.SNorm:
            return out*np.float32(2.0) - np.float32(1.0)
        if fmt in (ImageFormat.Gray8, ImageFormat.RGB8, ImageFormat.RGBA8):
            mul, off = [255.0]*4, [-128.0]*4
        elif fmt in (ImageFormat.Gray16, ImageFormat.RGB16, ImageFormat.RGBA16):
            mul, off = [65535.0]*4, [-32768.0]*4
        elif fmt == ImageFormat.RGB5:
            mul, off = [31.0, 31.0, 31.0, 0.0], [-16.0, -16.0, -16.0, 0.0]
        else:                                                    # RGB565
            mul, off = [31.0, 63.0, 31.0, 0.0], [-16.0, -32.0, -16.0, 0.0]
        c = out.shape[-1]
        m = np.array(mul[:c], np.float32)
        o = np.array(off[:c] if type == Type.Int else [0.0]*c, np.float32)
        v = out*m + o
        return (np.sign(v)*np.floor(np.abs(v) + np.float32(0.5))).astype(np.float32)    # std::round

    # ---- construction ---------------------------------------------------------------------------
    def __init__(self, *args, depth: int = 0, mip_levels: int = 1,
                 color_space: ColorSpace = ColorSpace.Linear, device_id: int = 0):
        """Texture(dimension, width, height, depth=0, mip_levels=1, color_space=Linear), the
        reference's constructor (Texture.h:536-538); Texture(width, height, depth=0, ...) is a 2-D
        texture; Texture() is invalid until initialize()."""
        self._device_id = device_id
        self._ctx: Optional[api.Context] = None
        self._valid = False
        self._reset_state()
        if not args:
            return
        if isinstance(args[0], Dimension):
            dimension, rest = args[0], list(args[1:])
        else:
            dimension, rest = Dimension.Dim2D, list(args)
        if len(rest) < 2:
            raise TypeError("Texture needs a width and a height")
        width, height = rest[0], rest[1]
        if len(rest) > 2:
            depth = rest[2]
        if len(rest) > 3:
            mip_levels = rest[3]
        if len(rest) > 4:
            color_space = rest[4]
        self.initialize(dimension, width, height, depth, mip_levels, color_space)

    def _reset_state(self):
        self._dim = Dimension.Dim2D
        self._w = self._h = self._depth = 0
        self._mips = 0
        self._faces = 0
        self._color_space = ColorSpace.Linear
        self._images: List[List[List[Optional[np.ndarray]]]] = []     # [mip][depth][face]
        self._textures: List[List[List[np.ndarray]]] = []
        self._format: Optional[Format] = None
        self._type: Optional[Type] = None
        self._alpha = Alpha.Standard
        self._mask = (True, True, True, True)

    def initialize(self, dimension, width: int, height: int, depth: int = 0, mip_levels: int = 1,
                   color_space: ColorSpace = ColorSpace.Linear) -> bool:
        """Texture::initialize (Texture.cpp:1136-1163)."""
        self.reset()
        dimension = Dimension(dimension)
        width, height, depth = int(width), int(height), int(depth)
        if width <= 0 or height <= 0 or depth < 0 or (dimension == Dimension.Dim3D and depth == 0):
            return False
        self._valid = True
        self._dim, self._w, self._h, self._depth = dimension, width, height, depth
        self._color_space = ColorSpace(color_space)
        self._mips = min(max(int(mip_levels), 1), self.max_mipmap_levels(dimension, width, height, depth))
        self._faces = 6 if dimension == Dimension.Cube else 1
        # (the reference sizes every level of a 3-D texture with the BASE depth here, slots its own
        # setImage then refuses: levels are sized with depth(mip) instead)
        self._images = [[[None]*self._faces for _ in range(self.depth(m))] for m in range(self._mips)]
        return True

    def reset(self):
        self._valid = False
        self._reset_state()

    def is_valid(self) -> bool:
        return self._valid

    def __bool__(self) -> bool:
        return self._valid

    # ---- geometry (Texture.cpp:1170-1232) -----------------------------------------------------
    def dimension(self) -> Dimension:
        return self._dim

    def color_space(self) -> ColorSpace:
        return self._color_space

    def is_array(self) -> bool:
        return self._valid and self._dim != Dimension.Dim3D and self._depth > 0

    def width(self, mip: int = 0) -> int:
        if not self._valid or not (0 <= mip < self._mips):
            return 0
        return max(self._w >> mip, 1)

    def height(self, mip: int = 0) -> int:
        if not self._valid or not (0 <= mip < self._mips):
            return 0
        return max(self._h >> mip, 1)

    def depth(self, mip: int = 0) -> int:
        if not self._valid or not (0 <= mip < self._mips):
            return 0
        if self._dim == Dimension.Dim3D:
            return max(self._depth >> mip, 1)
        return max(self._depth, 1)

    def mip_level_count(self) -> int:
        return self._mips

    def face_count(self) -> int:
        return self._faces

    # ---- images (Texture.cpp:1234-1318) -------------------------------------------------------
    @staticmethod
    def _face_args(args):
        """([face,] mip=0, depth=0) -> (face or None, mip, depth)"""
        if args and isinstance(args[0], CubeFace):
            face, rest = args[0], args[1:]
        else:
            face, rest = None, args
        mip = int(rest[0]) if len(rest) > 0 else 0
        depth = int(rest[1]) if len(rest) > 1 else 0
        return face, mip, depth

    def _slot(self, face: Optional[CubeFace], mip: int, depth: int) -> Optional[int]:
        """face index of a legal ([face,] mip, depth) address, else None"""
        if not self._valid or mip < 0 or depth < 0 or depth >= self.depth(mip):
            return None
        if face is None:
            return 0 if self._faces == 1 else None
        if self._faces != 6 and face != CubeFace.PosX:
            return None
        return int(face)

    def get_image(self, *args) -> Optional[np.ndarray]:
        """Texture::getImage([face,] mip, depth): None is the reference's invalid Image."""
        face, mip, depth = self._face_args(args)
        f = self._slot(face, mip, depth)
        if f is None or mip >= len(self._images) or depth >= len(self._images[mip]):
            return None
        return self._images[mip][depth][f]

    def set_image(self, image, *args, mip: Optional[int] = None, depth: Optional[int] = None) -> bool:
        """Texture::setImage(image, [face,] mipLevel = 0, depth = 0): the image must have the
        level's size; it is kept as the reference's RGBAF conversion would present it."""
        face, m, d = self._face_args(args)
        m = m if mip is None else int(mip)
        d = d if depth is None else int(depth)
        if self._textures:
            return False
        f = self._slot(face, m, d)
        image = _as_image(image)
        if f is None or image is None:
            return False
        if image.shape[1] != self.width(m) or image.shape[0] != self.height(m):
            return False
        self._images[m][d][f] = image
        return True

    def images_complete(self) -> bool:
        if not self._valid:
            return False
        return all(im is not None for level in self._images for dep in level for im in dep)

    # ---- mip generation (Texture.cpp:1320-1514) -----------------------------------------------
    def _context(self) -> api.Context:
        if self._ctx is None:
            self._ctx = api.Context(self._device_id)
        return self._ctx

    def _to_device(self, image: np.ndarray):
        import torch  # device memory: plumbing only
        host = np.ascontiguousarray(image)
        return host, torch.from_numpy(host).to("cuda:%d" % self._device_id)

    def _resize(self, image: np.ndarray, w: int, h: int, filter) -> np.ndarray:
        """Image::resize(w, h, filter) -> RGBAF, on the GPU."""
        import torch
        host, src = self._to_device(image)
        dst = torch.empty((h, w, 4), dtype=torch.float32, device=src.device)
        self._context().resize_device(src.data_ptr(), api.pixel_type_of(host), host.shape[1], host.shape[0],
                                      host.strides[0], dst.data_ptr(), w, h,
                                      color_space=self._color_space, filter=int(filter))
        return dst.cpu().numpy()

    def _chain_2d(self, base: np.ndarray, levels: int, filter) -> List[np.ndarray]:
        """levels 1..levels-1 of one 2-D image, each from the one before"""
        import torch
        if levels <= 1:
            return []
        host, src = self._to_device(base)
        h, w = host.shape[:2]
        dsts = [torch.empty((max(1, h >> k), max(1, w >> k), 4), dtype=torch.float32, device=src.device)
                for k in range(1, levels)]
        self._context().generate_mips_device(src.data_ptr(), api.pixel_type_of(host), w, h, host.strides[0],
                                             [d.data_ptr() for d in dsts], color_space=self._color_space,
                                             filter=int(filter))
        return [d.cpu().numpy() for d in dsts]

    def _level_3d(self, slices: Sequence[np.ndarray], filter) -> List[np.ndarray]:
        """the next level of a 3-D texture from the slices of one level: every slice resized in
        x, y, then generateMips3d along the depth (Texture.cpp:1384-1400, :103-227)"""
        import torch
        arrs = [np.asarray(s) for s in slices]
        if len({a.dtype for a in arrs}) > 1:
            # slices of mixed storage: every image becomes RGBAF first, as Image::convert does
            # (uint8 through v/255.0, Image.cpp:293-296) -- np.stack alone would promote 0..255
            arrs = [(a.astype(np.float64)/255.0).astype(np.float32) if a.dtype == np.uint8 else a.astype(np.float32)
                    for a in arrs]
        vol = np.ascontiguousarray(np.stack(arrs))
        src = torch.from_numpy(vol).to("cuda:%d" % self._device_id)
        d0, h0, w0 = vol.shape[:3]
        w, h, d = max(1, w0 >> 1), max(1, h0 >> 1), max(1, d0 >> 1)
        dst = torch.empty((d, h, w, 4), dtype=torch.float32, device=src.device)
        self._context().generate_mips3d_device(src.data_ptr(), api.pixel_type_of(vol[0]), w0, h0, d0,
                                               vol.strides[1], vol.strides[0], [dst.data_ptr()],
                                               color_space=self._color_space, filter=int(filter))
        out = dst.cpu().numpy()
        return [out[i] for i in range(d)]

    def generate_mipmaps(self, filter=api.ResizeFilter.CatmullRom, mip_levels: Optional[int] = None,
                         custom_mip_images: Optional[Dict[Tuple[int, int, int], CustomMipImage]] = None) -> bool:
        """Texture::generateMipmaps(filter, mipLevels = allMipLevels, customMipImages)
        (Texture.cpp:1320-1514): every level from the previous one through Image::resize in linear
        space (3-D textures: also along the depth), on the GPU.  custom_mip_images maps
        image_index([face,] mip, depth) to a CustomMipImage that replaces the generated level --
        MipReplacement.Once resumes the generated chain below it, Continue builds the lower levels
        from the replacement.  Box / Linear: the reference's in-tree arithmetic; Cubic, CatmullRom
        (the default, as in the reference) and BSpline: FreeImage's resampler restated (FreeImage is
        absent: parity unpinned).  Generated levels are RGBAF (float32) images."""
        if not self._valid or self._textures:
            return False
        if any(im is None for dep in self._images[0] for im in dep):
            return False
        custom = dict(custom_mip_images or {})
        for c in custom.values():
            if c is None or _as_image(c.image) is None:
                return False
        try:
            filter = api.ResizeFilter(filter)
        except ValueError:
            return False
        if mip_levels is None:
            mip_levels = self.allMipLevels
        levels = min(max(int(mip_levels), 1),
                     self.max_mipmap_levels(self._dim, self._w, self._h, max(self._depth, 1)))
        base = self._images[0]
        if self._dim == Dimension.Dim3D:
            # if one slice of a level is replaced, all must be, with one replacement mode (:1362-1378)
            plan = []
            for mip in range(1, levels):
                md = max(self._depth >> mip, 1)
                has = [(0, mip, d) in custom for d in range(md)]
                if any(has) and not all(has):
                    return False
                customs = [custom[(0, mip, d)] for d in range(md)] if all(has) else []
                if any(c.replacement != customs[0].replacement for c in customs):
                    return False
                plan.append(customs)
            self._mips = levels
            images = [base] + [None]*(levels - 1)
            inputs: Optional[List[np.ndarray]] = None     # generated state kept under a `Once` replacement
            for mip in range(1, levels):
                mw, mh = self.width(mip), self.height(mip)
                customs = plan[mip - 1]
                restore = bool(customs) and customs[0].replacement == MipReplacement.Once and mip < levels - 1
                generated = None
                if not customs or restore:
                    source = inputs if inputs is not None else [dep[0] for dep in images[mip - 1]]
                    generated = self._level_3d(source, filter)
                inputs = generated if restore else None
                if customs:
                    level = [self._resize(_as_image(c.image), mw, mh, filter) for c in customs]
                else:
                    level = generated
                images[mip] = [[im] for im in level]
            self._images = images
            return True
        self._mips = levels
        depth = max(self._depth, 1)
        images = [base] + [[[None]*self._faces for _ in range(depth)] for _ in range(levels - 1)]
        for d in range(depth):
            for f in range(self._faces):
                keys = [(f, mip, d) in custom for mip in range(1, levels)]
                if not any(keys):
                    for mip, im in enumerate(self._chain_2d(base[d][f], levels, filter), start=1):
                        images[mip][d][f] = im
                    continue
                prev = None
                for mip in range(1, levels):
                    mw, mh = self.width(mip), self.height(mip)
                    c = custom.get((f, mip, d))
                    restore = c is not None and c.replacement == MipReplacement.Once
                    cur = None
                    if c is None or restore:
                        cur = self._resize(prev if prev is not None else images[mip - 1][d][f], mw, mh, filter)
                    prev = cur if restore else None
                    images[mip][d][f] = self._resize(_as_image(c.image), mw, mh, filter) if c is not None else cur
        self._images = images
        return True

    # ---- conversion (Texture.cpp:1536-1561) ----------------------------------------------------
    def convert(self, format: Format, type: Type, quality: Quality = Quality.Normal,
                alpha_type: Alpha = Alpha.Standard,
                color_mask: Sequence[bool] = (True, True, True, True),
                threads: int = allCores) -> bool:
        del threads  # the GPU path has no thread count (PvrtcConverter-style whole surface)
        if not self.images_complete() or not self.is_format_valid(format, type):
            return False
        format, type = Format(format), Type(type)
        if self._color_space == ColorSpace.sRGB and not self.has_native_srgb(format, type):
            return False
        params = api.make_params(format, type, quality, alpha_type, color_mask, self._color_space)
        flat = [im for level in self._images for dep in level for im in dep]
        # the reference converts every image to RGBAF before it reaches a converter
        # (Converter.h:52-56): half-float images are bit-exact sources for BC6H and the
        # uncompressed packers, every other block kernel takes them as floats
        if not (format == Format.BC6H or int(format) < int(Format.BC1_RGB)):
            flat = [im.astype(np.float32) if im.dtype == np.float16 else im for im in flat]
        try:
            outs = self._context().encode(flat, params)
        except api.CfhipError as e:
            if e.code == api.E_UNSUPPORTED:
                return False  # createConverter -> nullptr -> convert() returns false
            raise
        it = iter(outs)
        self._textures = [[[next(it) for _ in dep] for dep in level] for level in self._images]
        # Converter::convert frees each source image once its surface is done (:586)
        self._images = [[[None]*len(dep) for dep in level] for level in self._images]
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

    def data(self, *args) -> Optional[np.ndarray]:
        """Texture::data([face,] mipLevel = 0, depth = 0): the payload bytes, None where the reference
        returns nullptr."""
        face, mip, depth = self._face_args(args)
        if not self._textures:
            return None
        f = self._slot(face, mip, depth)
        if f is None or mip >= len(self._textures):
            return None
        return self._textures[mip][depth][f]

    def data_size(self, *args) -> int:
        d = self.data(*args)
        return 0 if d is None else int(d.nbytes)

    # ---- saving (Texture.cpp:1636-1685) --------------------------------------------------------
    def _layout(self) -> containers.TextureLayout:
        dim = {Dimension.Dim1D: "1d", Dimension.Dim2D: "2d", Dimension.Dim3D: "3d", Dimension.Cube: "cube"}[self._dim]
        surfaces = [[[f.tobytes() for f in dep] for dep in level] for level in self._textures]
        return containers.TextureLayout(self._format, self._type, self._w, self._h, surfaces, dimension=dim,
                                        depth=self._depth)

    def save_bytes(self, file_type: FileType) -> Tuple[SaveResult, bytes]:
        """Texture::save(std::vector<uint8_t>&, fileType)."""
        if not self.converted():
            return SaveResult.Invalid, b""
        try:
            file_type = FileType(file_type)
        except ValueError:
            return SaveResult.UnknownFormat, b""
        buf = io.BytesIO()
        try:
            if file_type == FileType.DDS:
                containers.write_dds_texture(buf, self._layout(), self._color_space, self._alpha)
            elif file_type == FileType.KTX:
                containers.write_ktx_texture(buf, self._layout(), self._color_space)
            elif file_type == FileType.PVR:
                if not self.is_format_valid(self._format, self._type, FileType.PVR):
                    return SaveResult.Unsupported, b""
                containers.write_pvr_texture(buf, self._layout(), self._color_space, self._alpha)
            else:
                return SaveResult.UnknownFormat, b""
        except ValueError:
            return SaveResult.Unsupported, b""      # no DXGI / GL / PVR form of this (format, type)
        return SaveResult.Success, buf.getvalue()

    def save(self, file_name: Optional[str], file_type: FileType = FileType.Auto) -> SaveResult:
        """Texture::save(fileName, fileType = Auto)."""
        if not self.converted() or not file_name:
            return SaveResult.Invalid
        if FileType(file_type) == FileType.Auto:
            file_type = self.file_type(file_name)
        try:
            stream = open(file_name, "wb")
        except OSError:
            return SaveResult.WriteError
        with stream:
            result, payload = self.save_bytes(file_type)
            if result == SaveResult.Success:
                try:
                    stream.write(payload)
                except OSError:
                    return SaveResult.WriteError
        return result
