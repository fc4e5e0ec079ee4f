"""cuttlefish_amd -- MI355X-native block-texture encoder backend for Cuttlefish.

The product is the HIP library (csrc/ -> libcuttlefish_hip.so) behind the C-ABI of
include/cuttlefish_hip.h.  This package holds the host-side mirror of the
reference interface for that one path (Texture.convert / Converter) used by the
parity tests and bench.py; it never falls back to a CPU encoder.
"""
from .api import (Alpha, CfhipError, ColorSpace, Context, Format, PixelType, Quality,  # noqa: F401
                  ResizeFilter, Type,
                  device_count, load_library, make_params, payload_size, query, shard_rows)
from . import shard  # noqa: F401
from .texture import (CubeFace, CustomMipImage, Dimension, FileType, ImageFormat, MipReplacement, SaveResult,  # noqa: F401
                      Texture, image_index)

__all__ = ["Alpha", "CfhipError", "ColorSpace", "Context", "Format", "PixelType", "Quality",
           "ResizeFilter", "Type", "Texture", "Dimension", "CubeFace", "FileType", "SaveResult",
           "MipReplacement", "CustomMipImage", "ImageFormat", "image_index", "device_count", "load_library", "make_params", "payload_size",
           "query", "shard", "shard_rows"]
