#!/usr/bin/env python3
"""Freeze what two INDEPENDENT ENCODERS achieve on the fixture images (run here, where both exist):

  * Pillow 12.2's DDS writer (DXT1 / DXT3 / DXT5 / BC5), authors: the Pillow project;
  * Mesa 23.2.1's software texture compression behind glTexImage2D(GL_COMPRESSED_*), reached
    through tools/mesa_ref (S3TC: the former libtxc_dxtn, RGTC, BPTC incl. BC6H), authors: the
    Mesa project.

Neither shares code or authors with oracle/ or the kernels.  The reference's own encoders (rgbcx,
squish, bc7enc, Compressonator, ispc_texcomp) are absent submodules; these two are the encoders
that DO exist in this image, and they answer the question the oracle alone cannot: is our search
at least as good as somebody else's?  Output: tests/golden/independent_encoders.json -- per
(format, image) the PSNR each independent encoder reaches, measured on the decode of OUR decoder
(which is pinned to Pillow's and Mesa's decoders bit for bit), plus SHA-256 of their payloads.
tests/test_independent_encoders.py compares the oracle (CPU) and the kernels (GPU) with it.

    python tests/golden/make_independent_encoders.py
"""
import hashlib
import io
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mesa_lib as M          # noqa: E402
import oracle_lib as O        # noqa: E402
from cuttlefish_amd import synth      # noqa: E402

BC1, BC1A, BC2, BC3, BC4, BC5, BC6H, BC7 = 29, 30, 31, 32, 33, 34, 35, 36


def images():
    """name -> (h, w, 4) uint8: deterministic generators only (the JSON stores no pixels)."""
    out = {}
    for seed in (3, 11):
        p = synth.photo(256, 256, seed=seed)
        out["photo%d" % seed] = p
        o = p.copy()
        o[..., 3] = 255
        out["photo%d_opaque" % seed] = o
    out["gradient"] = synth.gradient(128, 128)            # the reference's own test pattern (TextureTest.cpp:53-61)
    return out


def metric(fmt, ref, dec):
    """PSNR over the channels the format stores."""
    if fmt == BC4:
        return synth.psnr(ref, dec, slice(0, 1))
    if fmt == BC5:
        return synth.psnr(ref, dec, slice(0, 2))
    if fmt == BC1:
        return synth.psnr(ref, dec, slice(0, 3))
    return synth.psnr(ref, dec)


def pillow_encode(img, pixel_format):
    from PIL import Image
    buf = io.BytesIO()
    if pixel_format == "BC5":
        Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB").save(buf, format="DDS", pixel_format="BC5")
    else:
        Image.fromarray(img, "RGBA").save(buf, format="DDS", pixel_format=pixel_format)
    data = buf.getvalue()
    hdr = 128 + (20 if data[84:88] == b"DX10" else 0)
    return np.frombuffer(data[hdr:], np.uint8).copy()


def main():
    assert M.available(), "needs Mesa's swrast_dri.so (this container)"
    import PIL
    rows = []
    pil = {BC1: "DXT1", BC2: "DXT3", BC3: "DXT5", BC5: "BC5"}
    for name, img in images().items():
        h, w = img.shape[:2]
        for fmt in (BC1, BC2, BC3, BC4, BC5, BC7):
            if fmt == BC1 and not name.endswith("opaque") and name != "gradient":
                continue                                   # BC1_RGB: opaque content only
            src = img
            if fmt == BC1:
                src = img.copy()
                src[..., 3] = 255
            row = {"format": fmt, "image": name}
            if fmt in pil:
                pay = pillow_encode(src, pil[fmt])
                dec = O.decode(pay, fmt, w, h)
                row["pillow_psnr"] = round(metric(fmt, src, dec), 4)
                row["pillow_sha256"] = hashlib.sha256(pay.tobytes()).hexdigest()[:16]
            pay = M.encode(fmt, src)
            dec = O.decode(pay, fmt, w, h)
            row["mesa_psnr"] = round(metric(fmt, src, dec), 4)
            row["mesa_sha256"] = hashlib.sha256(pay.tobytes()).hexdigest()[:16]
            rows.append(row)
    # BC6H: Mesa compresses float RGB; PSNR in the log domain on the half decode (synth.psnr_log)
    hdr = synth.hdr_probe(128, 128, seed=4)
    pay = M.encode(BC6H, np.ascontiguousarray(hdr.astype(np.float32)), typ=4)
    dec = O.decode_bc6h(pay, 128, 128, 4)
    rows.append({"format": BC6H, "image": "hdr_probe", "mesa_psnr": round(synth.psnr_log(hdr[..., :3], dec), 4),
                 "mesa_sha256": hashlib.sha256(pay.tobytes()).hexdigest()[:16]})
    out = {"pillow": PIL.__version__, "mesa": M.version(),
           "metric": "PSNR (dB, peak 255) over the stored channels of the image decoded by oracle/bcn_decode.c; "
                     "BC6H: synth.psnr_log on halves",
           "rows": rows}
    with open(os.path.join(HERE, "independent_encoders.json"), "w") as f:
        json.dump(out, f, indent=1)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
