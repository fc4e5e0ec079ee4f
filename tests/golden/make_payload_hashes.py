"""Generates tests/golden/payload_hashes.json: SHA-256 of the ORACLE's payload for every block
format x type x quality on fixed synthetic images, with the PSNR of its decode.  The upstream
encoders are absent, so nothing else would notice the oracle's search drifting between rounds:
with this fixture every change of an encoder's output is an explicit, reviewed diff of this file
(hash AND the PSNR before / after).  tests/test_payload_hashes.py checks the oracle against it
(CPU) and the HIP kernels against it (GPU).

    python tests/golden/make_payload_hashes.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from cuttlefish_amd import synth  # noqa: E402

W, H = 64, 48
# (name, format, type, image kind)
CASES = [("BC1_RGB", 29, 0, "opaque"), ("BC1_RGBA", 30, 0, "cutout"), ("BC2", 31, 0, "alpha"),
         ("BC3", 32, 0, "alpha"), ("BC4_UNorm", 33, 0, "alpha"), ("BC4_SNorm", 33, 1, "float"),
         ("BC5_UNorm", 34, 0, "alpha"), ("BC5_SNorm", 34, 1, "float"), ("BC6H_UFloat", 35, 4, "hdr"),
         ("BC6H_Float", 35, 5, "hdr"), ("BC7", 36, 0, "alpha"), ("BC7_opaque", 36, 0, "opaque"),
         ("BC7_sRGB", 36, 0, "alpha"), ("ETC2_R8G8B8_sRGB", 38, 0, "opaque"),
         ("ASTC_4x4_UFloat", 43, 4, "hdr32"), ("ASTC_6x6_UFloat", 47, 4, "hdr32"), ("ASTC_8x8_UFloat", 50, 4, "hdr32"),
         ("ETC1", 37, 0, "opaque"), ("ETC2_R8G8B8", 38, 0, "opaque"), ("ETC2_R8G8B8A1", 39, 0, "cutout"),
         ("ETC2_R8G8B8A8", 40, 0, "alpha"), ("EAC_R11_UNorm", 41, 0, "alpha"), ("EAC_R11_SNorm", 41, 1, "float"),
         ("EAC_R11G11_UNorm", 42, 0, "alpha"), ("EAC_R11G11_SNorm", 42, 1, "float")] + \
        [("ASTC_%dx%d" % fp, 43 + i, 0, "alpha_big") for i, fp in enumerate(
            [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6), (10, 8),
             (10, 10), (12, 10), (12, 12)])]


def color_space(name):
    """cases named *_sRGB are encoded as sRGB images (perceptual metrics at >= Normal)"""
    return 1 if name.endswith("_sRGB") else 0


def image(kind):
    if kind == "alpha_big":
        # ASTC's levels rank differently sized seed lists and spread their lanes unevenly: a few hundred
        # blocks are too few for the PSNR ladder to mean anything (round 3), so these cases get 9x the area
        return np.ascontiguousarray(synth.photo(3*W, 3*H, seed=77))
    img = synth.photo(W, H, seed=77)
    if kind == "opaque":
        img = img.copy(); img[..., 3] = 255
    elif kind == "cutout":
        img = img.copy(); img[..., 3] = np.where(img[..., 0] > 128, 255, 0)
    elif kind == "float":
        f = img.astype(np.float32)/127.5 - 1.0
        return np.ascontiguousarray(f)
    elif kind == "hdr":
        return synth.hdr_probe(W, H, seed=78)
    elif kind == "hdr32":
        return np.ascontiguousarray(synth.hdr_probe(W, H, seed=78).astype(np.float32))
    return np.ascontiguousarray(img)


def quality_metric(name, fmt, typ, img, payload):
    if fmt == 35:
        dec = O.decode_bc6h(payload, W, H, typ).astype(np.float32)
        ref = img[..., :3].astype(np.float32)
        if typ == 4:
            ref = np.maximum(ref, 0)
        e = np.log2(1 + np.abs(dec)) - np.log2(1 + np.abs(ref))
        return round(float(10*np.log10(16.0**2/max(np.mean(e**2), 1e-12))), 3)
    if 37 <= fmt <= 40:
        dec = O.decode_etc(payload, fmt, W, H)
    elif fmt in (41, 42):
        return None
    elif fmt >= 43 and typ == 4:
        dec = O.decode_astc_hdr(payload, fmt, W, H)[0].astype(np.float32)
        e = np.log2(1 + np.abs(dec[..., :3])) - np.log2(1 + np.abs(img[..., :3].astype(np.float32)))
        return round(float(10*np.log10(16.0**2/max(np.mean(e**2), 1e-12))), 3)
    elif fmt >= 43:
        dec, _ = O.decode_astc(payload, fmt, img.shape[1], img.shape[0])
    else:
        dec = O.decode(payload, fmt, W, H, typ)
    if img.dtype != np.uint8:
        return None
    ch = slice(0, 3) if fmt in (29, 37, 38) else slice(0, 4)
    if fmt == 33:
        ch = slice(0, 1)
    if fmt == 34:
        ch = slice(0, 2)
    return round(float(synth.psnr(img, dec, ch)), 3)


_LNS = None


def lns_psnr(fmt, img, payload):
    """ASTC HDR profiles: PSNR on the 16-bit LNS values of the halves -- the domain the encoder minimises its error
    in (cfo_astc_lns16; peak 65535).  The log2(1 + x) figure above weighs dark texels differently and is not
    monotone in the level on this fixture; this one is what the ladder is held to."""
    global _LNS
    import ctypes
    if _LNS is None:
        L = O.lib()
        L.cfo_astc_lns16.argtypes = [ctypes.c_uint16]
        L.cfo_astc_lns16.restype = ctypes.c_int
        _LNS = np.array([L.cfo_astc_lns16(h) for h in range(65536)], np.int64)

    def lns(x):
        return _LNS[np.asarray(x, np.float32).clip(0, 65504).astype(np.float16).view(np.uint16)]
    dec = O.decode_astc_hdr(payload, fmt, W, H)[0].astype(np.float32)
    e = (lns(dec[..., :3]) - lns(img[..., :3])).astype(np.float64)
    return round(float(10*np.log10(65535.0**2/max(np.mean(e**2), 1e-12))), 3)


def build():
    out = {}
    for name, fmt, typ, kind in CASES:
        img = image(kind)
        for q in range(5):
            payload = O.encode(img, fmt, typ=typ, quality=q, threads=8, color_space=color_space(name))
            out["%s/q%d" % (name, q)] = {"sha256": hashlib.sha256(payload.tobytes()).hexdigest(),
                                          "psnr": quality_metric(name, fmt, typ, img, payload)}
            if fmt >= 43 and typ == 4:
                out["%s/q%d" % (name, q)]["psnr_lns"] = lns_psnr(fmt, img, payload)
    return out


if __name__ == "__main__":
    res = build()
    path = os.path.join(HERE, "payload_hashes.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    changed = [k for k in res if k in old and old[k]["sha256"] != res[k]["sha256"]]
    for k in changed:
        print("CHANGED %-24s psnr %s -> %s" % (k, old[k]["psnr"], res[k]["psnr"]))
    json.dump(res, open(path, "w"), indent=0, sort_keys=True)
    print("wrote %d entries (%d changed, %d new)" % (len(res), len(changed), len([k for k in res if k not in old])))
