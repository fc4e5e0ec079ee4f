"""Loader of tests/golden/real_blocks.npz -- blocks sampled from real photographs by
tests/golden/make_real_blocks.py (the photographs stay in the build container; the blocks are data).

Every quality claim of the codecs is measured on these blocks (tools/quality_tables.py, tools/bc7_lab.py,
tests/test_oracle_bounds.py); TEST INFRASTRUCTURE, nothing under cuttlefish_amd/ reads it.
"""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_blocks.npz")
_cache = None


def _load():
    global _cache
    if _cache is None:
        d = np.load(PATH)
        _cache = {k: d[k] for k in d.files}
    return _cache


def image_names(group="a"):
    """the photographs of a group: "a" = the eight the ladders were balanced on (round 5); "b" = the five held-out
    ones of round 6 (no tuning tool reads them; make_real_blocks.py)"""
    return [str(s) for s in _load()["images" + _sfx(group)]]


def _sfx(group):
    assert group in ("a", "b")
    return "_b" if group == "b" else ""


def blocks4(count=None, alpha=False, image=None, group="a"):
    """-> (n, 4, 4, 4) uint8 RGBA blocks.  alpha=False: opaque photograph blocks (interleaved over the
    images so any prefix is a fair sample); alpha=True: the alpha-carrying set (group "a" only)."""
    d = _load()
    if alpha:
        assert group == "a"
        b = d["rgba4"]
    else:
        rgb, src = d["rgb4" + _sfx(group)], d["rgb4" + _sfx(group) + "_img"]
        if image is not None:
            rgb = rgb[src == image_names(group).index(image)]
        else:
            nimg = len(d["images" + _sfx(group)])
            per = len(rgb) // nimg
            rgb = rgb.reshape(nimg, per, 4, 4, 3).transpose(1, 0, 2, 3, 4).reshape(-1, 4, 4, 3)
        b = np.concatenate([rgb, np.full(rgb.shape[:3] + (1,), 255, np.uint8)], axis=-1)
    if count is not None:
        b = b[:count]
    return np.ascontiguousarray(b)


def blocks_alpha(bw, bh, count=None):
    """-> (n, bh, bw, 4) uint8 blocks WITH a varying alpha channel of an ASTC footprint up to 12x12 (a photograph's
    colour, another photograph's luma as alpha; top-left crop of the rgba12 patches)."""
    b = _load()["rgba12"][:, :bh, :bw]
    if count is not None:
        b = b[:count]
    return np.ascontiguousarray(b)


def blocks(bw, bh, count=None, image=None, group="a"):
    """-> (n, bh, bw, 4) uint8 opaque blocks of an ASTC footprint up to 12x12 (top-left crop of the patches)."""
    d = _load()
    rgb, src = d["rgb12" + _sfx(group)], d["rgb12" + _sfx(group) + "_img"]
    if image is not None:
        rgb = rgb[src == image_names(group).index(image)]
    else:
        nimg = len(d["images" + _sfx(group)])
        per = len(rgb) // nimg
        rgb = rgb.reshape(nimg, per, 12, 12, 3).transpose(1, 0, 2, 3, 4).reshape(-1, 12, 12, 3)
    rgb = rgb[:, :bh, :bw]
    b = np.concatenate([rgb, np.full(rgb.shape[:3] + (1,), 255, np.uint8)], axis=-1)
    if count is not None:
        b = b[:count]
    return np.ascontiguousarray(b)


def strip(blks):
    """(n, bh, bw, 4) blocks -> the (bh, n*bw, 4) image whose block row they are."""
    return np.ascontiguousarray(np.concatenate(list(blks), axis=1))


def two_channel_gradients(bw, bh, nbx=6):
    """one block row of bw x bh blocks, each a bilinear gradient whose red / blue run along x and whose green runs along
    y (two independent weight fields: what a second weight plane on a coarse grid with many levels is for) -- the
    content class the held-out colour graphic of group b consists of (round 6)"""
    img = np.zeros((bh, bw*nbx, 4), np.uint8)
    img[..., 3] = 255
    yy, xx = np.mgrid[0:bh, 0:bw*nbx]
    for k in range(nbx):
        sl = slice(k*bw, (k + 1)*bw)
        x = (xx[:, sl] - k*bw)/(bw - 1)
        y = yy[:, sl]/(bh - 1)
        img[:, sl, 0] = np.round(40 + 20*k + (30 + 5*k)*x)
        img[:, sl, 1] = np.round(200 - 15*k - (25 + 3*k)*y)
        img[:, sl, 2] = np.round(60 + (10 + 2*k)*x)
    return img
