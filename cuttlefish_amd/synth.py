"""Deterministic synthetic surfaces for the block-encode path.

There is no dataset and no network: every test/bench input is generated here
from integer seeds, identically on every machine.

* ``gradient``   -- the reference's own test pattern, lib/test/TextureTest.cpp:53-61
                    (r = x/(w-1), g = y/(h-1), b = (w-1-x)/(w-1), a = (h-1-y)/(h-1)).
* ``photo``      -- "photographic" RGBA8 tile: 6 octaves of bilinear value noise on a
                    shared luminance field + weaker independent chroma fields, hard
                    edged rectangles/discs (these exercise the partitioned modes),
                    +-2 LSB grain, alpha = 255 except a 12.5 % band with a smooth
                    alpha ramp (exercises BC7 modes 4-7).
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0xC0FFEE


def gradient(width: int, height: int, dtype=np.uint8) -> np.ndarray:
    """Reference test colour (TextureTest.cpp:53-61) as RGBA8 or RGBA32F."""
    x = np.arange(width, dtype=np.float64)[None, :]
    y = np.arange(height, dtype=np.float64)[:, None]
    wd = max(width - 1, 1)
    hd = max(height - 1, 1)
    img = np.empty((height, width, 4), np.float64)
    img[..., 0] = x / wd
    img[..., 1] = y / hd
    img[..., 2] = (width - x - 1) / wd
    img[..., 3] = (height - y - 1) / hd
    if dtype == np.uint8:
        # same quantisation as toColorBlock (S3tcConverter.cpp:97-111) from float32
        f = img.astype(np.float32)
        return np.floor(np.clip(f, 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)
    return img.astype(np.float32)


def _value_noise(rng: np.random.Generator, width: int, height: int, cell: int) -> np.ndarray:
    gx = width // cell + 2
    gy = height // cell + 2
    lat = rng.random((gy, gx))
    xs = np.arange(width) / cell
    ys = np.arange(height) / cell
    x0 = xs.astype(np.int64)
    y0 = ys.astype(np.int64)
    fx = (xs - x0)[None, :]
    fy = (ys - y0)[:, None]
    a = lat[y0][:, x0]
    b = lat[y0][:, x0 + 1]
    c = lat[y0 + 1][:, x0]
    d = lat[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def _fbm(rng, width, height, base_cell=256, octaves=6, persistence=0.55):
    out = np.zeros((height, width))
    amp = 1.0
    norm = 0.0
    cell = base_cell
    for _ in range(octaves):
        out += amp * _value_noise(rng, width, height, max(cell, 1))
        norm += amp
        amp *= persistence
        cell = max(cell // 2, 1)
    return out / norm


def photo(width: int, height: int, seed: int = 1, alpha: bool = True) -> np.ndarray:
    """Photo-like RGBA8 tile, deterministic in (width, height, seed)."""
    rng = np.random.default_rng(SEED_BASE + seed)
    lum = _fbm(rng, width, height)
    img = np.empty((height, width, 4), np.float64)
    for c in range(3):
        chroma = _fbm(rng, width, height, base_cell=128, octaves=4)
        img[..., c] = lum * 0.75 + 0.35 * (chroma - 0.5) + 0.1
    # hard-edged shapes: 64 per Mpixel
    nshapes = max(4, (width * height * 64) // (1 << 20))
    yy, xx = np.mgrid[0:height, 0:width]
    for i in range(nshapes):
        cx = int(rng.integers(0, width))
        cy = int(rng.integers(0, height))
        rx = int(rng.integers(3, max(4, width // 16)))
        ry = int(rng.integers(3, max(4, height // 16)))
        col = rng.random(3)
        x0, x1 = max(cx - rx, 0), min(cx + rx, width)
        y0, y1 = max(cy - ry, 0), min(cy + ry, height)
        if i & 1:
            img[y0:y1, x0:x1, :3] = col
        else:
            sub = ((xx[y0:y1, x0:x1] - cx) / rx) ** 2 + ((yy[y0:y1, x0:x1] - cy) / ry) ** 2 <= 1.0
            img[y0:y1, x0:x1, :3][sub] = col
    img[..., 3] = 1.0
    if alpha:
        # 12.5 % band with a smooth alpha ramp
        y0 = height // 2
        y1 = y0 + max(height // 8, 1)
        ramp = np.linspace(0.0, 1.0, width)[None, :] * np.ones((y1 - y0, 1))
        img[y0:y1, :, 3] = ramp * (0.5 + 0.5 * lum[y0:y1])
    out = np.clip(img, 0, 1) * 255.0
    grain = rng.integers(-2, 3, size=(height, width, 3))
    out[..., :3] += grain
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def hdr_probe(width: int, height: int, seed: int = 4, signed: bool = False) -> np.ndarray:
    """HDR RGBA16F probe: exp2(12*noise - 6) (2^-6..2^6) with weak
    chroma variation plus four gaussian "suns" peaking near 6e4; alpha = 1.  signed=True
    flips the sign of one quadrant-sized region (BC6H SF16 inputs)."""
    rng = np.random.default_rng(SEED_BASE + 1000 + seed)
    lum = _fbm(rng, width, height)
    img = np.empty((height, width, 4), np.float64)
    for c in range(3):
        chroma = _fbm(rng, width, height, base_cell=128, octaves=4)
        img[..., c] = np.exp2(12.0 * lum - 6.0) * (0.6 + 0.8 * chroma)
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(4):
        cx, cy = rng.integers(0, width), rng.integers(0, height)
        sig = max(width, height) / float(rng.integers(24, 64))
        g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sig * sig))
        col = 0.5 + 0.5 * rng.random(3)
        img[..., :3] += 6.0e4 * g[..., None] * col
    img[..., :3] = np.minimum(img[..., :3], 65000.0)
    if signed:
        img[: height // 2, : width // 2, :3] *= -1.0
    img[..., 3] = 1.0
    return img.astype(np.float16)


def psnr_log(ref_half: np.ndarray, dec_half: np.ndarray) -> float:
    """PSNR of log2(1+|x|)*sign(x) mapped RGB (SURVEY.md 8d HDR metric), peak = log2(1+65504)."""
    def m(x):
        x = x[..., :3].astype(np.float64)
        return np.sign(x) * np.log2(1.0 + np.abs(x))
    d = m(ref_half) - m(dec_half)
    mse = float(np.mean(d * d))
    if mse == 0.0:
        return 99.0
    peak = np.log2(1.0 + 65504.0)
    return float(10.0 * np.log10(peak * peak / mse))


def psnr(a: np.ndarray, b: np.ndarray, channels=slice(0, 4)) -> float:
    d = a[..., channels].astype(np.float64) - b[..., channels].astype(np.float64)
    mse = float(np.mean(d * d))
    if mse == 0.0:
        return 99.0
    return float(10.0 * np.log10(255.0 * 255.0 / mse))


def psnr_y(a: np.ndarray, b: np.ndarray) -> float:
    """PSNR of Rec.709 luma."""
    wts = np.array([0.2126, 0.7152, 0.0722])
    ya = a[..., :3].astype(np.float64) @ wts
    yb = b[..., :3].astype(np.float64) @ wts
    mse = float(np.mean((ya - yb) ** 2))
    if mse == 0.0:
        return 99.0
    return float(10.0 * np.log10(255.0 * 255.0 / mse))


def photo2(width: int, height: int, seed: int = 1) -> np.ndarray:
    """A second opaque RGBA8 tile built the way a camera picture is: DETAILED luma (fine fractal texture,
    many small soft-edged objects, luma-only sensor grain) over SMOOTHER chroma (fields with nothing finer
    than a few texels, as chroma subsampling leaves them), converted Y Cb Cr -> R G B.  ``photo`` adds
    independent +-2 LSB grain to each channel, which makes the three channels of a block uncorrelated at
    the LSB level -- its BC7 blocks go to mode 5 (78 %) and never to modes 1 / 6; blocks of real
    photographs (tests/golden/real_blocks.npz) split 37 / 22 / 18 / 17 % over modes 1 / 3 / 6 / 5, and
    this tile is tuned to that split (tests/test_synth.py holds the histogram)."""
    rng = np.random.default_rng(SEED_BASE + 7000 + seed)
    # luma: large-scale light + fine texture whose strength itself varies over the picture
    base = _fbm(rng, width, height, base_cell=192, octaves=5, persistence=0.5)
    fine = _fbm(rng, width, height, base_cell=8, octaves=4, persistence=0.9) - 0.5
    strength = np.clip(_fbm(rng, width, height, base_cell=96, octaves=3) * 2.6 - 1.0, 0.0, 1.0) ** 2
    y = 0.08 + 0.84 * base + 1.3 * strength * fine
    cb = 0.30 * (_fbm(rng, width, height, base_cell=160, octaves=3) - 0.5) + \
        0.9 * strength * (_fbm(rng, width, height, base_cell=8, octaves=2) - 0.5)
    cr = 0.30 * (_fbm(rng, width, height, base_cell=160, octaves=3) - 0.5) + \
        0.9 * strength * (_fbm(rng, width, height, base_cell=8, octaves=2) - 0.5)
    # soft-edged objects: their own luma offset and chroma, one-texel transition
    nshapes = max(8, (width * height * 500) // (1 << 20))
    yy, xx = np.mgrid[0:height, 0:width]
    for i in range(nshapes):
        cx, cy = int(rng.integers(0, width)), int(rng.integers(0, height))
        rx = int(rng.integers(3, 48))
        ry = int(rng.integers(3, 48))
        x0, x1 = max(cx - rx - 2, 0), min(cx + rx + 2, width)
        y0, y1 = max(cy - ry - 2, 0), min(cy + ry + 2, height)
        if i & 1:
            d = np.maximum(np.abs(xx[y0:y1, x0:x1] - cx) - rx, np.abs(yy[y0:y1, x0:x1] - cy) - ry)
        else:
            r = np.sqrt(((xx[y0:y1, x0:x1] - cx) / rx) ** 2 + ((yy[y0:y1, x0:x1] - cy) / ry) ** 2)
            d = (r - 1.0) * min(rx, ry)
        cov = np.clip(0.5 - d, 0.0, 1.0)              # coverage: 1 inside, 0 outside, a ramp one texel wide
        dy, dcb, dcr = rng.random() * 0.5 - 0.25, rng.random() * 0.4 - 0.2, rng.random() * 0.4 - 0.2
        y[y0:y1, x0:x1] += cov * dy
        cb[y0:y1, x0:x1] = cb[y0:y1, x0:x1] * (1 - cov) + cov * dcb
        cr[y0:y1, x0:x1] = cr[y0:y1, x0:x1] * (1 - cov) + cov * dcr
    y += rng.normal(0.0, 1.5 / 255.0, size=(height, width))     # luma grain
    # chroma noise a few texels wide (what a decoded JPEG carries): decorrelates the channels of flat blocks
    cb += (5.0 / 255.0) * (_value_noise(rng, width, height, 3) - 0.5)
    cr += (5.0 / 255.0) * (_value_noise(rng, width, height, 3) - 0.5)
    img = np.empty((height, width, 4), np.float64)
    img[..., 0] = y + 1.402 * cr
    img[..., 1] = y - 0.344136 * cb - 0.714136 * cr
    img[..., 2] = y + 1.772 * cb
    img[..., 3] = 1.0
    return np.clip(np.floor(np.clip(img, 0, 1) * 255.0 + 0.5), 0, 255).astype(np.uint8)
