"""Drift guard: the oracle's payload for every block format x type x quality on fixed synthetic
images against the committed SHA-256 list (tests/golden/payload_hashes.json, generator
make_payload_hashes.py).  The reference's encoders are absent, so this is what makes a change of
an encoder's OUTPUT an explicit diff (hash + PSNR before / after) instead of a silent co-evolution
of oracle and kernel.  The GPU half checks the HIP kernels against the same committed hashes."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
import make_payload_hashes as G  # noqa: E402

HASHES = json.load(open(os.path.join(GOLDEN, "payload_hashes.json")))


@pytest.mark.parametrize("case", G.CASES, ids=lambda c: c[0])
def test_oracle_payload_hashes_unchanged(case):
    name, fmt, typ, kind = case
    img = G.image(kind)
    for q in range(5):
        payload = O.encode(img, fmt, typ=typ, quality=q, threads=4, color_space=G.color_space(name))
        want = HASHES["%s/q%d" % (name, q)]
        assert hashlib.sha256(payload.tobytes()).hexdigest() == want["sha256"], \
            "%s quality %d: the oracle's output changed -- regenerate the fixture and review " \
            "the PSNR diff (was %s)" % (name, q, want["psnr"])


def test_quality_never_costs_quality():
    """PSNR is non-decreasing in Texture::Quality for every format whose levels search nested
    candidate sets; ASTC's levels up to High shortlist partitions from differently sized lists (not
    nested) and get 0.1 dB of slack; High -> Highest is nested (same candidates in the same order,
    every config list a prefix of the deeper one) and must never lose."""
    for name, fmt, typ, kind in G.CASES:
        ps = [HASHES["%s/q%d" % (name, q)]["psnr"] for q in range(5)]
        if ps[0] is None or G.color_space(name):
            continue      # (sRGB cases minimise a perceptual error, not the recorded PSNR)
        # (the ASTC cases use a 192 x 144 image: on 64 x 48 -- a few dozen blocks of the larger
        # footprints -- one block decided the ladder)
        slack = 0.1 if name.startswith("ASTC") else 0.01
        if name.startswith("ASTC") and name.endswith("UFloat"):
            # The HDR profiles minimise their error on 16-bit LNS values; the log2(1 + x) PSNR recorded beside the
            # hash weighs dark texels differently and is not monotone in the level on this fixture (round-4 review:
            # 1 dB of slack).  The ladder is held to the domain it optimises: psnr_lns, with the slack of the LDR
            # cases (4x4 54.10 / 54.46 / 54.90 / 55.50 / 55.71, 6x6 45.61 .. 45.77, 8x8 41.64 .. 41.88).
            ps = [HASHES["%s/q%d" % (name, q)]["psnr_lns"] for q in range(5)]
        for a, b in zip(ps, ps[1:]):
            assert b >= a - slack, (name, ps)
        assert ps[4] >= ps[3] - 0.01, (name, ps)
        assert ps[4] >= ps[0], (name, ps)


@pytest.mark.gpu
@pytest.mark.parametrize("case", G.CASES, ids=lambda c: c[0])
def test_gpu_payload_hashes(gpu_ctx, case):
    from cuttlefish_amd import ColorSpace, Format, Type, make_params
    name, fmt, typ, kind = case
    img = G.image(kind)
    for q in range(5):
        got = gpu_ctx.encode([img], make_params(Format(fmt), Type(typ), q, color_space=ColorSpace(G.color_space(name))))[0]
        assert hashlib.sha256(np.asarray(got).tobytes()).hexdigest() == HASHES["%s/q%d" % (name, q)]["sha256"], \
            "%s quality %d" % (name, q)
