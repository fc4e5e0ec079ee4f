"""BASELINE config 5 ("256 x 2048x2048 RGBA8 mipchains, BC7") against the CPU oracle.

The workload is Converter::convert's loop over [mip][depth][face] surfaces (lib/src/Converter.cpp:
521-589) for a texture array whose chains come from Texture::generateMipmaps (lib/src/Texture.cpp:
1320-1514): here the chain is generated on the GPU (cfhip_generate_mips_array_device) and every
surface of every texture goes through ONE batched cfhip_encode_device per source pixel type.

(a) one 2048 x 2048 texture: its 12-level chain and the payload of every level byte-equal to
    O.mip_chain -> O.encode -- the small levels (<= 256 x 256) whole, the large ones whole too when the
    host has the cores for it, else on strips of block rows (blocks are independent);
(b) the full 256-texture array: payload size 1 431 662 592 B, determinism, the LPT plan of
    shard.assign_surfaces over 8 ranks covers all 3 072 surfaces exactly once and the 8 per-rank
    batched encodes reproduce the single-call payload, three textures re-derived through (a).
"""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from cuttlefish_amd import ColorSpace, Context, Format, PixelType, Type, make_params, shard

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (gpu_texture, c5_geometry, c5_surfaces: the bench's own workload builders)

N = 2048
QUALITY = 2
BC7 = int(Format.BC7)


@pytest.fixture(scope="module")
def ctx():
    with Context(0) as c:
        yield c


def _gpu_chains(ctx, bases, geo):
    """The chains of `bases` (list of (N, N, 4) uint8 device tensors) and their batched payload."""
    dev = bases[0].device
    chains = [[torch.empty((d, d, 4), dtype=torch.float32, device=dev) for d in geo["dims"][1:]] for _ in bases]
    out = torch.zeros(len(bases) * geo["chain_bytes"], dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()                      # torch's producers finish before the library's stream reads
    ctx.generate_mips_array_device([b.data_ptr() for b in bases], PixelType.RGBA8, N, N, N * 4,
                                   [[c.data_ptr() for c in ch] for ch in chains],
                                   color_space=ColorSpace.Linear, filter=0)
    ctx.encode_device(bench.c5_surfaces(geo, bases, chains, out), make_params(Format.BC7, Type.UNorm, QUALITY))
    torch.cuda.synchronize()
    return chains, out


def _check_texture_against_oracle(base_np, chain_gpu, payload, geo, whole_levels_up_to):
    """chain_gpu: list of (d, d, 4) float32 arrays (levels 1..); payload: the texture's chain bytes."""
    ref_chain = O.mip_chain(base_np, geo["levels"], filter=0, color_space=0)
    threads = bench.usable_cpus()
    off = 0
    for k, d in enumerate(geo["dims"]):
        if k:
            assert np.array_equal(chain_gpu[k - 1], ref_chain[k]), "mip level %d differs from the oracle" % k
        src = base_np if k == 0 else ref_chain[k]
        got = payload[off:off + geo["nbytes"][k]]
        if d <= whole_levels_up_to:
            ref = O.encode(src, BC7, quality=QUALITY, threads=threads)
            assert np.array_equal(ref, got), "payload of level %d (%d x %d) differs from the oracle" % (k, d, d)
        else:
            # strips of 4 block rows: top, through the alpha band in the middle, bottom
            bx = d // 4
            rows = got.reshape(bx, bx * 16)
            for r0 in (0, bx // 2, bx // 2 + bx // 16, bx - 4):
                ref = O.encode(src[r0 * 4:(r0 + 4) * 4], BC7, quality=QUALITY, threads=threads)
                assert np.array_equal(ref, rows[r0:r0 + 4].reshape(-1)), \
                    "payload of level %d, block rows %d..%d differs from the oracle" % (k, r0, r0 + 4)
        off += geo["nbytes"][k]
    assert off == geo["chain_bytes"] == payload.size


def test_c5_one_texture_chain_and_payload_equal_the_oracle(ctx):
    geo = bench.c5_geometry(N)
    assert geo["levels"] == 12 and geo["px_chain"] == 5592405 and geo["blocks_chain"] == 349527
    dev = torch.device("cuda", 0)
    base = bench.gpu_texture(torch, N, 7, dev)
    chains, out = _gpu_chains(ctx, [base], geo)
    # enough host cores: every level whole (5.6 Mpixel through the oracle); otherwise whole up to 256 x 256
    whole = N if bench.usable_cpus() >= 8 else 256
    _check_texture_against_oracle(base.cpu().numpy(), [c.cpu().numpy() for c in chains[0]],
                                  out.cpu().numpy(), geo, whole)


@pytest.mark.timeout(1500)
def test_c5_full_array_256_textures(ctx):
    T, WORLD = 256, 8
    geo = bench.c5_geometry(N)
    dev = torch.device("cuda", 0)
    bases = [bench.gpu_texture(torch, N, t, dev) for t in range(T)]
    chains, out = _gpu_chains(ctx, bases, geo)
    assert out.numel() == 1431662592 == T * geo["chain_bytes"]
    assert T * geo["px_chain"] == 1431655680 and T * geo["blocks_chain"] == 89478912
    assert T * (geo["px_chain"] * 4 + geo["chain_bytes"]) == bench.C5_ALGO_BYTES_FULL

    # determinism: a second pass over the same sources gives the same 1.43 GB
    chains2, out2 = _gpu_chains(ctx, bases, geo)
    assert torch.equal(out, out2)
    del chains2, out2

    # LPT over all 3 072 surfaces and 8 ranks -- every surface on exactly one rank,
    # loads within one largest surface of each other ...
    blocks = [shard.block_count(d, d) for _ in range(T) for d in geo["dims"]]
    plan = shard.assign_surfaces(blocks, WORLD)
    flat = sorted(i for p in plan for i in p)
    assert flat == list(range(T * geo["levels"]))
    loads = [sum(blocks[i] for i in p) for p in plan]
    assert sum(loads) == 89478912 and max(loads) - min(loads) <= max(blocks)
    # ... and the 8 per-rank batched encodes (what each rank of the 8-GPU job launches) write the same bytes
    sharded = torch.zeros_like(out)
    offs = np.concatenate([[0], np.cumsum([geo["nbytes"][i % geo["levels"]] for i in range(len(blocks))])])
    params = make_params(Format.BC7, Type.UNorm, QUALITY)
    for p in plan:
        surf = []
        for i in p:
            t, k = divmod(i, geo["levels"])
            d = geo["dims"][k]
            src = bases[t] if k == 0 else chains[t][k - 1]
            surf.append({"pixels": src.data_ptr(), "pixel_type": PixelType.RGBA8 if k == 0 else PixelType.RGBA32F,
                         "width": d, "height": d, "row_pitch_bytes": d * (4 if k == 0 else 16),
                         "out": sharded.data_ptr() + int(offs[i]), "out_capacity": geo["nbytes"][k]})
        surf.sort(key=lambda e: int(e["pixel_type"]))
        ctx.encode_device(surf, params)
    torch.cuda.synchronize()
    assert torch.equal(out, sharded)
    del sharded

    # three textures of the array against the oracle (levels <= 256 x 256 whole, the rest on strips)
    cb = geo["chain_bytes"]
    for t in (0, 101, 255):
        _check_texture_against_oracle(bases[t].cpu().numpy(), [c.cpu().numpy() for c in chains[t]],
                                      out[t * cb:(t + 1) * cb].cpu().numpy(), geo, 256)
