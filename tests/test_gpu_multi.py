"""cfhip_encode_multi: several contexts of ONE process share the surfaces of a call (the
reference's CLI / library are a single process).  On a one-GPU box the contexts live on the same
device, which exercises the assignment, the worker threads and the per-thread device selection;
with more devices visible each context gets its own."""
import numpy as np
import pytest

from cuttlefish_amd import Context, Format, Type, api, make_params, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt,quality", [(Format.BC7, 2), (Format.ASTC_6x6, 1), (Format.ETC2_R8G8B8A8, 2)])
def test_multi_context_equals_single_context(gpu_ctx, fmt, quality):
    params = make_params(fmt, Type.UNorm, quality)
    chain = [synth.photo(max(1, 96 >> i), max(1, 64 >> i), seed=200 + i) for i in range(7)]
    chain += [synth.photo(52, 36, seed=300), synth.photo(5, 3, seed=301)]
    want = gpu_ctx.encode(chain, params)
    ndev = api.device_count()
    others = [Context(d % ndev) for d in range(1, 3)]
    try:
        got = gpu_ctx.encode_multi(others, chain, params)
    finally:
        for c in others:
            c.close()
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_multi_reports_the_failure_of_any_share(gpu_ctx):
    params = make_params(Format.ASTC_6x6, Type.UFloat, 2)      # refused: ASTC takes RGBA8 or RGBA32F pixels
    imgs = [(synth.photo(16, 16, seed=1).astype(np.float32)/255.0).astype(np.float16) for _ in range(3)]
    other = Context(0)
    try:
        with pytest.raises(api.CfhipError):
            gpu_ctx.encode_multi([other], imgs, params)
    finally:
        other.close()
