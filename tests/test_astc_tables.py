"""The HIP library's ASTC tables (csrc/astc_tables.h, host code: canonical partition lists,
config lists per class) against the oracle's independently written builder
(oracle/astc_tables.c, oracle/astc_encode.c).  Host-only: runs without a GPU."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O

FOOTPRINTS = [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6),
              (10, 8), (10, 10), (12, 10), (12, 12)]


def _info(fn, bw, bh):
    np3 = (ctypes.c_int*3)()
    nc = (ctypes.c_int*10)()
    modes = (ctypes.c_uint16*(10*64))()
    fn(bw, bh, np3, nc, modes)
    return list(np3), list(nc), np.array(modes, np.uint16)


@pytest.mark.parametrize("bw,bh", FOOTPRINTS)
def test_partition_and_config_tables_match_the_oracle(hip_lib, bw, bh):
    a = _info(O.lib().cfo_astc_table_info, bw, bh)
    b = _info(hip_lib.cfhip_debug_astc_table_info, bw, bh)
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2])
    assert all(50 < v <= 1024 for v in a[0]) and a[1][0] > 8
