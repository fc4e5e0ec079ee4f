"""bench.py end to end on the GPU box: the N-rank flow launched the way the driver launches N = 1
-- plain `python bench.py --gpus N` -- must start its own ranks, and the strong-scaling / C5
self-checks must actually compare bytes.  On a one-GPU box the ranks share the device through
the BENCH_DIST_BACKEND=gloo test hook (the data path is the same code; only the wire differs)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["BENCH_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_bench_gpus2_self_launches_and_reports_strong_scaling(hip_lib):
    line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "1024", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert [p["rank"] for p in line["per_rank"]] == [0, 1]
    assert all(p["kernel_ms"] > 0 for p in line["per_rank"])
    st = line["strong_scaling"]
    assert st["scaling"] == "strong" and st["value"] > 0
    assert st["block_rows_per_rank"] == [128, 128]
    assert st["sharded_equals_local"]["equal"] is True
    assert st["sharded_equals_local"]["bytes_compared"] == 256*256*16
    assert set(st["phases_ms_max_over_ranks"]) == {"scatter", "encode", "gather"}


@pytest.mark.timeout(1200)
def test_bench_c5_checks_textures_at_one_and_two_ranks(hip_lib):
    one = _run(["--config", "c5", "--gpus", "1", "--textures", "4", "--tex-size", "256", "--steps", "1"])
    chk = one["sharded_equals_local"]
    assert chk["equal"] is True and len(chk["textures_checked"]) >= 2
    two = _run(["--config", "c5", "--gpus", "2", "--textures", "8", "--tex-size", "256", "--steps", "1"])
    chk = two["sharded_equals_local"]
    assert two["n_gpus"] == 2 and chk["equal"] is True
    assert {c["owner_rank"] for c in chk["textures_checked"]} == {0, 1}
    assert all(c["equal"] for c in chk["textures_checked"])


@pytest.mark.timeout(600)
def test_a_stuck_strong_scaling_leg_cannot_take_the_weak_line_down(hip_lib):
    """The strong-scaling leg is an extra: with its watchdog set to fire at once, the job still prints
    exactly one line, exit code 0, with the weak-scaling value and an error note in place of the leg."""
    line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "1024", "--no-cpu-baseline"],
                extra_env={"BENCH_STRONG_TIMEOUT_S": "0.01"})
    assert line["n_gpus"] == 2 and line["value"] > 0 and len(line["per_rank"]) == 2
    st = line["strong_scaling"]
    assert "error" in st or st.get("value", 0) > 0
