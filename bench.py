#!/usr/bin/env python3
"""bench.py -- headline benchmark of the block-encode hot path (BASELINE.json metric).

Default workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): BC7 UNORM,
Texture::Quality::Normal ("quality"), one 4096x4096 RGBA8 synthetic photographic
tile per GPU, resident in HBM when the timed region starts.  A "step" = one pass of
the hot path (cfhip_encode_device) over that tile.  Multi-GPU: independent tiles,
one process per GPU, no data-path collective (blocks/surfaces are independent:
weak scaling); torch.distributed (RCCL) is used only for the barriers and the
max-over-ranks of the elapsed time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launched plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the script starts its N
ranks itself (re-exec under torch.distributed.run on 127.0.0.1); under a launcher it is one rank.
For N > 1 the same JSON line also carries `strong_scaling`: ONE 4096x4096 tile held by rank 0, its
block rows cut by cfhip_shard_rows (Converter.cpp:540-583 parallelises inside a surface), source
rows scattered over RCCL, every rank encoding its range, payload ranges gathered to rank 0 as
device buffers and compared byte for byte with rank 0's own whole-tile encode; the RCCL scatter
is timed beside per-rank pinned-host uploads of the same byte counts.

`--config c5` runs BASELINE.json configs[4] instead: a texture array of 256 x
(2048x2048 RGBA8 + its 12-level mip chain), BC7 Normal, the textures LPT-sharded over the N
ranks (cuttlefish_amd/shard.py), mips generated on the GPU, ONE batched encode per rank, payload
gathered to rank 0 as exact-size device buffers over RCCL, and a sub-sample of the gathered
payload checked byte-for-byte against a local re-encode ("strong" scaling: the batch is fixed).

`--config c3` runs BASELINE.json configs[2]: ASTC 6x6 LDR at Texture::Quality::High on the same kind of tile, one
JSON line of the same shape (roofline of `cfhip_astc_encode_kernel`, cpu_baseline = the oracle on a strip, payload compared).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM
(8 TB/s, /opt/skills/guides/MI355X_MICROARCH.md): algorithmic bytes = 5 B/px
(read RGBA8 once + write 1 B/px payload once) x 16 777 216 px per launch, divided by
the kernel's average duration from hipEvents recorded on the launch stream.  What really
bounds the kernel is VALU issue: `roofline.valu_issue` prices the kernel's OWN instruction mix
(tools/isa_mix.py, read from the library that runs) so its fraction is <= 1 by construction.
`cpu_baseline` times the CPU oracle ("port": our from-spec encoder at the same
search settings, NOT bc7enc_rdo -- its sources are absent) on a bounded strip of
the same tile on this box's usable host cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIZE = 4096
FORMAT_NAME = "BC7"
HBM_PEAK_GBPS = 8000.0
GPU_CLOCK_HZ = 2.4e9          # MI355X peak engine clock (MI355X_MICROARCH.md)
N_SIMD = 1024                 # 256 CUs x 4 SIMDs
ALGO_BYTES_PER_PIXEL = 5.0   # 4 B RGBA8 read + 16 B / 16 px payload write
QNAMES = ["Lowest", "Low", "Normal", "High", "Highest"]
ROUND = 6                     # this build's round: vs_previous_round reads the driver's records BELOW it (a re-run inside
                              # the round, once BENCH_r06.json exists, must not compare the build with itself)
MIN_VS_PREVIOUS = 0.90        # regression gate: a headline below this share of the previous round's is flagged in the
                              # line (`regression_gate.ok`), fails tests/test_gpu_bench.py and, with BENCH_STRICT=1, the run


def usable_cpus() -> int:
    """Host threads this process may really use: the scheduler affinity, capped by the cgroup
    CPU quota (os.cpu_count() reports the machine, not the container's share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, n)


def _previous_round(metric, n_gpus, value):
    """The driver's record of the previous round's run of this metric (BENCH_rNN.json at the repo root, NN < ROUND), so
    that a change of the search budget of a quality level shows in the line itself (round-4 ADVICE: Normal went 4 459 ->
    2 018 Mpixel/s for 0.3 dB and nothing in the line said so).  None when no record of the same metric is there."""
    import glob
    import re
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "BENCH_r[0-9][0-9].json"))):
        if int(re.search(r"BENCH_r(\d\d)\.json$", path).group(1)) >= ROUND:
            continue
        try:
            rec = json.load(open(path)).get("parsed") or {}
            if rec.get("metric") == metric and rec.get("n_gpus", 1) == n_gpus and rec.get("value"):
                best = {"record": os.path.basename(path), "value": rec["value"],
                        "ratio": round(value / rec["value"], 4)}
        except (OSError, ValueError):
            pass
    return best


def _oracle_flags():
    """the compiler flags the timed oracle was built with (oracle/Makefile)"""
    try:
        for ln in open(os.path.join(ROOT, "oracle", "Makefile")):
            if ln.startswith("CFLAGS"):
                return "gcc " + ln.split("=", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(img, gpu_payload, width, quality, budget_s=12.0):
    """Time the CPU oracle on a strip of the tile (>= 10 s of host work on all usable cores) and
    check the GPU payload of the same strip against it.  This is the only place bench.py touches
    oracle/ (test infrastructure)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from cuttlefish_amd import Format, synth

    cores = usable_cpus()
    fmt = int(Format.BC7)
    bw = width // 4
    # single-thread rate on two block rows, then the thread-scaling ladder on a short strip
    t0 = time.perf_counter()
    O.encode(img[:8], fmt, quality=quality, threads=1)
    dt1 = max(time.perf_counter() - t0, 1e-3)
    rate1 = 8 * width / 1e6 / dt1
    ladder = {}
    for t in sorted({max(1, cores // 4), cores}):
        rows_t = max(2, min(SIZE // 4, int(1.5 * t * rate1 * 1e6 / (4 * width))))
        t0 = time.perf_counter()
        O.encode(img[:rows_t * 4], fmt, quality=quality, threads=t)
        ladder[t] = rows_t * 4 * width / 1e6 / max(time.perf_counter() - t0, 1e-3)
    rows = int(min(SIZE // 4, max(2, budget_s * ladder[cores] * 1e6 / (4 * width))))
    y0 = (SIZE // 2) - (rows * 4) // 2          # centred: includes the alpha band
    y0 -= y0 % 4
    y0 = max(0, y0)
    strip = img[y0:y0 + rows * 4]
    t0 = time.perf_counter()
    ref = O.encode(strip, fmt, quality=quality, threads=cores)
    dt = time.perf_counter() - t0
    mpix = strip.shape[0] * strip.shape[1] / 1e6
    got = gpu_payload.reshape(-1, bw * 16)[y0 // 4:y0 // 4 + rows].reshape(-1)
    dec_cpu = O.decode(ref, fmt, width, rows * 4)
    dec_gpu = O.decode(got, fmt, width, rows * 4)
    return {
        "value": round(mpix / dt, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "flags": _oracle_flags(),
        "value_1_thread": round(rate1, 4),
        "scaling": {str(t): round(v, 3) for t, v in ladder.items()},
        "host_threads_visible": os.cpu_count(),
        "sample": "rows %d..%d of the same 4096x4096 tile (%d blocks): %.1f s wall on %d threads = %.0f "
                  "core-seconds of CPU work" % (y0, y0 + rows * 4, rows * bw, dt, cores, dt * cores),
        "psnr_y_cpu": round(synth.psnr_y(strip, dec_cpu), 3),
        "psnr_y_gpu": round(synth.psnr_y(strip, dec_gpu), 3),
        "psnr_rgba_gpu": round(synth.psnr(strip, dec_gpu), 3),
        "gpu_payload_equals_cpu": bool(np.array_equal(ref, got)),
    }


def second_tile(ctx, torch, size, params, steps, stream, check):
    """The same launch on a tile built like a camera picture (synth.photo2: detailed luma over smooth chroma; its
    BC7 blocks split over the modes the way blocks of real photographs do, tests/test_synth.py) -- the headline
    tile of SURVEY 8(d) sends 3 of 4 blocks to mode 5 and hardly any through the second pass of the search.
    A 1024x1024 tile repeated 4 x 4 (the numpy generator takes seconds per Mpixel; a block's cost does not depend
    on its neighbours).  check = True: PSNR and the payload of the first 64 rows against the CPU oracle."""
    import numpy as np
    from cuttlefish_amd import PixelType, synth
    base = synth.photo2(1024, 1024, seed=1)
    rep_ = max(1, size // 1024)
    img = np.ascontiguousarray(np.tile(base, (rep_, rep_, 1)))
    n = img.shape[0]
    src = torch.from_numpy(img).cuda()
    out = torch.empty((n // 4) * (n // 4) * 16, dtype=torch.uint8, device="cuda")
    surf = [{"pixels": src.data_ptr(), "pixel_type": PixelType.RGBA8, "width": n, "height": n,
             "row_pitch_bytes": n * 4, "out": out.data_ptr(), "out_capacity": out.numel()}]
    for _ in range(2):
        ctx.encode_device(surf, params, stream)
    torch.cuda.synchronize()
    ctx.profile_begin()
    for _ in range(steps):
        ctx.encode_device(surf, params, stream)
    torch.cuda.synchronize()
    kernel_ms, launches = ctx.profile_end()
    k = kernel_ms / max(launches, 1)
    res = {"tile": "synth.photo2 1024x1024 (seed 1) repeated %d x %d" % (rep_, rep_), "kernel_ms": round(k, 4),
           "mpixels_per_s_kernel": round(n * n / 1e3 / k, 1), "launches": launches}
    payload = out.cpu().numpy()
    first = payload.reshape(-1, 16)[:, 0]
    lowbit = first & (~first + 1)
    res["bc7_mode_percent"] = {str(m): round(float((lowbit == (1 << m)).mean()) * 100.0, 1) for m in range(8)}
    if check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        from cuttlefish_amd import Format
        rows = 64
        strip = img[:rows * 4, :1024]
        got = np.ascontiguousarray(payload.reshape(n // 4, n // 4, 16)[:rows, :256]).reshape(-1)
        ref = O.encode(strip, int(Format.BC7), quality=int(params.quality), threads=usable_cpus())
        dec = O.decode(got, int(Format.BC7), 1024, rows * 4)
        res["psnr_rgb_gpu"] = round(synth.psnr(strip, dec, slice(0, 3)), 3)
        res["gpu_payload_equals_cpu"] = bool(np.array_equal(ref, got))
    return res


def tolerance_view(ctx, torch, surf, size, stream, normal_value, steps=3):
    """north_star: ">= 50 Mpixel/s BC7 'quality' encode at <= 0.1 dB PSNR delta".  The reference's encoders are absent, so
    the delta is measured against a bound (cfo_bc7_wide_search) on blocks of real photographs, two photograph groups
    (profiles/quality_real.json, written by tools/quality_real.py --json on the oracle whose bytes the kernels emit).  The
    headline level (Normal) is OUTSIDE 0.1 dB there; this key names the lowest level that is inside on both groups and
    times it on the same tile in the same run (a few launches after the headline loop; never part of `value`)."""
    from cuttlefish_amd import Format, Type, make_params
    gaps = None
    try:
        js = json.load(open(os.path.join(ROOT, "profiles", "quality_real.json")))
        gaps = {g: js["BC7 opaque/%s" % g]["pooled_gap_nhh"] for g in ("a", "b")}
    except (OSError, KeyError, ValueError):
        pass
    out = {"target_db": 0.1, "bound": "cfo_bc7_wide_search on blocks of real photographs (tests/golden/real_blocks.npz, groups a / b)",
           "gap_db_normal_high_highest": gaps, "source": "profiles/quality_real.json (tools/quality_real.py)"}
    level = None
    if gaps:
        for k, q in enumerate((2, 3, 4)):
            if all(gaps[g][k] <= 0.1 for g in gaps):
                level = q
                break
    out["headline_level_within_target"] = bool(level == 2)
    out["lowest_level_within_target"] = QNAMES[level] if level is not None else None
    if level is not None and level != 2:
        params = make_params(Format.BC7, Type.UNorm, level)
        ctx.encode_device(surf, params, stream)
        torch.cuda.synchronize()
        ctx.profile_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.encode_device(surf, params, stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kernel_ms, launches = ctx.profile_end()
        out["mpixels_per_s"] = round(size * size * steps / dt / 1e6, 1)
        out["kernel_ms"] = round(kernel_ms / max(launches, 1), 4)
        out["x_north_star_50"] = round(out["mpixels_per_s"] / 50.0, 2)
    elif level == 2:
        out["mpixels_per_s"] = round(normal_value, 1)
        out["x_north_star_50"] = round(normal_value / 50.0, 2)
    return out


PCIE_GBPS = 63.0             # x16 Gen5, one direction (MI355X_MICROARCH.md host link)


def end_to_end(ctx, img, params, device_payload, kernel_ms, reps=5):
    """The REAL boundary: cfhip_encode on host buffers -- upload, kernels, download, sync -- for
    the two layouts a Converter can hand over: the RGBA32F bottom-up scanlines of an Image (Converter.h:52-56,
    Image.cpp:340-343: 16 B/px, negative pitch; quantised to UNORM8 by host threads with toColorBlock's
    arithmetic, S3tcConverter.cpp:97-111, while earlier strips upload and encode) and RGBA8.  Timed after the
    headline loop, never part of `value`."""
    import numpy as np
    n = img.shape[0] * img.shape[1]
    out = {}
    imgf = np.ascontiguousarray((img.astype(np.float32) / 255.0)[::-1])[::-1]     # bottom-up storage, top-down view
    assert imgf.strides[0] < 0
    for name, src, bpp in (("rgba32f_bottom_up", imgf, 16), ("rgba8", img, 4)):
        got = ctx.encode([src], params)[0]          # first call: pinned buffers, worker pool
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            got = ctx.encode([src], params)[0]
            ts.append(time.perf_counter() - t0)
        dt = min(ts)
        wire = n * 4 + got.nbytes                  # what crosses PCIe: UNORM8 texels up, payload down
        floor_pcie = max(n * 4, got.nbytes) / (PCIE_GBPS * 1e9) * 1e3     # full duplex: the larger direction
        floor = max(floor_pcie, kernel_ms)
        out[name] = {"ms": round(dt * 1e3, 3), "ms_median": round(sorted(ts)[len(ts) // 2] * 1e3, 3),
                     "mpix_s": round(n / dt / 1e6, 1), "host_bytes_read": n * bpp, "pcie_bytes": int(wire),
                     "pcie_floor_ms": round(floor_pcie, 3), "kernel_ms": round(kernel_ms, 3),
                     "x_floor": round(dt * 1e3 / floor, 3),
                     "payload_equals_device_path": bool(np.array_equal(got, device_payload))}
    out["note"] = ("floor = max(PCIe at %.0f GB/s for the larger direction, kernel time): the pipeline overlaps host "
                   "gather / quantise, upload, encode and download, so the slowest stage bounds it; raw RGBA32F over "
                   "PCIe would be %d MB = %.2f ms" % (PCIE_GBPS, n * 16 // 1000000, n * 16 / (PCIE_GBPS * 1e9) * 1e3))
    out["host_threads"] = usable_cpus()
    return out


def _cached_photo(synth, size, seed):
    """synth.photo takes ~40 s for 4096x4096 (single-threaded numpy): keep the deterministic tile
    in a git-ignored cache next to the repo so that back-to-back runs (N = 1, 2, 4, 8) reuse it.
    Any cache problem falls back to generating it."""
    import numpy as np
    path = os.path.join(ROOT, ".bench_cache", "photo_%d_seed%d.npy" % (size, seed))
    try:
        if os.path.exists(path):
            img = np.load(path)
            if img.shape == (size, size, 4) and img.dtype == np.uint8:
                return img
    except Exception:
        pass
    img = synth.photo(size, size, seed=seed)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".%d.tmp" % os.getpid()
        with open(tmp, "wb") as f:          # np.save on a path would append ".npy" to the temp name
            np.save(f, img)
        os.replace(tmp, path)
    except Exception:
        pass
    return img


def valu_issue_view(kernel_substr, avg_kernel_s, quality=2):
    """The kernel's mix-weighted VALU issue ceiling.  The instruction mix comes from the device
    code inside the library that just ran (tools/isa_mix.py); the dynamic instruction counts
    (SQ_INSTS_VALU and the per-class counters) and the HBM traffic come from the committed
    rocprofv3 PMC pass of the SAME code (profiles/bc7_pmc.json, matched by the code hash --
    PMC passes cannot run inside the timed benchmark).  Returns (valu_issue, traffic, source)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_mix
        mix = isa_mix.kernel_mix(os.path.join(ROOT, "cuttlefish_amd", "libcuttlefish_hip.so"), kernel_substr)
    except BaseException as e:        # no llvm tools on this box: report nothing rather than guess
        return {"error": "isa mix unavailable: %s" % (e,)}, None, None
    view = {"kernel": mix["kernel"], "code_sha256": mix["code_sha256"],
            "static_valu_fast_share": mix["valu_fast_share"],
            "static_cycles_per_valu_inst": mix["valu_cycles_per_inst"],
            "classes": "2-cycle: fp32 fma/mul/add, plain add/sub/and/or/xor/mov/ashr; 4-cycle: the rest "
                       "(tools/ubench/valu_rate.hip, profiles/r01_valu_rate.txt)"}
    traffic = source = None
    # one committed PMC file per quality level (the dynamic counts depend on it)
    pname = "bc7_pmc.json" if quality == 2 else "bc7_pmc_q%d.json" % quality
    ppath = os.path.join(ROOT, "profiles", pname)
    if os.path.exists(ppath):
        pj = json.load(open(ppath))
        if pj.get("code_sha256") == mix["code_sha256"] and pj.get("quality", 2) == quality:
            traffic = pj.get("traffic_bytes_per_launch")
            source = "from committed profile %s (same code hash %s)" % (pj.get("source"), mix["code_sha256"])
            insts = pj.get("valu_wave_insts_per_launch")
            if insts:
                cyc = pj.get("valu_issue_cycles_per_launch")     # sum over classes of count x class cycles
                if not cyc:
                    cyc = insts * mix["valu_cycles_per_inst"]
                ceiling_s = cyc / N_SIMD / GPU_CLOCK_HZ
                # the same mix at the rates the microbenchmark actually measured for the two classes at
                # 8 waves / SIMD (2.6 and 4.2 cycles, loop overhead included: profiles/r01_valu_rate.txt)
                # -- an upper view of the fraction; `frac` below uses the ideal 2 / 4 cycles
                fast_share = (4.0 - cyc / insts) / 2.0
                cyc_meas = insts * (2.6 * fast_share + 4.2 * (1.0 - fast_share))
                view.update({"wave_insts_per_launch": insts,
                             "issue_cycles_per_launch": int(cyc),
                             "cycles_per_inst": round(cyc / insts, 4),
                             "frac_at_measured_class_rates": round(cyc_meas / N_SIMD / GPU_CLOCK_HZ / avg_kernel_s, 4),
                             "class_source": pj.get("class_source", "static ISA mix"),
                             "ceiling_ms": round(ceiling_s * 1e3, 4),
                             "frac": round(ceiling_s / avg_kernel_s, 4),
                             "note": "frac = (sum over instruction classes of count x issue cycles) / "
                                     "(1024 SIMDs x 2.4 GHz x kernel time): the share of the kernel's "
                                     "time the VALU needs at full issue rate for ITS mix"})
        else:
            view["note"] = "profiles/%s describes other code (hash %s): no PMC-derived figures" % \
                (pname, pj.get("code_sha256"))
    return view, traffic, source


def gpu_texture(torch, size, seed, device):
    """Deterministic synthetic RGBA8 texture generated ON the GPU (C5 needs 256 of them; the numpy
    generator of the C2 tile takes seconds per texture): smooth luminance + chroma fields from
    bilinearly upsampled low-resolution noise, hard-edged rectangles, +-2 LSB grain, opaque alpha
    with one ramped band.  Same seed -> same bytes on every MI355X."""
    g = torch.Generator(device=device)
    g.manual_seed(0xC0FFEE + seed)
    lo = torch.rand((1, 4, 33, 33), generator=g, device=device)
    lum = torch.rand((1, 1, 9, 9), generator=g, device=device)
    up = torch.nn.functional.interpolate
    f = 0.45 * up(lo, size=(size, size), mode="bilinear", align_corners=True) + \
        0.55 * up(lum, size=(size, size), mode="bilinear", align_corners=True)
    f = f[0].permute(1, 2, 0).contiguous()
    rects = torch.randint(0, size, (24, 4), generator=g, device=device)
    cols = torch.rand((24, 3), generator=g, device=device)
    for r in range(24):
        x0, y0, w, h = [int(v) for v in rects[r].tolist()]
        f[y0:y0 + 1 + h // 6, x0:x0 + 1 + w // 6, :3] = cols[r]
    f[..., 3] = 1.0
    band = slice(size // 2, size // 2 + size // 8)
    f[band, :, 3] = torch.linspace(0.0, 1.0, size, device=device)[None, :]
    grain = (torch.rand((size, size, 4), generator=g, device=device) * 5.0 - 2.0).floor()
    grain[..., 3] = 0.0
    return (f * 255.0 + grain).round().clamp(0, 255).to(torch.uint8).contiguous()


C5_ALGO_BYTES_FULL = 7158285312      # 256 x (5 592 405 px x 4 B read + 349 527 blocks x 16 B written)


def c5_geometry(n):
    """Level sizes, payload bytes per level and block count of ONE n x n texture with its full chain
    (Texture::generateMipmaps down to 1 x 1, lib/src/Texture.cpp:1320-1514)."""
    from cuttlefish_amd import Format, Type, payload_size, shard
    levels = n.bit_length()
    dims = [max(1, n >> k) for k in range(levels)]
    nbytes = [payload_size(Format.BC7, Type.UNorm, d, d) for d in dims]
    return {"levels": levels, "dims": dims, "nbytes": nbytes, "chain_bytes": sum(nbytes),
            "px_chain": sum(d * d for d in dims), "blocks_chain": sum(shard.block_count(d, d) for d in dims)}


def c5_surfaces(geo, base_list, chain_list, out_buf):
    """The [texture][mip] surfaces of Converter::convert's loop (Converter.cpp:521-527) as one
    cfhip_encode_device call: texture i's payload at i * chain_bytes, its levels in mip order."""
    from cuttlefish_amd import PixelType
    dims, nbytes, chain_bytes = geo["dims"], geo["nbytes"], geo["chain_bytes"]
    s = []
    for i in range(len(base_list)):
        off = i * chain_bytes
        for k, d in enumerate(dims):
            src = base_list[i] if k == 0 else chain_list[i][k - 1]
            s.append({"pixels": src.data_ptr(), "pixel_type": PixelType.RGBA8 if k == 0 else PixelType.RGBA32F,
                      "width": d, "height": d, "row_pitch_bytes": d * (4 if k == 0 else 16),
                      "out": out_buf.data_ptr() + off, "out_capacity": nbytes[k]})
            off += nbytes[k]
    s.sort(key=lambda e: int(e["pixel_type"]))            # one batched launch per source type
    return s


def c5_cpu_texture(base_np, geo, quality, threads):
    """ONE texture of C5 on the CPU oracle (test infrastructure): Texture::generateMipmaps (Box, linear
    space) then the BC7 encode of every level -> (payload bytes of the chain, seconds in the mip
    generation, seconds in the encode)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from cuttlefish_amd import Format
    t0 = time.perf_counter()
    chain = O.mip_chain(base_np, geo["levels"], filter=0, color_space=0)
    t1 = time.perf_counter()
    parts = [O.encode(base_np, int(Format.BC7), quality=quality, threads=threads)]
    for lvl in chain[1:]:
        parts.append(O.encode(lvl, int(Format.BC7), quality=quality, threads=threads))
    t2 = time.perf_counter()
    return np.concatenate(parts), t1 - t0, t2 - t1


def c5_cpu_baseline(torch, bases, mine, out, geo, quality, budget_s=12.0):
    """cpu_baseline of the C5 line: whole textures of the same batch through the CPU oracle (mip chain
    + encode of all 12 levels, the job model of Converter.cpp:521-589 with `cores` threads) until
    `budget_s` of wall time is spent, each compared byte for byte with the GPU payload of that texture."""
    import numpy as np
    cores = usable_cpus()
    done, t_mip, t_enc, equal = [], 0.0, 0.0, True
    cb = geo["chain_bytes"]
    t_start = time.perf_counter()
    for pos, t in enumerate(mine):
        ref, a, b = c5_cpu_texture(bases[pos].cpu().numpy(), geo, quality, cores)
        t_mip += a
        t_enc += b
        got = out[pos * cb:(pos + 1) * cb].cpu().numpy()
        equal = equal and bool(np.array_equal(ref, got))
        done.append(t)
        if time.perf_counter() - t_start > budget_s:
            break
    mpix = len(done) * geo["px_chain"] / 1e6
    return {"value": round(mpix / (t_mip + t_enc), 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "encode_only_value": round(mpix / t_enc, 4),
            "sample": "textures %s of the same batch, whole chains (%d levels, %.2f Mpixel each): mip generation "
                      "%.1f s on 1 thread + BC7 encode %.1f s on %d threads" %
                      (done, geo["levels"], geo["px_chain"] / 1e6, t_mip, t_enc, cores),
            "host_threads_visible": os.cpu_count(),
            "gpu_payload_equals_cpu": equal, "bytes_compared": len(done) * cb}


def run_c5(args, rank, local_rank, world, backend):
    import numpy as np
    import torch
    import torch.distributed as dist
    from cuttlefish_amd import ColorSpace, Context, Format, PixelType, Type, make_params, payload_size, shard

    dev = torch.device("cuda", local_rank)
    n, T = args.tex_size, args.textures
    geo = c5_geometry(n)
    levels, dims, nbytes = geo["levels"], geo["dims"], geo["nbytes"]
    px_chain, chain_bytes, blocks_chain = geo["px_chain"], geo["chain_bytes"], geo["blocks_chain"]
    plan = shard.assign_surfaces([blocks_chain] * T, world)      # units = textures (a chain stays on its rank)
    mine = plan[rank]
    ctx = Context(local_rank)
    # A stream of its own, current for the whole run: on a real stream the library only ENQUEUES (5 632 mip
    # kernels + 2 encode launches per step for 256 textures) and the host runs ahead into the next step
    # while the GPU encodes; on torch's legacy default stream (handle 0 = "the context's stream") every
    # call would synchronise -- 257 host round trips per step, 35 ms of 356.
    run_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(run_stream)
    stream = run_stream.cuda_stream
    params = make_params(Format.BC7, Type.UNorm, args.quality)

    bases = [gpu_texture(torch, n, t, dev) for t in mine]
    torch.cuda.synchronize()          # the textures exist before any other stream reads them
    chains = [[torch.empty((d, d, 4), dtype=torch.float32, device=dev) for d in dims[1:]] for _ in mine]
    out = torch.empty(len(mine) * chain_bytes, dtype=torch.uint8, device=dev)

    def surfaces(base_list, chain_list, out_buf):
        return c5_surfaces(geo, base_list, chain_list, out_buf)
    surf = surfaces(bases, chains, out)
    base_ptrs = [b.data_ptr() for b in bases]
    chain_ptrs = [[c.data_ptr() for c in ch] for ch in chains]
    sizes = [len(plan[r]) * chain_bytes for r in range(world)]

    def step(gather=True):
        if mine:
            # the rank's textures are the layers of one array: one launch per pass and level for all of them
            ctx.generate_mips_array_device(base_ptrs, PixelType.RGBA8, n, n, n * 4, chain_ptrs,
                                           color_space=ColorSpace.Linear, filter=0, stream=stream)
        if surf:
            ctx.encode_device(surf, params, stream)
        if gather and world > 1:
            torch.cuda.current_stream().synchronize()
            return shard.exchange(out, sizes, rank, world, dst=0)
        return None

    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    ctx.profile_begin()
    t0 = time.perf_counter()
    parts = None
    for _ in range(args.steps):
        parts = step()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.profile_end()
    # encode-only time of this rank's share (no gather), for the per-rank load table
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step(gather=False)
    torch.cuda.synchronize()
    local_ms = (time.perf_counter() - t1) * 1e3
    stats = torch.tensor([elapsed, local_ms, float(len(mine))], dtype=torch.float64,
                         device=dev if backend == "nccl" else "cpu")
    allstats = [stats.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(allstats, stats)
    elapsed = max(float(s[0]) for s in allstats)

    if rank == 0:
        # byte-equality of the batched / sharded result against a per-texture re-encode on rank 0
        # (texture regenerated from its seed, mips regenerated, ONE texture per encode call): the
        # first and last texture of rank 0's own share -- so the check is not vacuous at N = 1 --
        # and, for N > 1, of rank 1's and the last rank's share as they arrived through the gather
        check = {"textures_checked": [], "equal": True}
        local_view = out if parts is None else parts[0]
        todo = [(0, pos, local_view) for pos in sorted({0, len(plan[0]) - 1}) if plan[0]]
        if world > 1 and parts is not None:
            for r in sorted({1, world - 1}):
                todo += [(r, pos, parts[r]) for pos in sorted({0, len(plan[r]) - 1}) if plan[r]]
        for r, pos, buf in todo:
            t = plan[r][pos]
            b = [gpu_texture(torch, n, t, dev)]
            c = [[torch.empty((d, d, 4), dtype=torch.float32, device=dev) for d in dims[1:]]]
            o = torch.empty(chain_bytes, dtype=torch.uint8, device=dev)
            # torch produced the texture on ITS stream; the library reads it on the stream it is given
            # (torch's legacy default stream has handle 0 = "the context's own stream"): finish first
            torch.cuda.synchronize()
            ctx.generate_mips_device(b[0].data_ptr(), PixelType.RGBA8, n, n, n * 4,
                                     [x.data_ptr() for x in c[0]], color_space=ColorSpace.Linear,
                                     filter=0, stream=stream)
            ctx.encode_device(surfaces(b, c, o), params, stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(o, buf[pos * chain_bytes:(pos + 1) * chain_bytes]))
            check["textures_checked"].append({"texture": t, "owner_rank": r, "equal": same})
            check["equal"] = check["equal"] and same
        pixels = float(T) * px_chain
        line = {
            "metric": "Mpixels/s encode, BC7 texture-array batch (256 x 2048x2048 RGBA8 mip chains)",
            "value": round(pixels * args.steps / elapsed / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic (generated on the GPU from per-texture seeds)",
            "config": {"workload": "C5: %d textures x (%dx%d RGBA8 + %d-level Box mip chain generated on the "
                                   "GPU), BC7 UNORM Texture::Quality::%s, textures LPT-sharded over the ranks, "
                                   "one batched encode per rank, exact-size device gather to rank 0 over %s"
                                   % (T, n, n, levels, QNAMES[args.quality], "RCCL" if backend == "nccl" else backend),
                       "format": FORMAT_NAME, "quality": args.quality, "textures": T,
                       "surfaces": T * levels, "blocks": T * blocks_chain,
                       "parallelism": "texture-sharded x%d" % world},
            "per_rank": [{"rank": r, "textures": int(allstats[r][2]), "encode_ms": round(float(allstats[r][1]), 3),
                          "gather_bytes": sizes[r]} for r in range(world)],
            "kernel_ms_rank0": round(kernel_ms / max(args.steps, 1), 3),
            "sharded_equals_local": check,
        }
        # the dominant kernel is the BC7 block encoder: two batched launches per step (the RGBA8 level-0
        # surfaces, the RGBA32F mip surfaces); its hipEvent time on the launch stream, rank 0's share
        step_kernel_s = kernel_ms / 1e3 / max(args.steps, 1)
        algo = len(mine) * (px_chain * 4 + chain_bytes)      # 4 B/px read + payload written
        achieved = algo / step_kernel_s / 1e9 if step_kernel_s > 0 else 0.0
        line["roofline"] = {
            "bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 8), "traffic": None, "kernel": ctx.last_kernel_name(),
            "kernel_ms_per_step": round(step_kernel_s * 1e3, 4), "launches_per_step": launches // max(args.steps, 1),
            "algorithmic_bytes_per_step": int(algo),
            "algorithmic_bytes_full_config": C5_ALGO_BYTES_FULL,
            "note": "rank 0's share; algorithmic bytes = 4 B/px source read + 1 B/px payload write over every "
                    "level (SURVEY 8d); the mip levels really travel as RGBA32F (16 B/px, the reference's RGBAF), "
                    "so the bytes the launches move are higher; VALU-issue-bound search (DESIGN.md section 4.1)"}
        line["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = c5_cpu_baseline(torch, bases, mine, out, geo, args.quality)
        print(json.dumps(line), flush=True)
    ctx.close()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks here (one process per GPU,
    RCCL rendezvous on 127.0.0.1) and pass rank 0's JSON line through."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def strong_c2(args, ctx, torch, dist, rank, world, backend, img, local_out):
    """Strong-scaling leg of C2: rank 0's tile split by block rows over the ranks."""
    from cuttlefish_amd import Format, PixelType, Type, make_params, shard
    size = args.size
    dev = torch.device("cuda", torch.cuda.current_device())
    params = make_params(Format.BC7, Type.UNorm, args.quality)
    full = torch.from_numpy(img).to(dev).reshape(-1) if rank == 0 else None
    ranges = shard.row_ranges(size, 4, world)
    y0, y1, _, _ = ranges[rank]
    my_bytes = (y1 - y0) * size * 4

    def one(timings=None):
        return shard.encode_rows_sharded_device(ctx, full, size, size, PixelType.RGBA8, params, rank, world,
                                                src=0, dst=0, timings=timings)

    def barrier():
        dist.barrier()
    for _ in range(max(1, args.warmup)):
        got = one()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = one()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # phase split of one more step (synchronised between the phases: not part of `elapsed`)
    tm = {}
    one(tm)
    # the alternative input leg: every rank uploads ITS byte count from pinned host memory
    pinned = torch.from_numpy(img.reshape(-1)[:max(my_bytes, 1)].copy()).pin_memory()
    dst = torch.empty(max(my_bytes, 1), dtype=torch.uint8, device=dev)
    dst.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize(); barrier()
    t1 = time.perf_counter()
    dst.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    h2d = time.perf_counter() - t1
    cpu = backend != "nccl"
    stats = torch.tensor([elapsed, tm.get("scatter_s", 0.0), tm.get("encode_s", 0.0), tm.get("gather_s", 0.0), h2d],
                         dtype=torch.float64, device="cpu" if cpu else dev)
    allstats = [stats.clone() for _ in range(world)]
    dist.all_gather(allstats, stats)
    if rank != 0:
        return None
    elapsed = max(float(a[0]) for a in allstats)
    equal = bool(torch.equal(got, local_out))
    return {
        "workload": "ONE %dx%d RGBA8 tile held by rank 0, block rows split over %d ranks by cfhip_shard_rows; "
                    "per step: scatter of source rows (%s) -> BC7 encode of the rank's rows -> exact-size "
                    "device-buffer gather of the payload to rank 0" % (size, size, world, "RCCL" if not cpu else backend),
        "scaling": "strong", "value": round(size * size * args.steps / elapsed / 1e6, 3), "unit": "Mpixels/s",
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "phases_ms_max_over_ranks": {"scatter": round(max(float(a[1]) for a in allstats) * 1e3, 4),
                                     "encode": round(max(float(a[2]) for a in allstats) * 1e3, 4),
                                     "gather": round(max(float(a[3]) for a in allstats) * 1e3, 4)},
        "input_leg_alternative": {"pinned_h2d_ms_max_over_ranks": round(max(float(a[4]) for a in allstats) * 1e3, 4),
                                  "bytes_per_rank": [(r1 - r0) * size * 4 for (r0, r1, _, _) in ranges],
                                  "note": "each rank uploads its own row range from pinned host memory instead "
                                          "of receiving it from rank 0"},
        "block_rows_per_rank": [b - a for (_, _, a, b) in ranges],
        "sharded_equals_local": {"equal": equal, "bytes_compared": int(got.numel()),
                                 "against": "rank 0's whole-tile encode of the same tile"},
    }


def run_c3(args, rank, local_rank, world, backend):
    """BASELINE.json configs[2] (SURVEY 8d "C3"): ASTC 6x6 LDR, Texture::Quality::High ("thorough"), one 4096x4096 RGBA8
    tile per GPU resident in HBM, weak scaling like the headline.  One JSON line: `roofline` prices the kernel against HBM
    (algorithmic bytes = the source read once + 16 B per 6x6 block written once) and carries `valu_busy`, the measure
    that actually bounds this kernel, replayed from the committed PMC pass (profiles/r06_astc_lds_pmc.txt: PMC passes
    cannot run inside a timed benchmark); `cpu_baseline` = the oracle on a strip of the same tile, payload compared."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cuttlefish_amd import Context, Format, PixelType, Type, make_params, synth

    size = args.size
    q = 3 if args.quality == 2 else args.quality          # (--quality defaults to the headline's Normal; C3 is High)
    img = _cached_photo(synth, size, 1 + rank)
    src = torch.from_numpy(img).cuda()
    bx, by = (size + 5) // 6, (size + 5) // 6
    out = torch.empty(bx * by * 16, dtype=torch.uint8, device="cuda")
    params = make_params(Format.ASTC_6x6, Type.UNorm, q)
    surf = [{"pixels": src.data_ptr(), "pixel_type": PixelType.RGBA8, "width": size, "height": size,
             "row_pitch_bytes": size * 4, "out": out.data_ptr(), "out_capacity": out.numel()}]
    ctx = Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        ctx.encode_device(surf, params, stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.encode_device(surf, params, stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.profile_end()
    per_rank = [{"rank": 0, "kernel_ms": round(kernel_ms / max(launches, 1), 4)}]
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms / max(launches, 1)], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
        allt = [t.clone() for _ in range(world)]
        dist.all_gather(allt, t)
        elapsed = max(float(a[0]) for a in allt)
        per_rank = [{"rank": r, "kernel_ms": round(float(allt[r][1]), 4)} for r in range(world)]
    if rank == 0:
        value = float(size * size) * world * args.steps / elapsed / 1e6
        avg_kernel_s = kernel_ms / 1e3 / max(launches, 1)
        algo_bytes = size * size * 4 + bx * by * 16
        achieved = algo_bytes / avg_kernel_s / 1e9
        line = {"metric": "Mpixels/s encode, ASTC 6x6 4096x4096 RGBA8", "value": round(value, 3), "unit": "Mpixels/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": "ASTC 6x6 LDR Texture::Quality::%s, one %dx%d RGBA8 synthetic photo tile per GPU, "
                                       "resident in HBM (BASELINE config 3)" % (QNAMES[q], size, size),
                           "format": "ASTC_6x6", "quality": q, "blocks_per_launch": bx * by,
                           "parallelism": "surface-per-gpu x%d" % world},
                "per_rank": per_rank,
                "roofline": {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBPS, 8), "traffic": None,
                             "kernel": ctx.last_kernel_name(), "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                             "launches": launches, "algorithmic_bytes_per_launch": int(algo_bytes),
                             "note": "a VALU-issue-bound search: the HBM fraction is tiny by construction; what bounds the kernel "
                                     "is valu_busy (DESIGN.md 4.5)"}}
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "astc_c3_pmc.json")))
            line["roofline"]["traffic"] = pj.get("traffic_bytes_per_launch")
            line["roofline"]["valu_busy"] = pj.get("valu_busy")
            line["roofline"]["lds_bank_conflict_share"] = pj.get("lds_bank_conflict_share")
            line["roofline"]["pmc_source"] = pj.get("source")
            line["roofline"]["traffic_measured_in_this_run"] = False
        except (OSError, ValueError):
            pass
        # north_star's 0.1 dB at C3's level: the same view as the headline's `tolerance` key, read from the same table
        try:
            js = json.load(open(os.path.join(ROOT, "profiles", "quality_real.json")))
            gaps = {g: js["ASTC 6x6/%s" % g]["pooled_gap_nhh"] for g in ("a", "b")}
            inside = [QNAMES[lv] for k, lv in enumerate((2, 3, 4)) if all(gaps[g][k] <= 0.1 for g in gaps)]
            line["tolerance"] = {"target_db": 0.1, "bound": "cfo_astc_wide_search on blocks of real photographs "
                                                             "(tests/golden/real_blocks.npz, groups a / b)",
                                 "gap_db_normal_high_highest": gaps, "this_level_within_target": bool(QNAMES[q] in inside),
                                 "lowest_level_within_target": inside[0] if inside else None,
                                 "source": "profiles/quality_real.json (tools/quality_real.py)"}
        except (OSError, KeyError, ValueError):
            line["tolerance"] = None
        if world == 1 and not args.no_cpu_baseline and size == SIZE:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            cores = usable_cpus()
            fmt = int(Format.ASTC_6x6)
            t0 = time.perf_counter()
            O.encode(img[:48], fmt, quality=q, threads=cores)
            rate = 48 * size / 1e6 / max(time.perf_counter() - t0, 1e-3)
            rows = int(min(by, max(8, 10.0 * rate * 1e6 / (6 * size))))
            strip = img[:rows * 6]
            t0 = time.perf_counter()
            ref = O.encode(strip, fmt, quality=q, threads=cores)
            dt = time.perf_counter() - t0
            got = out.cpu().numpy()[:rows * bx * 16]
            line["cpu_baseline"] = {"value": round(rows * 6 * size / 1e6 / dt, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                                    "flags": _oracle_flags(),
                                    "sample": "rows 0..%d of the same tile (%d blocks): %.1f s wall on %d threads" % (rows * 6, rows * bx, dt, cores),
                                    "gpu_payload_equals_cpu": bool(np.array_equal(ref, got))}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    ctx.close()


def run_c2(args, rank, local_rank, world, backend):
    import numpy as np
    import torch
    import torch.distributed as dist
    from cuttlefish_amd import Context, Format, PixelType, Type, make_params, synth

    size = args.size
    # every rank encodes its own tile (independent surfaces; weak scaling)
    img = _cached_photo(synth, size, 1 + rank)
    src = torch.from_numpy(img).cuda()
    out = torch.empty((size // 4) * (size // 4) * 16, dtype=torch.uint8, device="cuda")
    params = make_params(Format.BC7, Type.UNorm, args.quality)
    surf = [{"pixels": src.data_ptr(), "pixel_type": PixelType.RGBA8, "width": size,
             "height": size, "row_pitch_bytes": size * 4, "out": out.data_ptr(),
             "out_capacity": out.numel()}]
    ctx = Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        ctx.encode_device(surf, params, stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.encode_device(surf, params, stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.profile_end()

    per_rank = [{"rank": 0, "kernel_ms": round(kernel_ms / max(launches, 1), 4)}]
    strong = None
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms / max(launches, 1)], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
        allt = [t.clone() for _ in range(world)]
        dist.all_gather(allt, t)
        elapsed = max(float(a[0]) for a in allt)
        per_rank = [{"rank": r, "kernel_ms": round(float(allt[r][1]), 4)} for r in range(world)]

    def guarded_strong(line):
        """The strong-scaling leg runs AFTER the weak line is complete and can never take it down: an
        exception is reported in the line, and a watchdog prints the weak line and leaves if the leg
        does not finish (a collective that hangs on hardware this code has not met would otherwise
        cost the whole measurement).  Exactly ONE line is ever printed (a lock and a flag decide between
        the watchdog and the main thread); with BENCH_STRONG_STRICT=1 a leg that failed leaves with exit status 3."""
        import threading
        done = threading.Event()
        printed = threading.Lock()
        limit = float(os.environ.get("BENCH_STRONG_TIMEOUT_S", "240"))

        def leave(note):
            if rank == 0 and printed.acquire(blocking=False):      # never released: one line per job
                line["strong_scaling"] = {"error": note}
                print(json.dumps(line), flush=True)
            # the other ranks may sit in a collective: no orderly teardown.  The weak line stands and carries the
            # error note; the exit status stays 0 unless BENCH_STRONG_STRICT=1 asks for 3 (a harness that
            # discards the output of a failed command would otherwise lose the headline measurement to a
            # failure of the extra leg)
            os._exit(3 if os.environ.get("BENCH_STRONG_STRICT", "0") == "1" else 0)

        def watchdog():
            if not done.wait(limit):
                leave("the strong-scaling leg did not finish within %.0f s; the weak-scaling line stands" % limit)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            res = strong_c2(args, ctx, torch, dist, rank, world, backend, img, out)
        except Exception as e:                      # noqa: BLE001 -- reported, not hidden
            done.set()
            leave("%s: %s" % (type(e).__name__, e))
        done.set()
        if rank == 0 and not printed.acquire(blocking=False):
            os._exit(3 if os.environ.get("BENCH_STRONG_STRICT", "0") == "1" else 0)   # the watchdog printed while the leg was finishing
        return res

    if rank != 0 and world > 1 and not args.no_strong:
        guarded_strong(None)
    if rank == 0:
        pixels_per_step = float(size * size) * world
        value = pixels_per_step * args.steps / elapsed / 1e6
        avg_kernel_s = kernel_ms / 1e3 / max(launches, 1)
        algo_bytes = ALGO_BYTES_PER_PIXEL * size * size
        achieved = algo_bytes / avg_kernel_s / 1e9
        line = {
            "metric": "Mpixels/s encode, BC7 4096x4096 RGBA8",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BC7 UNORM Texture::Quality::%s, one %dx%d RGBA8 synthetic "
                                   "photo tile per GPU, resident in HBM" % (QNAMES[args.quality], size, size),
                       "format": FORMAT_NAME, "quality": args.quality,
                       "blocks_per_launch": (size // 4) ** 2, "parallelism": "surface-per-gpu x%d"
                       % world},
            "per_rank": per_rank,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 8),
                         "traffic": None, "kernel": ctx.last_kernel_name(),
                         "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": launches,
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "note": "VALU-issue-bound search (see valu_issue); the HBM fraction is "
                                 "tiny by construction (DESIGN.md roofline section)"},
        }
        if size == SIZE:
            ksub = "cfhip_bc7_encode_kernelILi0ELb1ELb%dE" % (1 if args.quality >= 3 else 0)
            view, traffic, source = valu_issue_view(ksub, avg_kernel_s, args.quality)
            line["roofline"]["valu_issue"] = view
            if traffic is not None:
                line["roofline"]["traffic"] = traffic
                line["roofline"]["traffic_source"] = source
            # PMC passes cannot run inside the timed region: traffic and the dynamic instruction
            # counts are REPLAYED from the committed profile of the same code hash, never measured here
            line["roofline"]["traffic_measured_in_this_run"] = False
        if world == 1 and not args.no_cpu_baseline and size == SIZE:
            payload = out.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline(img, payload, size, args.quality)
        else:
            line["cpu_baseline"] = None
        if size == SIZE:
            line["vs_previous_round"] = _previous_round(line["metric"], world, value)
            if line["vs_previous_round"] and args.quality == 2:
                line["regression_gate"] = {"min_ratio": MIN_VS_PREVIOUS, "ok": line["vs_previous_round"]["ratio"] >= MIN_VS_PREVIOUS}
        # (the headline level's payload, taken before the tolerance leg encodes another level into the same buffer)
        device_payload = out.cpu().numpy() if (world == 1 and not args.no_end_to_end) else None
        if world == 1 and size == SIZE and args.quality == 2 and not args.no_tolerance:
            line["tolerance"] = tolerance_view(ctx, torch, surf, size, stream, value)
        if world == 1 and size == SIZE and not args.no_second_tile:
            line["second_tile"] = second_tile(ctx, torch, size, params, max(3, args.steps // 2), stream, not args.no_cpu_baseline)
            line["second_tile"]["vs_headline_tile"] = round(line["second_tile"]["kernel_ms"] / (avg_kernel_s * 1e3), 3)
        if world == 1 and not args.no_end_to_end:
            torch.cuda.synchronize()
            line["end_to_end"] = end_to_end(ctx, img, params, device_payload, avg_kernel_s * 1e3)
        if world > 1 and not args.no_strong:
            strong = guarded_strong(line)
        if strong is not None:
            line["strong_scaling"] = strong
        print(json.dumps(line), flush=True)
        gate = line.get("regression_gate")
        if gate and not gate["ok"]:
            print("bench.py: headline %.1f is below %.2f x the previous round's record (%s)" %
                  (value, MIN_VS_PREVIOUS, line["vs_previous_round"]), file=sys.stderr, flush=True)
            if os.environ.get("BENCH_STRICT", "0") == "1":
                ctx.close()
                raise SystemExit(4)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--quality", type=int, default=2)
    ap.add_argument("--size", type=int, default=SIZE)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c5"])
    ap.add_argument("--textures", type=int, default=256, help="c5: textures in the array")
    ap.add_argument("--tex-size", type=int, default=2048, help="c5: base level size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="c2, N = 1: skip the host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--no-strong", action="store_true", help="c2, N > 1: skip the row-split strong-scaling leg")
    ap.add_argument("--no-tolerance", action="store_true", help="c2, N = 1: skip the `tolerance` leg (the lowest level within 0.1 dB, timed)")
    ap.add_argument("--no-second-tile", action="store_true",
                    help="c2, N = 1: skip the camera-like second tile (profile passes: the kernel statistics and PMC "
                         "averages of the run must be the headline tile's alone)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.config == "c2" else (8 if args.config == "c3" else 3)
    if args.warmup is None:
        args.warmup = 3 if args.config == "c2" else 1

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: become one (the ranks re-enter main() with WORLD_SIZE set)
        if not os.environ.get("CFHIP_LIB"):
            from cuttlefish_amd import build as _build
            if _build.is_stale():
                _build.build()
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_rank == 0 and not os.environ.get("CFHIP_LIB"):
        # never measure a library older than its sources (content hash, cuttlefish_amd/build.py):
        # a no-op after `__graft_entry__.build()`, one hipcc run otherwise
        from cuttlefish_amd import build as _build
        if _build.is_stale():
            _build.build()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # BENCH_DIST_BACKEND=gloo is a TEST HOOK: it lets the N-rank flow (rendezvous, barriers,
    # max-over-ranks, rank-0 JSON) run on a box with fewer GPUs than ranks, ranks sharing devices.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        if args.config == "c5":
            run_c5(args, rank, local_rank, world, backend)
        elif args.config == "c3":
            run_c3(args, rank, local_rank, world, backend)
        else:
            run_c2(args, rank, local_rank, world, backend)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
