"""Register budget of the planned kernels: the fp32 hybrid kernel is built for five waves per SIMD -- round
2 lost 10 % on the latency-bound low-degree graphs when a refactoring pushed it from 94 to 100 registers
(profiles/r02/ab_occupancy.log) -- so the compiler's own resource report is part of the CPU suite:
hipcc -Rpass-analysis=kernel-resource-usage on the kernel translation units (cross-compiles without a GPU, ~1 min)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hc-spmm_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resource_usage(unit):
    cmd = [HIPCC, "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, unit), "-o", os.devnull]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=CSRC)
    assert r.returncode == 0, r.stdout[-2000:]
    out, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("agprs", r"\bAGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


@pytest.fixture(scope="module")
def usage():
    """Both translation units, compiled side by side (the 16-bit one takes about a minute)."""
    from concurrent.futures import ThreadPoolExecutor
    units = ("spmm_kernels.hip", "spmm_kernels_h16.hip", "fused_rows.hip")
    with ThreadPoolExecutor(3) as ex:
        return dict(zip(units, ex.map(_resource_usage, units)))


def _demangled_args(name):
    """hybrid_plan_kernel<E, L, VEC, UNROLL, MINW, FUSED> -> (E, L, VEC, UNROLL, MINW, FUSED) from the mangled name."""
    m = re.search(r"hybrid_plan_kernelINS_(\w+?)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])E", name)
    return (m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6) == "1") if m else None


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_fp32_planned_kernels_keep_five_waves_per_simd(usage):
    usage = usage["spmm_kernels.hip"]
    plain = {n: u for n, u in usage.items() if _demangled_args(n) and not _demangled_args(n)[5]}
    fused = {n: u for n, u in usage.items() if _demangled_args(n) and _demangled_args(n)[5]}
    # 16 bytes per lane for every width of at least 4 columns: L in {4..64} at VEC = 4, plus the L = 4 builds of D = 2, 3 (VEC = 2)
    # and D = 1 (VEC = 1); the in-launch fused form for VEC = 4
    assert len(plain) == 7 and len(fused) == 5, sorted(usage)
    for n, u in plain.items():
        a = _demangled_args(n)
        assert u["occupancy"] >= 5 and u["vgprs"] + u.get("agprs", 0) <= 96, (a, u)
        # 96 registers + 20 B/lane of scratch (one reload in a depth-1 loop) measured 0.3-3 % FASTER than 94 / no scratch
        # (profiles/r02/ab_occupancy.log); anything beyond that is a regression to look at
        # (the 8-byte-per-lane builds, which serve only embedding widths not divisible by 4, may take one more dword)
        assert u["scratch"] <= (20 if a[2] == 4 else 24), (a, u)
    for n, u in fused.items():
        assert u["occupancy"] >= 4 and u["scratch"] <= 24, (_demangled_args(n), u)
    tiny = {n: u for n, u in usage.items() if "tiny_kernel" in n}
    assert len(tiny) == 7
    for n, u in tiny.items():  # the tiny-task launch exists for its occupancy: eight waves per SIMD, nothing spilled
        assert u["occupancy"] >= 8 and u["scratch"] == 0, (n, u)
    for n, u in usage.items():
        if "fixup_kernel" in n:
            assert u["scratch"] == 0 and u["occupancy"] >= 5, (n, u)
        if "hybrid_window_kernel" in n:  # plan-free kernel (the reference's launch shape): never spills
            assert u["scratch"] == 0 and u["occupancy"] >= 4, (n, u)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_16bit_planned_kernels_keep_four_waves_and_do_not_spill(usage):
    usage = usage["spmm_kernels_h16.hip"]
    planned = {n: u for n, u in usage.items() if _demangled_args(n)}
    assert planned
    for n, u in planned.items():
        a = _demangled_args(n)
        # (the 64-lanes-per-task builds -- 16-bit rows wider than 256 / 512 columns in one pass -- carry one 20-byte reload since
        # the argument block grew in round 3; every build a panel-major launch uses is spill-free)
        assert u["occupancy"] >= 4 and u["scratch"] <= (24 if a[1] == 64 else 0), (a, u)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_fused_tile_kernels_do_not_spill(usage):
    """Row-tile form of the fused operators (fused_rows.hip): the sparse-row tiles are built for five waves per SIMD, the
    dense-window tiles for four; neither spills (a one-launch build of both spilled 140 bytes per lane and lost 25 %)."""
    usage = usage["fused_rows.hip"]
    tiles = {n: u for n, u in usage.items() if "fused_tiles_kernel" in n}
    # (L, DV, waves per workgroup) in {(8, 2, 4), (16, 4, 4), (32, 4, 4), (16, 4, 8: column-chunked)} x output tiles in {1, 2, 4} x {sparse, dense}
    assert len(tiles) == 24
    for n, u in tiles.items():
        m = re.search(r"ELi(\d+)ELi(\d+)EEEvNS_9TilesArgsE", n)
        kind, waves = int(m.group(1)), int(m.group(2))
        if waves == 4:  # whole-row tiles: nothing spilled
            assert u["scratch"] == 0 and u["occupancy"] >= (5 if kind == 1 else 4), (n, u)
        else:           # column-chunked tiles keep the output accumulators across the chunks
            assert u["scratch"] <= (0 if kind == 1 else 32) and u["occupancy"] >= 4, (n, u)
