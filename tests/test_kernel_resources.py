"""The sweep kernels' register budget, read from the code objects inside the built library (no GPU needed).

Why this is a test: the gated sweeps run one 1024-thread workgroup per CU (128 registers per lane) and their inner chain is
LDS -> gathers -> sums; a spilled register there is a dependent scratch access per step.  The launchers therefore only
select variants that hipcc allocates without spills (gatmh_sweep_rows, sweep_rows_for) -- a toolchain or code change that
makes one of them spill, or pushes one past 128 registers, must fail here and not show up as a slower epoch."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels():
    lib = os.path.join(ROOT, "dorylus_amd", "libdorylus_hip.so")
    if not (os.path.exists(lib) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("library or llvm tools missing")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=d, check=True, capture_output=True,
                                   text=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", notes):
                dem = m.group(1)
                out[dem] = (int(m.group(2)), int(m.group(3)))
    return out


def _tparams(mangled, stem):
    """template integers of  <stem>ILi32ELi4ELi2ELb1E...  -> [32, 4, 2, 1]"""
    m = re.search(stem + r"I((?:L[ib]\d+E)+)E", mangled)
    return [int(x) for x in re.findall(r"L[ib](\d+)E", m.group(1))] if m else None


def test_selectable_sweep_variants_do_not_spill():
    ks = _kernels()
    seen = {"fwd": 0, "src": 0, "k1s": 0}
    bad = []
    for name, (vgpr, spill) in ks.items():
        p = _tparams(name, "gatmh_forward_sweep_kernel")
        if p:   # GROUP, HL, R, LOADER -- the rule of gatmh_sweep_rows(pass 0)
            group, hl, r, _ = p
            cap = 2 if group == 16 else 4
            if r <= cap:
                seen["fwd"] += 1
                if spill:
                    bad.append((name, vgpr, spill))
            continue
        p = _tparams(name, "gatmh_src_sweep_kernel")
        if p:   # the rule of gatmh_sweep_rows(pass 1)
            group, hl, r, _ = p
            cap = 2 if group == 16 else (4 if hl != 16 else 2)
            if r <= cap:
                seen["src"] += 1
                if spill:
                    bad.append((name, vgpr, spill))
            continue
        p = _tparams(name, "spmm_sweep_kernel")
        if p:   # GROUP, R, UNIT, PAIR, LOADER: the default launches = loader wave on 32 lanes, rows in pairs only with it,
            group, r, unit, pair, loader = p   # no pairs on 16 lanes (they need three slabs: 32-lane tensors)
            default = (group == 32 and loader == 1) or (group == 16 and pair == 0 and r <= 6)   # (sweep_rows_for: six rows at most on 16 lanes)
            if default:
                seen["k1s"] += 1
                if spill:
                    bad.append((name, vgpr, spill))
    assert all(seen.values()), seen
    assert not bad, bad


def test_sweep_kernels_fit_sixteen_waves_per_cu():
    """one 1024-thread workgroup per CU = four waves per SIMD = at most 128 registers per lane"""
    for name, (vgpr, _) in _kernels().items():
        if "sweep_kernel" in name:
            assert vgpr <= 128, (name, vgpr)
