"""Build-time guard for the kernels that issue global loads / stores through inline asm with hand-counted `s_waitcnt vmcnt(n)` (csrc/conv3x3.hip rolling-slab kernels,
csrc/stem3p.hip wave-role weight gradient): the compiler does not know those registers are in flight, so a scratch spill or reload between them shifts the counts and
the kernel silently computes on stale data.  The resource report of hipcc (-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU) must show no scratch
and no spilled VGPR for them.  The shifted-window kernels (gemm.hip, conv_s2.hip) only count LDS-DMA groups -- a spill there costs time, not correctness -- and are
reported, not enforced."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARDED = {"conv3x3.hip": ("conv3x3_c64_kernel", "conv3x3_c64_res_kernel", "wgrad3x3_c64"), "stem3p.hip": ("stem3p_wgrad_roles_kernel",)}


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", sorted(GUARDED))
def test_asm_load_kernels_do_not_spill(src, tmp_path):
    from avec_amd import build as b
    cmd = ["hipcc"] + b.FLAGS + ["-c", os.path.join(b.CSRC, src), "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    cur, seen = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs Spill|ScratchSize \[bytes/lane\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
            seen[cur] = {}
        else:
            seen[cur][m.group(1)] = int(m.group(2))
    hit = {k: v for k, v in seen.items() if any(g in k for g in GUARDED[src])}
    assert hit, "none of %r found in the resource report of %s" % (GUARDED[src], src)
    bad = {k: v for k, v in hit.items() if v.get("VGPRs Spill", 0) or v.get("ScratchSize [bytes/lane]", 0)}
    assert not bad, "inline-asm load kernels must not spill (their vmcnt waits are hand-counted): %r" % bad


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_guard_page_allocator_builds_and_exports_the_pluggable_allocator_entry_points(tmp_path):
    """tools/guard/guard_alloc.cpp (test infrastructure of tests/test_gpu_bounds.py): compiles here without a GPU and exports what torch.cuda.memory.CUDAPluggableAllocator binds"""
    import ctypes
    so = str(tmp_path / "libguard_alloc.so")
    r = subprocess.run(["hipcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "guard", "guard_alloc.cpp")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    dll = ctypes.CDLL(so)
    for name in ("guard_malloc", "guard_free", "guard_stats"):
        assert hasattr(dll, name), name
