"""Build libavec_hip.so (gfx950) from avec_amd/csrc/*.hip with hipcc, in-tree.

    python -m avec_amd.build [--force] [--report]

hipcc cross-compiles without a GPU; the .so ships to the GPU box with the repo snapshot."""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "libavec_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INC, "-I" + CSRC, "-Wno-unused-value"]
# kernel arguments preloaded into SGPRs by the command processor (gfx940+): the leading scalar / pointer arguments of a kernel (up to 16 dwords) are in registers when
# the first wave starts instead of behind an s_load round trip to the kernel-argument buffer (which misses every cache in a step that moves 50 GB between two launches
# of the same kernel).  Kernels that take one struct by value (the product / convolution launchers) are unaffected.  AVEC_KERNARG_PRELOAD=0 builds without it.
if os.environ.get("AVEC_KERNARG_PRELOAD", "16") != "0":
    FLAGS += ["-mllvm", "-amdgpu-kernarg-preload-count=" + os.environ.get("AVEC_KERNARG_PRELOAD", "16")]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INC, "avec_hip.h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, report):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _deps_mtime()) and not report:
        return obj, ""
    cmd = ["hipcc"] + FLAGS + ["-c", src, "-o", obj]
    if report:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    return obj, r.stderr


def summarize(stderr):
    rows, cur = [], {}
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k] = v
    out = []
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()[:110]
        out.append("%-110s vgpr=%-4s agpr=%-3s occ=%s spillV=%s scratch=%s" % (
            name, r.get("VGPRs"), r.get("AGPRs"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill"),
            r.get("ScratchSize [bytes/lane]")))
    return "\n".join(out)


def build(force=False, report=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(lambda s: _compile(s, report), srcs))
    objs = [o for o, _ in res]
    if report:
        for (o, err), s in zip(res, srcs):
            print("==", os.path.basename(s))
            print(summarize(err))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, report="--report" in sys.argv)
