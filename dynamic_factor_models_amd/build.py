"""In-tree build of libdfmhip.so (hipcc, gfx950 only).  `python -m dynamic_factor_models_amd.build`."""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libdfmhip.so")
SOURCES = ["collapse.hip", "collapse_miss.hip", "collapse_dma.hip", "collapse_mfma.hip", "collapse_wide.hip", "collapse_wide2.hip", "recursion.hip", "recursion_wave.hip", "recursion_pair.hip", "recursion_chunk.hip", "recursion_tile.hip", "recursion_mbf16.hip", "recursion_comp.hip", "fastpath.hip", "scan_mfma32.hip", "em_update_grid.hip", "pass_fused.hip", "mstep.hip", "mstep_mfma.hip", "mstep_wide.hip", "mstep_ar.hip", "mstep_obs.hip", "mstep_miss.hip", "pca.hip", "gram_xx_wide.hip", "als.hip", "boot.hip", "breaks.hip", "synth.hip", "capi.hip", "multi.hip", "probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# per-file flags.  recursion_tile.hip: MFMA accumulators in VGPRs (gfx950 takes either file for srcC / vDst).  The default
# allocation put the 16 x 16 tiles in AGPRs and bracketed every v_mfma with 8 + 8 v_accvgpr moves -- 16 of 129 instructions per
# block pivot of a chain that issues one instruction per ~6 cycles (948 -> 132 v_accvgpr in the kernel).
FILE_FLAGS = {"recursion_tile.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libdfmhip.so (ROCm toolchain required)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """diag=True: the diagnostics library lib/libdfmhip_diag.so (-DDFM_DIAG: ablation switches, phase stamps, *_OLD kernels;
    objects under lib/diag/).  The default library is built without it and ignores those switches."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "diag") if diag else LIBDIR
    os.makedirs(objdir, exist_ok=True)
    so = os.path.join(LIBDIR, "libdfmhip_diag.so") if diag else SO
    flags = FLAGS + (["-DDFM_DIAG"] if diag else [])
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "dfm_hip.h"))
    hipcc = _hipcc()
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc, *flags, *FILE_FLAGS.get(s, []), "-c", src, "-o", obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if verbose or res.returncode:
                    sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
                if res.returncode:
                    raise RuntimeError(f"hipcc failed: {' '.join(cmd)}")
    if force or jobs or _stale(so, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs, "-ldl", "-lpthread"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("link of libdfmhip.so failed")
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
