"""Build libadsorbdiff_hip.so (gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container (``__graft_entry__.build()``).  The .so is git-ignored but travels to
the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libadsorbdiff_hip.so"
SOURCES = ["api.hip", "gemm.hip", "gemm16.hip", "mlp16.hip", "graph.hip", "message.hip", "message_bwd.hip", "rbf_wgrad.hip", "nodewise.hip", "stepper.hip", "peaks.hip", "collect.hip", "frames.hip", "train.hip", "incremental.hip", "eqv2_kernels.hip", "eqv2_gemm16.hip", "eqv2_api.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "adsorbdiff_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # No SLP vectorisation: with it hipcc (ROCm 7.2) packs adjacent f32 adds/mults of the message
    # kernel's epilogue into v_pk_*_f32 next to the f16 MFMA loop and the kernel then returned
    # run-to-run different sums in lanes 16-31 (bisected: deterministic and correct without).
    # Packed f32 VALU math is also slower beside MFMAs (MI355X_MICROARCH.md, filler prices).
    "-fno-slp-vectorize",
]
EXTRA_FLAGS = {}  # per-file additions to FLAGS
OBJ_DIR = CSRC / "build"  # git-ignored


def _compile_one(args):
    cc, src, obj = args
    res = subprocess.run([cc, *FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)],
                         capture_output=True, text=True)
    return src.name, res.returncode, res.stdout + res.stderr


def build(force: bool = False, verbose: bool = False) -> Path:
    """One object per translation unit (compiled in parallel, only when stale), then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    cc = _hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    hdr_t = max([p.stat().st_mtime for p in CSRC.glob("*.h")] + [(PKG.parent / "include" / "adsorbdiff_hip.h").stat().st_mtime,
                                                                  Path(__file__).stat().st_mtime])
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = CSRC / s, OBJ_DIR / (s + ".o")
        objs.append(obj)
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_t):
            jobs.append((cc, src, obj))
    if verbose:
        print("compiling:", [j[1].name for j in jobs])
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for name, rc, out in ex.map(_compile_one, jobs):
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {name}:\n{out}")
    res = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs],
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
