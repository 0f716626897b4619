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
SOURCES = ["api.hip", "gemm.hip", "gemm16.hip", "graph.hip", "message.hip", "nodewise.hip", "stepper.hip", "peaks.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "common.h", PKG.parent / "include" / "adsorbdiff_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        # No SLP vectorisation: with it hipcc (ROCm 7.2) packs adjacent f32 adds/mults of the message
        # kernel's epilogue into v_pk_*_f32 next to the f16 MFMA loop and the kernel then returns
        # run-to-run different sums in lanes 16-31 (scratch/ bisect: deterministic and correct without).
        # Packed f32 VALU math is also slower beside MFMAs (MI355X_MICROARCH.md, filler prices).
        "-fno-slp-vectorize",
        "-o", str(LIB),
    ] + [str(CSRC / s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
