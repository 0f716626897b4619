#!/bin/bash
# rw_rbf_wgrad_kernel (csrc/rbf_wgrad.hip) under the microscope.  Variant libraries are built HERE (CPU container) into
# scratch/ab/:  bash profiles/scripts/rw_ablation.sh build   (VARIANTS="1 2 .." = RW_ABL bit sets: wrong results, timing only;
# PROF=1 adds the cycle-counter build lib_rw_prof.so), then on the GPU:  gpurun -- bash profiles/scripts/rw_ablation.sh run
set -uo pipefail
cd "$(dirname "$0")/../.."
if [ "${1:-build}" = build ]; then
  mkdir -p scratch/ab
  objs=$(ls adsorbdiff_amd/csrc/build/*.o | grep -v rbf_wgrad)
  for v in ${VARIANTS:-}; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DRW_ABL=$v -Iinclude -c adsorbdiff_amd/csrc/rbf_wgrad.hip -o scratch/ab/rw_$v.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/lib_rw_$v.so $objs scratch/ab/rw_$v.o
  done
  if [ "${PROF:-0}" = 1 ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DRW_PROF=1 -Iinclude -c adsorbdiff_amd/csrc/rbf_wgrad.hip -o scratch/ab/rw_prof.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/lib_rw_prof.so $objs scratch/ab/rw_prof.o
  fi
  exit 0
fi
export TMPDIR=/tmp
if [ -f scratch/ab/lib_rw_prof.so ]; then
  ADF_LIB_PATH=$PWD/scratch/ab/lib_rw_prof.so python3 bench.py --mode train --steps 1 --warmup 0 2>&1 | grep rw_prof | sort | uniq -c | sort -rn | head -4
fi
for lib in default $(ls scratch/ab/lib_rw_*.so | grep -v prof); do
  if [ $lib = default ]; then unset ADF_LIB_PATH; else export ADF_LIB_PATH=$PWD/$lib; fi
  rm -rf gpurun_out/rwab
  rocprofv3 --kernel-trace -d gpurun_out/rwab -o t --output-format csv -- python3 bench.py --mode train --steps 1 --warmup 1 > /dev/null 2>&1
  echo "$lib: $(python3 profiles/scripts/trace_by_grid.py gpurun_out/rwab 40 | grep 'rw_rbf_wgrad_kernel<false>' | cut -c1-60)"
done
rm -rf gpurun_out/rwab
