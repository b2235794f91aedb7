#!/bin/bash
# round 5, call 29: four-wavefront chain with a raised wavefront priority (does the issue arbiter protect it from co-resident kernels?)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c29
make -C deepbedmap_amd/csrc MEASURE=1 -j16 > /dev/null 2>&1 || { echo base build failed; exit 1; }
build() { # name flags
  d=gpurun_out/variants/$1; mkdir -p $d
  cp deepbedmap_amd/csrc/build_measure/*.o $d/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDBM_MEASURE $2 -c deepbedmap_amd/csrc/trunk_fused_bwd.hip -o $d/trunk_fused_bwd.o 2> $d/build.err || { echo "$1: build failed"; tail -5 $d/build.err; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libdbm_measure.so $d/*.o
  rm -f $d/*.o
}
build nw4p3 "-DTFB_NWAVE=4 -DTFB_PRIO_ALL=3"
build nw8p3 "-DTFB_NWAVE=8 -DTFB_PRIO_ALL=3"
build nw8 "-DTFB_NWAVE=8"
{
for rep in 1 2; do
  for v in nw8 nw8p3 nw4p3; do
    echo "[$v] $(DBM_LIB=$PWD/gpurun_out/variants/$v/libdbm_measure.so timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done
DBM_LIB=$PWD/gpurun_out/variants/nw4p3/libdbm_measure.so timeout 300 python3 tools/experiments/step_shapes.py trunk_bwd 10 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r5c29/out.txt 2>&1
cat gpurun_out/r5c29/out.txt
