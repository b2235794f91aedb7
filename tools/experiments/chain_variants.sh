#!/bin/bash
# builds libdbm_measure.so variants of the chain kernel (-D flags) into gpurun_out/variants/<name>/ and times the trunk launches
# usage: chain_variants.sh "name:-DFLAG=.. -DFLAG2" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
make -C deepbedmap_amd/csrc MEASURE=1 -j16 > /dev/null 2>&1 || { echo base build failed; exit 1; }
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  d=gpurun_out/variants/$name; mkdir -p $d
  cp deepbedmap_amd/csrc/build_measure/*.o $d/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDBM_MEASURE $flags -c deepbedmap_amd/csrc/${SRC:-trunk_fused_bwd}.hip -o $d/${SRC:-trunk_fused_bwd}.o 2> $d/build.err || { echo "$name: build failed"; tail -5 $d/build.err; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libdbm_measure.so $d/*.o
  echo "== $name ($flags)"
  DBM_LIB=$PWD/$d/libdbm_measure.so ${ENVV} python tools/experiments/step_shapes.py trunk 10 2>&1 | grep -v amdgpu.ids
  rm -f $d/*.o
done
