#!/bin/bash
# round 5, call 28: the chain kernel with FOUR wavefronts per workgroup (one per SIMD, <= 256 registers: half of every SIMD's register file
# free for other kernels) against eight -- parity, standalone time, step time
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r5c28
make -C deepbedmap_amd/csrc MEASURE=1 -j16 > /dev/null 2>&1 || { echo base build failed; exit 1; }
for nw in 4 8; do
  d=gpurun_out/variants/nw$nw; mkdir -p $d
  cp deepbedmap_amd/csrc/build_measure/*.o $d/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDBM_MEASURE -DTFB_NWAVE=$nw -c deepbedmap_amd/csrc/trunk_fused_bwd.hip -o $d/trunk_fused_bwd.o 2> $d/build.err || { echo "nw$nw: build failed"; tail -5 $d/build.err; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libdbm_measure.so $d/*.o
  rm -f $d/*.o
done
{
echo "== parity with four wavefronts"
DBM_LIB=$PWD/gpurun_out/variants/nw4/libdbm_measure.so timeout 900 python3 -m pytest tests/test_gpu_model.py -x -q -k "trunk or generator_backward or c3lin" 2>&1 | tail -4
for nw in 4 8; do
  echo "== standalone / in-step, NWAVE=$nw"
  DBM_LIB=$PWD/gpurun_out/variants/nw$nw/libdbm_measure.so timeout 300 python3 tools/experiments/step_shapes.py trunk 10 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2 3; do
  for nw in 8 4; do
    echo "[NWAVE=$nw] $(DBM_LIB=$PWD/gpurun_out/variants/nw$nw/libdbm_measure.so timeout 200 python3 tools/step_only.py 100 2>&1 | tail -1)"
  done
done
} > gpurun_out/r5c28/out.txt 2>&1
cat gpurun_out/r5c28/out.txt
