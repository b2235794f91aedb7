#!/bin/bash
# Stand-alone micro-benchmarks of the weight-gradient kernels (per-phase cycle counts via -DDBM_WG_TIMING).
# usage: tools/wgrad_bench/build.sh && tools/wgrad_bench/trunk 12 ; tools/wgrad_bench/discriminator
set -e
cd "$(dirname "$0")"
sed 's|#include "dbm_internal.h"|#include "../../deepbedmap_amd/csrc/dbm_internal.h"|' ../../deepbedmap_amd/csrc/wgrad.hip > wgrad_abl.hip
for t in trunk discriminator; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -DDBM_WG_TIMING -o $t $t.cpp
done
