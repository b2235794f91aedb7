#!/bin/bash
# libdbm_measure.so = the product sources with -DDBM_MEASURE: the work-skipping ablation switches (DBM_ABL_SKIP, DBM_NO_WGRAD,
# DBM_TFB_ABL, DBM_CL16_ABL, DBM_ABL_NOPACK) exist only here.  Use it from the tools with
#     DBM_LIB=$PWD/deepbedmap_amd/libdbm_measure.so DBM_NO_WGRAD=1 python tools/phases.py
# bench.py refuses to run with DBM_LIB (or any of the switches) set.
set -e
cd "$(dirname "$0")/.."
make -C deepbedmap_amd/csrc MEASURE=1 -j"$(nproc)"
ls -la deepbedmap_amd/libdbm_measure.so
