cd $GRAFT_REPO_ROOT
bash tools/build_measure.sh > /dev/null 2>&1 || echo build failed
export DBM_LIB=$PWD/deepbedmap_amd/libdbm_measure.so
for a in 0 1 2 3 4 7; do DBM_TFB_ABL=$a python tools/experiments/step_shapes.py trunk 10 2>&1 | grep -v amdgpu.ids; done
