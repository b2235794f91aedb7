for cfg in "DBM_ITER_TAIL=0" "DBM_ITER_TAIL=1" "DBM_ITER_TAIL=1 DBM_PF_PRIORITY=1" "DBM_ITER_TAIL=0 DBM_PF_PRIORITY=1"; do
  for rep in 1 2; do
    echo "$cfg: $(env $cfg python tools/step_only.py 100 2>/dev/null | tail -1)"
  done
done
