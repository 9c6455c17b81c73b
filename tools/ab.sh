#!/bin/bash
# same-box A/B of the working-tree library against tools/bin/libbsm_prev.so (tools/build_prev.sh):
# interleaved kbench runs of the configs given (default c2 c2w c3s bem)
CFGS=${@:-"c2 c2w c3s bem"}
for c in $CFGS; do
  for rep in 1 2; do
    echo "== $c prev: $(BSM_LIB=$PWD/tools/bin/libbsm_prev.so python tools/kbench.py $c 300 2>/dev/null | tail -1)"
    echo "== $c new : $(python tools/kbench.py $c 300 2>/dev/null | tail -1)"
  done
done
