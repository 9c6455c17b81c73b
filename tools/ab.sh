set -e
P=$PWD/tools/bin/libbsm_prev.so
for rep in 1 2; do
  echo "prev: $(BSM_LIB=$P python tools/bem_real.py 300 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"
  echo "new : $(python tools/bem_real.py 300 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"
  echo "prev: $(BSM_LIB=$P python tools/bem_real.py 300 real 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"
  echo "new : $(python tools/bem_real.py 300 real 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"
  for cfg in "bem 60" "c3 100" "c5s 60"; do
    echo "prev: $(BSM_LIB=$P python tools/kbench.py $cfg 2>&1 | grep -v amdgpu.ids | tail -1)"
    echo "new : $(python tools/kbench.py $cfg 2>&1 | grep -v amdgpu.ids | tail -1)"
  done
done > gpurun_out/ab_result.txt 2>&1
cat gpurun_out/ab_result.txt
