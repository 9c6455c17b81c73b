# same-box A/B of two builds: tools/bin/libbsm_prev.so (BSM_LIB) vs the in-tree library
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/ab_tests.txt 2>&1 || { tail -30 gpurun_out/ab_tests.txt; exit 1; }
tail -2 gpurun_out/ab_tests.txt
P=$PWD/tools/bin/libbsm_prev.so
for rep in 1 2; do
  for cfg in "c2" "c2 500 T" "c3 100" "c5s 60" "bem 60"; do
    echo "prev: $(BSM_LIB=$P python tools/kbench.py $cfg 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
    echo "new : $(python tools/kbench.py $cfg 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
  done
  echo "prev: $(BSM_LIB=$P python tools/bem_real.py 300 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"
  echo "new : $(python tools/bem_real.py 300 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"
  echo "prev: $(BSM_LIB=$P python tools/bem_real.py 300 real 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"
  echo "new : $(python tools/bem_real.py 300 real 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' ')"
done > gpurun_out/ab_result.txt 2>&1
cat gpurun_out/ab_result.txt
