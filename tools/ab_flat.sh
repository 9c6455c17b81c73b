set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "vbcrs or blocksparse or fuzz or config1 or config2 or config4 or edge or degenerate" > gpurun_out/r02_flat_tests.txt 2>&1 || { tail -30 gpurun_out/r02_flat_tests.txt; exit 1; }
tail -3 gpurun_out/r02_flat_tests.txt
for cfg in c2 c2u c2p; do
  for rep in 1 2; do
    BSM_NOFLAT=1 python tools/kbench.py $cfg 500 2>&1 | tail -1
    python tools/kbench.py $cfg 500 2>&1 | tail -1
    BSM_ALLFLAT=1 python tools/kbench.py $cfg 500 2>&1 | tail -1
  done
done > gpurun_out/r02_flat_ab4.txt 2>&1
cat gpurun_out/r02_flat_ab4.txt
