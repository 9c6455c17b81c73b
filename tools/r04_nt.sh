set -e
for K in 50 100 150 200 250 300 400; do
for nt in 1 0; do
ABB_K=$K BSM_NT=$nt python tools/abbench.py bem_f32 bem_f64 bem_c128 2>/dev/null | sed "s/^/nt=$nt /"
done; done > gpurun_out/r04_nt_sweep.txt
cat gpurun_out/r04_nt_sweep.txt
