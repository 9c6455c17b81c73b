set -e
L=$PWD/blocksparsematrices.jl_amd
for i in 1 2; do
python tools/abbench.py bem_c128 bem_f64 bem_f32 c3 c5s 2>/dev/null | sed "s/^/L8 /"
BSM_LIB=$L/libbsmrocm_l4.so python tools/abbench.py bem_c128 bem_f64 bem_f32 c3 c5s 2>/dev/null | sed "s/^/L4 /"
done > gpurun_out/r04_l4.txt
cat gpurun_out/r04_l4.txt
