set -e
L=$PWD/blocksparsematrices.jl_amd
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
BSM_LIB=$L/libbsmrocm_head.so python tools/abbench.py bem_c128 bem_f64 bem_c64 bem_f32 c3 c5s 2>/dev/null | sed "s/^/head /"
python tools/abbench.py bem_c128 bem_f64 bem_c64 bem_f32 c3 c5s 2>/dev/null | sed "s/^/new  /"
done > gpurun_out/r04_p6_ab.txt
cat gpurun_out/r04_p6_ab.txt
