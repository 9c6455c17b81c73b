set -e
mkdir -p gpurun_out
python tools/abbench.py bem_c128 bem_f64 bem_c64 bem_f32 > gpurun_out/r04_ab_base.txt 2>&1
ABB_ACC=gather python tools/abbench.py bem_c128 bem_f64 bem_c64 bem_f32 >> gpurun_out/r04_ab_base.txt 2>&1
python tools/abbench.py c2 c2x20 c3 c4s c5s >> gpurun_out/r04_ab_base.txt 2>&1
cat gpurun_out/r04_ab_base.txt
