#!/bin/bash
# builds the committed (HEAD) state of the library as tools/bin/libbsm_prev.so for same-box A/B runs
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/prev && mkdir -p /tmp/prev $R/tools/bin
git -C $R archive HEAD blocksparsematrices.jl_amd/csrc include | tar -x -C /tmp/prev
make -s -C /tmp/prev/blocksparsematrices.jl_amd/csrc OUT=$R/tools/bin/libbsm_prev.so
ls -la $R/tools/bin/libbsm_prev.so
