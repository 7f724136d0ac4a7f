cd $GRAFT_REPO_ROOT
export BENCH_ARGS="--no-extra"
V=$PWD/tudocomp_amd/lib/variants
tools/ab.sh "TDC_GPU_LIB=$V/peel0.so" "TDC_GPU_LIB=$V/peel1.so" "TDC_GPU_LIB=$V/peel0.so" "TDC_GPU_LIB=$V/peel1.so" "TDC_GPU_LIB=$V/peel0.so" "TDC_GPU_LIB=$V/peel1.so" > gpurun_out/c11_ab.log 2> gpurun_out/c11_ab.err
grep "^==\|^value" gpurun_out/c11_ab.log | cut -c1-200 | paste - -
grep -o "ws_leaf_sort_kernel=[0-9.]*" gpurun_out/c11_ab.log | paste - - - - - -
