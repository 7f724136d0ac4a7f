cd $GRAFT_REPO_ROOT
export BENCH_ARGS="--no-extra"
V=$PWD/tudocomp_amd/lib/variants
( timeout -k 10 600 python -m pytest tests/test_gpu_wsort.py tests/test_gpu_sa_refine.py tests/test_gpu_sort.py -x -q 2>&1 | tail -2 )
tools/ab.sh "TDC_GPU_LIB=$V/swz0.so" "X=swz1" "TDC_GPU_LIB=$V/swz0.so" "X=swz1" "TDC_GPU_LIB=$V/swz0.so" "X=swz1" > gpurun_out/c16_ab.log 2> gpurun_out/c16_ab.err
grep "^==\|^value" gpurun_out/c16_ab.log | cut -c1-200 | paste - -
grep -o "rs_scatter_kernel<u64>=[0-9.]*\|ws_leaf_sort_kernel=[0-9.]*" gpurun_out/c16_ab.log | paste - - 
