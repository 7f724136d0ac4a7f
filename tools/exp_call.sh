cd $GRAFT_REPO_ROOT
export BENCH_ARGS="--no-extra"
V=$PWD/tudocomp_amd/lib/variants
tools/ab.sh "X=new" "TDC_GPU_LIB=$V/prev.so" "X=new" "TDC_GPU_LIB=$V/prev.so" > gpurun_out/c9_ab.log 2> gpurun_out/c9_ab.err
grep -v "^ \|kernels" gpurun_out/c9_ab.log | cut -c1-300
grep -o "rs_scatter_kernel<u64>=[0-9.]*\|rs_count_kernel=[0-9.]*\|ws_leaf_sort_kernel=[0-9.]*" gpurun_out/c9_ab.log | paste - - -
