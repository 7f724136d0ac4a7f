set -x
cd $GRAFT_REPO_ROOT
export BENCH_ARGS="--no-extra"
V=$PWD/tudocomp_amd/lib/variants
( TDC_GPU_LIB=$V/fill1.so timeout -k 10 500 python -m pytest tests/test_gpu_wsort.py tests/test_gpu_sa_refine.py -x -q 2>&1 | tail -3 ) > gpurun_out/c1_test_fill1.log 2>&1
( TDC_GPU_LIB=$V/fill2.so timeout -k 10 500 python -m pytest tests/test_gpu_wsort.py tests/test_gpu_sa_refine.py -x -q 2>&1 | tail -3 ) > gpurun_out/c1_test_fill2.log 2>&1
tools/ab.sh "X=0" "TDC_GPU_LIB=$V/prof.so" "TDC_GPU_LIB=$V/fill1.so" "TDC_GPU_LIB=$V/fill2.so" "TDC_GPU_LIB=$V/fill1p.so" "TDC_GPU_LIB=$V/exp.so TDC_GPU_EXP_OVERLAP=2" "TDC_GPU_LIB=$V/exp.so TDC_GPU_EXP_OVERLAP=1" "X=1" > gpurun_out/c1_ab.log 2> gpurun_out/c1_ab.err
cat gpurun_out/c1_test_fill1.log gpurun_out/c1_test_fill2.log
grep -v "^ \|kernels" gpurun_out/c1_ab.log | cut -c1-400
grep wl_prof gpurun_out/c1_ab.err | tail -12
