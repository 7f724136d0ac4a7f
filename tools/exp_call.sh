cd $GRAFT_REPO_ROOT
export BENCH_ARGS="--no-extra"
V=$PWD/tudocomp_amd/lib/variants
tools/ab.sh "TDC_GPU_FLATTEN_REFILL=0" "TDC_GPU_FLATTEN_REFILL=1" "TDC_GPU_FLATTEN_REFILL=0" "TDC_GPU_FLATTEN_REFILL=1" > gpurun_out/c4_ab.log 2> gpurun_out/c4_ab.err
grep -v "^ \|kernels" gpurun_out/c4_ab.log | cut -c1-1200
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/c4_gputests.log 2>&1; tail -5 gpurun_out/c4_gputests.log
