cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/c15_gputests.log 2>&1; tail -4 gpurun_out/c15_gputests.log
( timeout -k 10 400 python tools/stress_check.py 240 601 2>&1 | tail -3 ) > gpurun_out/r06_stress.txt; cat gpurun_out/r06_stress.txt
( timeout -k 10 300 python tools/pathological_check.py 2>&1 | tail -12 ) > gpurun_out/r06_pathological.txt; tail -4 gpurun_out/r06_pathological.txt
( timeout -k 10 300 python tools/pathological_check.py 40000000 2>&1 | tail -12 ) > gpurun_out/r06_pathological_40M.txt; tail -4 gpurun_out/r06_pathological_40M.txt
