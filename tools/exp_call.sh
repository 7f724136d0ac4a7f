cd $GRAFT_REPO_ROOT
( timeout -k 10 500 python tools/pathological_check.py 40000000 2>&1 | tail -14 ) > gpurun_out/r06_pathological_40M.txt; tail -6 gpurun_out/r06_pathological_40M.txt
bash tools/profile_round.sh r6b > /dev/null 2>&1; ls gpurun_out/prof_r6b | head
