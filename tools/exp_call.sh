cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/c5_gputests.log 2>&1; tail -5 gpurun_out/c5_gputests.log
python tools/lz78_check.py 33554432 2>&1 | tail -2
python tools/lz78_check.py 1000000000 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 > gpurun_out/c5_bench.json 2> gpurun_out/c5_bench.err; python3 -c "
import json
j=json.loads(open('gpurun_out/c5_bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['stages_ms'])
for k in j:
    if k.startswith('configs') or k in ('decompress','hbm_resident','stream_matches_golden'): print(k, j[k])
"
