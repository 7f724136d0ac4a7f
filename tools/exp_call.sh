cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_sa_refine.py tests/test_gpu_wsort.py tests/test_gpu_sort.py -x -q 2>&1 | tail -15
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
import tudocomp_amd as T
n = 1_000_000_000
d = T.gen_dna(n, 7)
text = np.concatenate([d, np.zeros(1, dtype=np.uint8)])
for opts in ({}, {"sa_stars": 0}):
    with T.Context(0, options=opts) as ctx:
        for it in range(2):
            out, st = ctx.lcpcomp_compress(text, 2, 1, T.CODER_ARITH)
        print(opts, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("ms_") or k in ("sa_rounds", "sa_star_chains", "out_len")}, flush=True)
PY
