import sys, time, random
sys.path.insert(0, '.')
import numpy as np, tudocomp_amd as T
from oracle import oracle as O
rng = random.Random(1)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
blk = bytes(rng.randrange(97, 123) for _ in range(R))
data = blk + b"#" + blk + T.gen_english(300_000, 5).tobytes()
text = O.escape(data)
with T.Context(0) as ctx:
    t0 = time.time(); out, st = ctx.lcpcomp_compress(text, 5, 1); t1 = time.time()
    print("GPU %.2f s  levels %d small %d maxlcp %d factors %d" % (t1 - t0, st["levels"], st["small_levels"], st["maxlcp"], st["factors"]))
t0 = time.time(); want, _ = O.lcpcomp_huff_compress(text, 5, 1); t1 = time.time()
print("oracle %.2f s  equal %s" % (t1 - t0, out == want))
