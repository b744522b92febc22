import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zk_fhe_amd as zk
from zk_fhe_amd import inputs
ctx = zk.Context(0)
N, k, Q = 16384, 19, (1 << 60) - 93
text = json.dumps(inputs.generate(N, Q, 7, 19, seed=3))
srs = zk.Srs(ctx, k)
pk = zk.BfvProvingKey(ctx, srs, text, (N, Q, 7, 19), zk.BfvConfig(k, 1, 62, 17, 2, 109))
for r in range(2):
    proof, inst, tm = pk.prove(text, b"s%d" % r)
    print(tm)
