import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
import zk_fhe_amd as zk, zk_fhe_amd.batch as batch
from zk_fhe_amd import inputs
ctx = zk.Context(0)
cfgj = json.load(open("tests/golden/bfv/bfv_config.json"))
srs = zk.Srs(ctx, 13)
pk = zk.BfvProvingKey(ctx, srs, json.dumps(inputs.empty(1024)), (1024, 536870909, 7, 19), zk.BfvConfig.from_pinning(cfgj), replay=True)
texts = [json.dumps(inputs.generate(1024, 536870909, 7, 19, seed=i)).encode() for i in range(4)]
ctxs = [ctx] + [zk.Context(0) for _ in range(11)]
fn = lambda c, j: pk.prove(texts[j % 4], b"s%d" % j, ctx=c)
batch.run_concurrent(list(range(24)), ctxs, fn)
t0, c0 = time.perf_counter(), time.process_time()
batch.run_concurrent(list(range(192)), ctxs, fn)
t1, c1 = time.perf_counter(), time.process_time()
print("proofs/s %.1f   CPU ms per proof %.1f   busy cores %.1f" % (192 / (t1 - t0), 1e3 * (c1 - c0) / 192, (c1 - c0) / (t1 - t0)))
