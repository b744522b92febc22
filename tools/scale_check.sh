#!/bin/bash
# Scaling check for a node with several MI355X (nothing in this repository has run on one: DESIGN.md section 8).
#
#   bash tools/scale_check.sh [out_dir]          # default out_dir: gpurun_out/scale
#
# Runs plain `python bench.py --gpus N` (the script launches its N ranks itself: one per GPU, RCCL over xGMI) at 1 / 2 / 4 / 8
# ranks -- as many as the node has -- in both multi-GPU modes and both host hashing modes, one JSON line each:
#   batch                independent k = 13 proofs, proof i -> rank i mod W (BASELINE configs[2]; weak scaling, no data-path collective)
#   one-proof-sharded    every proof made by all ranks (BASELINE configs[4]: --config k19; commitments by point range + ncclAllGather
#                        of the partials, quotient by column, evaluations by index; strong scaling)
#   ZKFHE_HASH_MODE      latency (one core per Poseidon sponge) | shared (the eight-lane AVX-512 service: for hosts with few CPUs per GPU)
# and writes <out_dir>/scale.jsonl (one line per run, with "ranks", "mode", "hash_mode" added) plus a table on stdout.  Efficiency is
# for the reader to compute: value(N) / (N * value(1)) for batch, value(N) / value(1) for one-proof-sharded.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/scale}
mkdir -p "$OUT"
cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NGPU; usable CPUs: $(python -c 'import zk_fhe_amd.batch as b; print(b.usable_cpus())')" | tee "$OUT/host.txt"
: > "$OUT/scale.jsonl"
run() {  # ranks mode hash_mode tag extra bench arguments...
  local n=$1 mode=$2 hm=$3 tag=$4; shift 4
  local log="$OUT/${mode}_${tag}_${hm}_n${n}"
  # the PLAIN form: `python bench.py --gpus N` starts its own N ranks under torch.distributed.run (one per GPU, RCCL) and refuses to
  # print a line whose n_gpus is not N; at N = 1 the process group is forced as well, so that the 1-GPU row of the table has paid
  # for the same collectives as the others
  ZKFHE_BENCH_FORCE_DIST=1 ZKFHE_HASH_MODE=$hm python bench.py --gpus "$n" --no-cpu-baseline --no-traffic-pass --mode "$mode" "$@" > "$log.json" 2> "$log.err"
  python - "$log.json" "$n" "$mode" "$hm" >> "$OUT/scale.jsonl" <<'PY'
import json, sys
path, n, mode, hm = sys.argv[1:5]
lines = [l for l in open(path) if l.startswith("{")]
if not lines:
    print(json.dumps({"ranks": int(n), "mode": mode, "hash_mode": hm, "error": "no JSON line (see the .err file)"}))
else:
    d = json.loads(lines[0])
    d.update(ranks=int(n), mode=mode, hash_mode=hm)
    print(json.dumps(d))
PY
}
for n in 1 2 4 8; do
  [ "$n" -le "$NGPU" ] || continue
  for hm in latency shared; do
    run $n batch $hm k13 --steps 96 --warmup 4                                           # 96 proofs per rank, 16 in flight per GPU
  done
  run $n one-proof-sharded latency k19 --config k19 --steps 4 --warmup 1 --transcript blake2b --steady-seconds 0
  run $n one-proof-sharded latency k16 --config k16 --steps 6 --warmup 1 --transcript blake2b --steady-seconds 0
done
python - "$OUT/scale.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
print("%-18s %-8s %5s %12s %10s %s" % ("mode", "hash", "ranks", "proofs/s", "ms/step", "host CPU ms/proof by rank"))
for r in rows:
    if "error" in r:
        print("%-18s %-8s %5d  %s" % (r["mode"], r["hash_mode"], r["ranks"], r["error"]))
        continue
    c = r["config"]
    print("%-18s %-8s %5d %12.2f %10.3f %s  [%s]" % (r["mode"], r["hash_mode"], r["ranks"], r["value"], r["ms_per_step"],
          " ".join("%.1f" % v for v in c["host_cpu_ms_per_proof_by_rank"]), r["metric"]))
PY
