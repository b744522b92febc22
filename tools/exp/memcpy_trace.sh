cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/mc -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --streams 1 --transcript blake2b --no-cpu-baseline --steady-seconds 0 > /dev/null 2>&1
f=$(find /tmp/mc -name "*memory_copy_trace.csv" | head -1)
head -1 $f
tail -40 $f
