#!/bin/bash
# round 6, GPU pass 5: the whole suite (durations), then the round's profiles
set -u
mkdir -p gpurun_out/r6e
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r6e/pytest.log 2>&1; tail -25 gpurun_out/r6e/pytest.log
bash tools/profile.sh 6 > gpurun_out/r6e/profile.log 2>&1; tail -5 gpurun_out/r6e/profile.log
