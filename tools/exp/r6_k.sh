#!/bin/bash
# round 6, GPU pass 10: the tree as committed with the per-column tied products as the default -> smoke, the whole GPU suite, the driver's
# command, then the round's profiles again (tools/profile.sh 6) so that profiles/r6_* describe the kernels that ship
set -u
bash tools/exp/r6_final.sh
bash tools/profile.sh 6 "a b c d e f g" > gpurun_out/r6z/profile.log 2>&1
tail -5 gpurun_out/r6z/profile.log
