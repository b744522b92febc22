#!/bin/bash
# round 6, GPU pass 6: the gate's rotations from neighbouring lanes (probe), soak runs with announced proofs
set -u
OUT=gpurun_out/r6f; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -I zk-fhe_amd/csrc tools/exp/quotient_gate_probe.hip -o /tmp/qprobe 2> $OUT/qprobe_build.log
for k in 13 16 19; do /tmp/qprobe $k; done > $OUT/quotient_probe.txt 2>&1; cat $OUT/quotient_probe.txt
python tools/soak.py --proofs 6000 --streams 12 --transcript poseidon --announce 3 > $OUT/soak_k13_poseidon_announce.json 2> $OUT/soak1.err; cat $OUT/soak_k13_poseidon_announce.json
python tools/soak.py --proofs 3000 --streams 16 --transcript poseidon --hash-mode shared --gate 4 --announce 1 > $OUT/soak_k13_shared_announce.json 2> $OUT/soak2.err; cat $OUT/soak_k13_shared_announce.json
python tools/soak.py --proofs 3000 --streams 8 --transcript blake2b --announce 2 > $OUT/soak_k13_blake2b_announce.json 2> $OUT/soak3.err; cat $OUT/soak_k13_blake2b_announce.json
python tools/soak.py --proofs 300 --streams 3 --transcript poseidon --config k16 --inputs 6 --announce 3 > $OUT/soak_k16_announce.json 2> $OUT/soak4.err; cat $OUT/soak_k16_announce.json
tail -3 $OUT/soak*.err
