#!/bin/bash
# round 6, GPU pass 20: phase clocks inside k_ntt13 (instrumented build libzkfhe_hip_clk.so = a copy of zk-fhe_amd/ built with wall_clock64 stamps in
# csrc/ntt13.hip and an extern "C" zkfhe_debug_ntt13_clk that copies them out; not part of the tree)
set -u
OUT=gpurun_out/r6v; mkdir -p $OUT
L=zk-fhe_amd/libzkfhe_hip.so
cp $L /tmp/lib_ship.so; cp zk-fhe_amd/libzkfhe_hip_clk.so $L
python tools/exp/ntt13_clock.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ntt13_clock.txt
cp /tmp/lib_ship.so $L
