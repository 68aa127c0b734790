#!/bin/bash
# GPU box: phases of the one-workgroup orthogonalisation step (s_memtime stamps, 10 ns ticks) at two shapes
mkdir -p tools/probe/_bin gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Iinclude tools/probe/spec_stamp_probe.hip -o tools/probe/_bin/spec_stamp_probe 2> gpurun_out/spec_stamp_build.log || { tail -5 gpurun_out/spec_stamp_build.log; exit 1; }
for shape in "500 12" "200 32" "200 12"; do for b in spec_stamp_probe; do echo "n p = $shape  $b"; MSM_SPEC_PERSIST=0 timeout -k 10 60 tools/probe/_bin/$b $shape || exit 1; done; done
