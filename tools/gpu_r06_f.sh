#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/bands_rows16 tools/ubench/bands_rows16.hip 2>/dev/null && for i in 1 2 3; do /tmp/bands_rows16; done | tee $O/ubench_bands_rows16.txt
