#!/bin/bash
# round 6, twenty-second GPU call: a probe for the next round -- can the copy engine write the SAM text straight into the output file's pages?
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/probes/direct_output.cpp -o /tmp/direct_output -lpthread > gpurun_out/r06v_direct_output.log 2>&1
timeout 420 /tmp/direct_output 16 7 >> gpurun_out/r06v_direct_output.log 2>&1
echo "exit $?" >> gpurun_out/r06v_direct_output.log
rm -f /dev/shm/kart_probe_a /dev/shm/kart_probe_b /dev/shm/kart_probe_c /dev/shm/kart_probe_d
cat gpurun_out/r06v_direct_output.log
