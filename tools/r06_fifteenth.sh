#!/bin/bash
# round 6, fifteenth GPU call: the corrected gz test; where configs[3] (-pacbio) spends its time on the round's code (kernel trace at 400 k reads, stage seconds at 2 M)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_hostpath_gpu.py -q -m gpu > gpurun_out/r06o_gz_tests.log 2>&1
tail -3 gpurun_out/r06o_gz_tests.log
bash tools/profile_pacbio.sh 400000 > gpurun_out/r06o_profile_pacbio.log 2>&1
cp gpurun_out/prof_pb/*/pb_kernel_stats.csv gpurun_out/r06o_pacbio_kernel_stats.csv 2>/dev/null || find gpurun_out/prof_pb -name "pb_kernel_stats.csv" -exec cp {} gpurun_out/r06o_pacbio_kernel_stats.csv \;
rm -rf gpurun_out/prof_pb
CHUNKS="4096:4096" NO_HOST_LONG=1 CHECK=0 bash tools/ab_long.sh 2000000 > gpurun_out/r06o_ab_long_2m.log 2>&1
cat gpurun_out/r06o_profile_pacbio.log | tail -25; cat gpurun_out/r06o_ab_long_2m.log | tail -12
