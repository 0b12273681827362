#!/bin/bash
# tools/ab_output_register.sh [GB] -- copy engine straight into the mapped output file vs. the product's D2H + CPU copy (see the .cpp)
set -u
G=${1:-16}
B=tools/_build/ab_output_register
[ -x $B ] || hipcc -O2 --offload-arch=gfx950 tools/ab_output_register.cpp -o $B -lpthread
D=/dev/shm
grep -E "MemTotal|MemAvailable|Shmem:" /proc/meminfo
cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory.current 2>/dev/null
df -h /dev/shm | tail -1
for spec in "1 1024 6" "1 256 6" "0 1024 1" "0 1024 2" "0 1024 4" "0 256 4" "0 256 8" "0 64 8" "2 1024 4" "2 256 8" "4 1024 4" "4 256 8" "3 1024 4"; do
	set -- $spec
	timeout 300 $B $D $G $2 $3 $1 || echo "mode $1 window $2 threads $3: failed ($?)"
done
