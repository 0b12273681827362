#!/bin/bash
# tools/ab_output_register.sh [GB] [pinned] -- copy engine straight into the mapped output file vs. the product's D2H + CPU copy (see the .cpp)
#   "pinned": every thread on the CPUs of ONE L3 domain (cpu0's), as the product keeps its I/O threads
set -u
G=${1:-16}
B=tools/_build/ab_output_register
[ -x $B ] || hipcc -O2 --offload-arch=gfx950 tools/ab_output_register.cpp -o $B -lpthread
D=/dev/shm
PIN=""
if [ "${2:-}" = "pinned" ]; then
	L3=$(cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list)
	PIN="taskset -c $L3"
	echo "threads kept on the L3 domain of cpu0: $L3"
fi
grep -E "MemTotal|MemAvailable|Shmem:" /proc/meminfo
for spec in ${SPECS:-"1 1024 6" "1 1024 4" "1 1024 8" "0 1024 1" "0 1024 2" "0 1024 3" "0 1024 4" "0 256 4" "0 256 6" "3 1024 2" "3 1024 4"}; do
	set -- $spec
	timeout 300 $PIN $B $D $G $2 $3 $1 || echo "mode $1 window $2 threads $3: failed ($?)"
done
