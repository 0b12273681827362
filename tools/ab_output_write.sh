#!/bin/bash
# tools/ab_output_write.sh -- the box's output-file cost, alone (see ab_output_write.c)
gcc -O2 -pthread tools/ab_output_write.c -o /tmp/ab_output_write || exit 1
uname -r; nproc; cat /sys/kernel/mm/transparent_hugepage/shmem_enabled /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
grep -E " /dev/shm | /tmp | / " /proc/mounts
cat /sys/fs/cgroup/cpu.max 2>/dev/null
if [ "$1" = "alloc" ]; then
  for rep in 1 2; do
  for m in 6 7 8; do for a in 1 2 3; do for t in 2 4 6; do /tmp/ab_output_write /dev/shm/abw.$$ 8 $t $m $a; done; done; done
  /tmp/ab_output_write /dev/shm/abw.$$ 8 1 1; /tmp/ab_output_write /dev/shm/abw.$$ 8 4 0; /tmp/ab_output_write /dev/shm/abw.$$ 8 4 2
  done
  exit 0
fi
for m in 0 1 2 3 4 5; do for t in 1 2 4 8 16; do /tmp/ab_output_write /dev/shm/abw.$$ 4 $t $m; done; done
for t in 1 4; do /tmp/ab_output_write /tmp/abw.$$ 4 $t 0; /tmp/ab_output_write /tmp/abw.$$ 4 $t 1; done
