#!/bin/bash
# tools/ab_output_write_pinned.sh -- the same measurements with the process pinned to a few neighbouring CPUs (lock hand-offs stay in one L3)
gcc -O2 -pthread tools/ab_output_write.c -o /tmp/ab_output_write 2>/dev/null || exit 1
lscpu | grep -E "NUMA node|Socket|Thread|Model name|L3" | head -12
for cpus in "" "0-7" "0-15" "0-3"; do
  pre=""; [ -n "$cpus" ] && pre="taskset -c $cpus"
  echo "== cpus: ${cpus:-unpinned}"
  for rep in 1 2; do
    $pre /tmp/ab_output_write /dev/shm/abw.$$ 8 4 0
    $pre /tmp/ab_output_write /dev/shm/abw.$$ 8 1 1
    $pre /tmp/ab_output_write /dev/shm/abw.$$ 8 2 1
    $pre /tmp/ab_output_write /dev/shm/abw.$$ 8 4 6 1
    $pre /tmp/ab_output_write /dev/shm/abw.$$ 8 4 7 1
    $pre /tmp/ab_output_write /dev/shm/abw.$$ 8 4 7 2
    $pre /tmp/ab_output_write /dev/shm/abw.$$ 8 3 8 2
  done
done
