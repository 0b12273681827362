#!/bin/bash
# tools/ab_output_write_threads.sh -- how many writer threads one L3 domain's file write path takes (mode 0 = mmap copies, mode 6 = with
# allocator threads that fallocate ahead); usage: ab_output_write_threads.sh [quick]
gcc -O2 -pthread tools/ab_output_write.c -o /tmp/ab_output_write 2>/dev/null || exit 1
if [ "$1" = "alloc" ]; then
  for cpus in "0-7,128-135"; do
    echo "== cpus: $cpus"
    for t in 4 6 8 10; do
      for a in 1 2 3; do
        for rep in 1 2; do taskset -c $cpus /tmp/ab_output_write /dev/shm/abw.$$ 8 $t 6 $a | sed "s/^/allocators $a | /"; done
      done
    done
  done
  exit 0
fi
for cpus in "0-7" "0-7,128-135"; do
  echo "== cpus: $cpus"
  for t in 2 4 6 8 12; do
    for rep in 1 2; do
      taskset -c $cpus /tmp/ab_output_write /dev/shm/abw.$$ 8 $t 0
      taskset -c $cpus /tmp/ab_output_write /dev/shm/abw.$$ 8 $t 6 1
    done
  done
done
