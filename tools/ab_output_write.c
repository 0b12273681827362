// tools/ab_output_write.c -- how fast can N threads put G bytes of text into ONE fresh file on this box?
// (the stream pipeline ends in exactly this: 7.5 GB of SAM per 20 M reads; the per-page kernel cost of the output file decides the run)
//   modes: mmap (shared mapping + memcpy, T threads), pwrite (T threads, disjoint ranges), falloc+mmap, mmap+MADV_HUGEPAGE, reuse (no O_TRUNC)
// build: gcc -O2 -pthread tools/ab_output_write.c -o /tmp/ab_output_write
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static char *src;
static size_t total, piece = 1 << 20;
static int fd, T, mode;
static char *map;
static size_t next_piece;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;

// modes 6..8: A allocator threads run ahead of the copiers: fallocate a 64 MB step, then (7, 8) MADV_POPULATE_WRITE it
static volatile size_t alloc_done[8];
static int A = 1;
static size_t step = 64 << 20;
static void *allocator(void *arg)
{
	int me = (int)(long)arg;
	for (size_t at = (size_t)me * step; at < total; at += (size_t)A * step) {
		size_t n = at + step <= total ? step : total - at;
		if (mode != 8 && fallocate(fd, 0, (off_t)at, (off_t)n) != 0) { perror("fallocate"); exit(1); }
#ifdef MADV_POPULATE_WRITE
		if (mode >= 7 && madvise(map + at, n, MADV_POPULATE_WRITE) != 0) { perror("madvise"); exit(1); }
#endif
		__atomic_store_n(&alloc_done[me], at + n, __ATOMIC_RELEASE);
	}
	__atomic_store_n(&alloc_done[me], (size_t)-1, __ATOMIC_RELEASE);
	return NULL;
}
static int allocated(size_t at, size_t n)
{
	// the step holding [at, at + n) belongs to allocator (at / step) % A
	size_t s0 = at / step, s1 = (at + n - 1) / step;
	for (size_t s = s0; s <= s1; ++s) {
		size_t d = __atomic_load_n(&alloc_done[s % (size_t)A], __ATOMIC_ACQUIRE);
		if (d != (size_t)-1 && d < (s + 1) * step && d < total) return 0;
	}
	return 1;
}

static void *worker(void *arg)
{
	(void)arg;
	for (;;) {
		pthread_mutex_lock(&mu);
		size_t at = next_piece;
		next_piece += piece;
		pthread_mutex_unlock(&mu);
		if (at >= total) return NULL;
		if (mode >= 6) while (!allocated(at, at + piece <= total ? piece : total - at)) usleep(50);
		size_t n = at + piece <= total ? piece : total - at;
		if (mode == 1) { if (pwrite(fd, src + (at % (64 << 20)), n, (off_t)at) != (ssize_t)n) { perror("pwrite"); exit(1); } }
		else memcpy(map + at, src + (at % (64 << 20)), n);
	}
}

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: %s FILE GBYTES THREADS mode(0 mmap,1 pwrite,2 falloc+mmap,3 mmap+hugepage,4 reuse+mmap,5 mmap+populate_write,6 allocator threads: fallocate ahead,7 ... + populate,8 populate only) [allocators]\n", argv[0]); return 2; }
	const char *path = argv[1];
	total = (size_t)(atof(argv[2]) * (1 << 30));
	T = atoi(argv[3]);
	mode = atoi(argv[4]);
	src = malloc(64 << 20);
	memset(src, 'A', 64 << 20);
	struct rusage r0, r1;
	if (mode == 4) {                       // a file that exists already with its pages: write it once first
		fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
		ftruncate(fd, (off_t)total);
		map = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		for (size_t a = 0; a < total; a += 4096) map[a] = 1;
		munmap(map, total);
		close(fd);
	}
	getrusage(RUSAGE_SELF, &r0);
	double t0 = now();
	fd = open(path, O_RDWR | O_CREAT | (mode == 4 ? 0 : O_TRUNC), 0644);
	if (fd < 0) { perror("open"); return 1; }
	if (mode != 1) {
		if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
		if (mode == 2 && posix_fallocate(fd, 0, (off_t)total) != 0) { perror("fallocate"); return 1; }
		map = mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		if (map == MAP_FAILED) { perror("mmap"); return 1; }
		if (mode == 3) madvise(map, total, MADV_HUGEPAGE);
#ifdef MADV_POPULATE_WRITE
		if (mode == 5) madvise(map, total, MADV_POPULATE_WRITE);
#endif
	}
	double t1 = now();
	pthread_t th[64], ath[8];
	if (argc > 5) A = atoi(argv[5]);
	if (mode >= 6) for (long i = 0; i < A; ++i) pthread_create(&ath[i], NULL, allocator, (void *)i);
	for (int i = 0; i < T; ++i) pthread_create(&th[i], NULL, worker, NULL);
	for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
	if (mode >= 6) for (int i = 0; i < A; ++i) pthread_join(ath[i], NULL);
	double t2 = now();
	if (mode != 1) munmap(map, total);
	close(fd);
	double t3 = now();
	getrusage(RUSAGE_SELF, &r1);
	double us = (r1.ru_utime.tv_sec - r0.ru_utime.tv_sec) + 1e-6 * (r1.ru_utime.tv_usec - r0.ru_utime.tv_usec);
	double sy = (r1.ru_stime.tv_sec - r0.ru_stime.tv_sec) + 1e-6 * (r1.ru_stime.tv_usec - r0.ru_stime.tv_usec);
	printf("mode %d threads %2d: setup %.3f s, copy %.3f s (%.2f GB/s), close %.3f s | user %.2f sys %.2f | %.2f us sys per 4 KB page\n", mode, T, t1 - t0, t2 - t1, total / (t2 - t1) / 1e9,
	       t3 - t2, us, sy, sy * 1e6 / (total / 4096.0));
	unlink(path);
	return 0;
}
