// tools/probe_tmpfs_pages.cpp -- what a fresh page of a tmpfs file costs on this machine, by the way it is first written: the bound of the
// FASTQ -> SAM step is the host filling ~9 M fresh 4 KB pages of /dev/shm per 100 M reads.  One thread, 1 GiB per variant:
//   store      mmap(MAP_SHARED) + memcpy               (a write fault per page: what the writers' mapping threads do)
//   huge       the same after madvise(MADV_HUGEPAGE)   (2 MB pages, if the tmpfs / sysfs settings allow them)
//   pwrite     pwrite() in 1 MB pieces                 (the writers' pwrite thread)
//   fallocate  fallocate() the range, then store
//   populate   madvise(MADV_POPULATE_WRITE), then store
// and the THP / shmem settings the machine runs with.  Build: g++ -O2 -o probe_tmpfs_pages tools/probe_tmpfs_pages.cpp
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void show(const char *path)
{
	FILE *f = fopen(path, "r");
	if (!f) { printf("%s: (absent)\n", path); return; }
	char buf[512];
	if (fgets(buf, sizeof(buf), f)) { buf[strcspn(buf, "\n")] = 0; printf("%s: %s\n", path, buf); }
	fclose(f);
}

static void meminfo(const char *tag)
{
	FILE *f = fopen("/proc/meminfo", "r");
	char buf[256];
	printf("  [%s]", tag);
	while (f && fgets(buf, sizeof(buf), f))
		if (!strncmp(buf, "ShmemHugePages", 14) || !strncmp(buf, "ShmemPmdMapped", 14)) { buf[strcspn(buf, "\n")] = 0; printf(" %s;", buf); }
	if (f) fclose(f);
	printf("\n");
}

int main(int argc, char **argv)
{
	const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
	const size_t bytes = (size_t)(argc > 2 ? atol(argv[2]) : 1024) << 20;
	show("/sys/kernel/mm/transparent_hugepage/enabled");
	show("/sys/kernel/mm/transparent_hugepage/shmem_enabled");
	show("/sys/kernel/mm/transparent_hugepage/hugepages-2048kB/shmem_enabled");
	show("/sys/kernel/mm/transparent_hugepage/hugepages-64kB/shmem_enabled");
	show("/proc/sys/kernel/osrelease");
	{
		FILE *f = fopen("/proc/mounts", "r");
		char buf[512];
		while (f && fgets(buf, sizeof(buf), f))
			if (strstr(buf, dir.c_str())) printf("mount: %s", buf);
		if (f) fclose(f);
	}
	char *src = (char *)malloc(bytes);
	memset(src, 'x', bytes);
	const char *names[] = {"store", "huge", "pwrite", "fallocate", "populate"};
	for (int round = 0; round < 3; ++round)          // (round 0 warms the machine up: its figures are printed but not to be quoted)
	for (int v = 0; v < 5; ++v) {
		const std::string path = dir + "/probe_tmpfs_pages." + std::to_string((long)getpid());
		int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
		if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { perror("open/ftruncate"); return 1; }
		double t0 = now(), t_prep = 0;
		if (v == 2) {
			for (size_t at = 0; at < bytes; at += 1 << 20)
				if (pwrite(fd, src + at, 1 << 20, (off_t)at) != (1 << 20)) { perror("pwrite"); break; }
		} else {
			char *m = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			if (m == MAP_FAILED) { perror("mmap"); return 1; }
			int rc = 0;
			if (v == 1) rc = madvise(m, bytes, MADV_HUGEPAGE);
			if (v == 3) rc = fallocate(fd, 0, 0, (off_t)bytes);
#ifdef MADV_POPULATE_WRITE
			if (v == 4) rc = madvise(m, bytes, MADV_POPULATE_WRITE);
#endif
			if (rc != 0) printf("  (%s: preparation failed: %s)\n", names[v], strerror(errno));
			t_prep = now() - t0;
			memcpy(m, src, bytes);
			if (v == 1) meminfo("after the huge variant");
			munmap(m, bytes);
		}
		double dt = now() - t0;
		printf("round %d %-9s %.3f s for %zu MiB = %.2f GB/s, %.3f us per 4 KB page (preparation %.3f s)\n", round, names[v], dt, bytes >> 20, (double)bytes / dt / 1e9,
		       dt / (double)(bytes / 4096) * 1e6, t_prep);
		close(fd);
		unlink(path.c_str());
	}
	return 0;
}
