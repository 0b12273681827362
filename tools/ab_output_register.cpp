// tools/ab_output_register.cpp -- can the copy engine write SAM text straight into the mapped output file?
// (VERDICT r3 item 5 / DESIGN 8-1: the step's bound is the CPU copy of 37.8 GB of text per 100 M reads into fresh tmpfs pages.)
//   G GB of device memory -> ONE fresh file in <dir>, window by window (W MB), T threads, each window:
//     mode 0  mmap + hipHostRegister + hipMemcpyAsync(D2H into the window) + sync + hipHostUnregister + munmap
//     mode 1  the product's present form: D2H into a page-locked buffer, then memcpy into the mapping (page faults allocate the file's pages)
//     mode 2  as 0 with MADV_POPULATE_WRITE in front of the registration (pages allocated in one call instead of inside the pin)
//     mode 3  as 0 but the windows stay registered (no unregister / munmap until the end): what a run that keeps its windows pays
//     mode 4  fallocate the window first (pages allocated without zero-fill faults through the mapping), then as 0
//   prints GB/s of the whole job and the thread-seconds of every phase.
// build: hipcc -O2 --offload-arch=gfx950 tools/ab_output_register.cpp -o /tmp/ab_output_register -lpthread
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <thread>
#include <vector>

static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv)
{
	if (argc < 6) { fprintf(stderr, "usage: %s <dir> <total_GB> <window_MB> <threads> <mode>\n", argv[0]); return 2; }
	const char *dir = argv[1];
	const size_t total = (size_t)atoll(argv[2]) << 30, W = (size_t)atoll(argv[3]) << 20;
	const int T = atoi(argv[4]), mode = atoi(argv[5]);
	const size_t nwin = total / W;
	char path[512];
	snprintf(path, sizeof(path), "%s/ab_output_register_%d.bin", dir, (int)getpid());
	int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
	if (fd < 0) { perror("open"); return 1; }
	if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
	CK(hipSetDevice(0));
	char *dsrc = nullptr;
	CK(hipMalloc((void **)&dsrc, W));
	CK(hipMemset(dsrc, 'S', W));
	CK(hipDeviceSynchronize());
	std::atomic<size_t> next{0};
	std::atomic<long long> ns_map{0}, ns_pop{0}, ns_reg{0}, ns_copy{0}, ns_unreg{0}, ns_cpu{0};
	std::vector<std::pair<void *, size_t>> kept(nwin);
	const double t0 = now();
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t)
		th.emplace_back([&, t]() {
			CK(hipSetDevice(0));
			hipStream_t st;
			CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
			char *pinned = nullptr;
			if (mode == 1) CK(hipHostMalloc((void **)&pinned, W, hipHostMallocDefault));
			for (;;) {
				size_t w = next.fetch_add(1);
				if (w >= nwin) break;
				double a = now();
				if (mode == 4 && fallocate(fd, 0, (off_t)(w * W), (off_t)W) != 0) { perror("fallocate"); exit(1); }
				void *m = mmap(nullptr, W, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)(w * W));
				if (m == MAP_FAILED) { perror("mmap"); exit(1); }
				double b = now();
				ns_map += (long long)((b - a) * 1e9);
				if (mode == 1) {
					CK(hipMemcpyAsync(pinned, dsrc, W, hipMemcpyDeviceToHost, st));
					CK(hipStreamSynchronize(st));
					double c = now();
					ns_copy += (long long)((c - b) * 1e9);
					memcpy(m, pinned, W);
					double d = now();
					ns_cpu += (long long)((d - c) * 1e9);
					munmap(m, W);
					continue;
				}
#ifdef MADV_POPULATE_WRITE
				if (mode == 2 && madvise(m, W, MADV_POPULATE_WRITE) != 0) { perror("madvise"); exit(1); }
#endif
				double c = now();
				ns_pop += (long long)((c - b) * 1e9);
				CK(hipHostRegister(m, W, hipHostRegisterDefault));
				double d = now();
				ns_reg += (long long)((d - c) * 1e9);
				CK(hipMemcpyAsync(m, dsrc, W, hipMemcpyDeviceToHost, st));
				CK(hipStreamSynchronize(st));
				double e = now();
				ns_copy += (long long)((e - d) * 1e9);
				if (mode == 3) { kept[w] = {m, W}; continue; }
				CK(hipHostUnregister(m));
				munmap(m, W);
				ns_unreg += (long long)((now() - e) * 1e9);
			}
			if (pinned) CK(hipHostFree(pinned));
			CK(hipStreamDestroy(st));
		});
	for (std::thread &x : th) x.join();
	const double t1 = now();
	if (mode == 3)
		for (auto &k : kept)
			if (k.first) { CK(hipHostUnregister(k.first)); munmap(k.first, k.second); }
	const double t2 = now();
	// spot check: the file holds the text
	char buf[16];
	ssize_t r = pread(fd, buf, 16, (off_t)(total - 16));
	bool ok = r == 16 && buf[0] == 'S' && buf[15] == 'S';
	close(fd);
	unlink(path);
	printf("mode %d  %zu GB  window %zu MB  threads %d : %.2f s = %.1f GB/s%s | thread-seconds: mmap %.2f populate %.2f register %.2f d2h %.2f unregister+munmap %.2f cpu-copy %.2f | content %s\n",
	       mode, total >> 30, W >> 20, T, t1 - t0, (double)total / (t1 - t0) / 1e9, mode == 3 ? " (+ final unregister)" : "", 1e-9 * ns_map, 1e-9 * ns_pop, 1e-9 * ns_reg, 1e-9 * ns_copy, 1e-9 * ns_unreg, 1e-9 * ns_cpu, ok ? "ok" : "WRONG");
	if (mode == 3) printf("        final unregister of all windows: %.2f s\n", t2 - t1);
	return ok ? 0 : 1;
}
