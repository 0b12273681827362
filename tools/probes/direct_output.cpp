// tools/probes/direct_output.cpp -- MEASUREMENT AID (not part of the product).
//
// Can the SAM text go from the device straight into the pages of the output file?  Today it is copied twice on the host side of the link: D2H into a
// lane's page-locked buffer, then by the writer threads into a shared mapping of the (tmpfs) output file -- that second copy into FRESH pages is what
// bounds the step.  This probe registers windows of a shared mapping of a fresh /dev/shm file with the runtime
// (hipHostRegister: the pages are allocated and pinned there) and lets the copy engine write into them, against the product's way
// (page-locked buffer, then memcpy by T threads into the mapping).  Prints GB/s of each, per window size.
// Build: hipcc -O2 --offload-arch=gfx950 tools/probes/direct_output.cpp -o /tmp/direct_output -lpthread ; run: /tmp/direct_output [GB] [threads]
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
	const size_t total = (size_t)(argc > 1 ? atoll(argv[1]) : 8) << 30;
	const int threads = argc > 2 ? atoi(argv[2]) : 7;
	const size_t window = (size_t)1 << 30;
	char *dev = nullptr;
	CK(hipMalloc((void **)&dev, window));
	CK(hipMemset(dev, 0x41, window));
	hipStream_t st;
	CK(hipStreamCreate(&st));
	auto fresh = [&](const char *name, char *&map, int &fd) {
		fd = open(name, O_RDWR | O_CREAT | O_TRUNC, 0600);
		if (fd < 0 || ftruncate(fd, (off_t)total) != 0) return false;
		map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		return map != MAP_FAILED;
	};
	// ---- A: the product's way: D2H into a page-locked buffer, T threads copy it into the mapping ----
	{
		char *map; int fd;
		if (!fresh("/dev/shm/kart_probe_a", map, fd)) { printf("{\"error\": \"cannot map the file\"}\n"); return 1; }
		char *pin = nullptr;
		CK(hipHostMalloc((void **)&pin, window, hipHostMallocDefault));
		double t_d2h = 0, t_copy = 0;
		const double t0 = now();
		for (size_t off = 0; off < total; off += window) {
			double a = now();
			CK(hipMemcpyAsync(pin, dev, window, hipMemcpyDeviceToHost, st));
			CK(hipStreamSynchronize(st));
			double b = now();
			std::vector<std::thread> th;
			for (int t = 0; t < threads; ++t) th.emplace_back([&, t]() { size_t lo = window * (size_t)t / (size_t)threads, hi = window * (size_t)(t + 1) / (size_t)threads; memcpy(map + off + lo, pin + lo, hi - lo); });
			for (std::thread &x : th) x.join();
			t_d2h += b - a; t_copy += now() - b;
		}
		const double t1 = now();
		printf("{\"way\": \"page-locked buffer, then %d threads copy into the mapping\", \"GB\": %.1f, \"seconds\": %.3f, \"d2h_s\": %.3f, \"copy_s\": %.3f, \"GBps_serial\": %.2f, \"GBps_copy_alone\": %.2f}\n",
		       threads, total / 1e9, t1 - t0, t_d2h, t_copy, total / 1e9 / (t1 - t0), total / 1e9 / t_copy);
		(void)hipHostFree(pin); munmap(map, total); close(fd); unlink("/dev/shm/kart_probe_a");
	}
	// ---- B: the windows of the mapping registered with the runtime, the copy engine writes into the file's pages ----
	{
		char *map; int fd;
		if (!fresh("/dev/shm/kart_probe_b", map, fd)) { printf("{\"error\": \"cannot map the file\"}\n"); return 1; }
		double t_reg = 0, t_d2h = 0, t_unreg = 0;
		const double t0 = now();
		for (size_t off = 0; off < total; off += window) {
			double a = now();
			hipError_t e = hipHostRegister(map + off, window, hipHostRegisterDefault);
			if (e != hipSuccess) { printf("{\"way\": \"registered mapping\", \"error\": \"hipHostRegister: %s\"}\n", hipGetErrorString(e)); return 0; }
			double b = now();
			CK(hipMemcpyAsync(map + off, dev, window, hipMemcpyDeviceToHost, st));
			CK(hipStreamSynchronize(st));
			double c = now();
			CK(hipHostUnregister(map + off));
			t_reg += b - a; t_d2h += c - b; t_unreg += now() - c;
		}
		const double t1 = now();
		bool ok = true;
		for (size_t off = 0; off < total; off += (size_t)97 << 20) ok = ok && map[off] == 0x41;
		printf("{\"way\": \"windows of the mapping registered, the copy engine writes into the file's pages\", \"GB\": %.1f, \"seconds\": %.3f, \"register_s\": %.3f, \"d2h_s\": %.3f, \"unregister_s\": %.3f, "
		       "\"GBps_serial\": %.2f, \"bytes_arrived\": %s}\n", total / 1e9, t1 - t0, t_reg, t_d2h, t_unreg, total / 1e9 / (t1 - t0), ok ? "true" : "false");
		munmap(map, total); close(fd); unlink("/dev/shm/kart_probe_b");
	}
	// ---- C: as B, the registration of the NEXT window on a thread of its own while the copy engine fills this one ----
	{
		char *map; int fd;
		if (!fresh("/dev/shm/kart_probe_c", map, fd)) return 0;
		const size_t n = total / window;
		std::vector<int> state(n, 0);
		const double t0 = now();
		std::thread reg([&]() { for (size_t i = 0; i < n; ++i) { if (hipHostRegister(map + i * window, window, hipHostRegisterDefault) != hipSuccess) { state[i] = -1; return; } __atomic_store_n(&state[i], 1, __ATOMIC_RELEASE); } });
		bool ok = true;
		for (size_t i = 0; i < n && ok; ++i) {
			int s;
			while ((s = __atomic_load_n(&state[i], __ATOMIC_ACQUIRE)) == 0) std::this_thread::yield();
			if (s < 0) { ok = false; break; }
			if (hipMemcpyAsync(map + i * window, dev, window, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) ok = false;
		}
		reg.join();
		const double t1 = now();
		for (size_t i = 0; i < n; ++i) if (state[i] == 1) (void)hipHostUnregister(map + i * window);
		printf("{\"way\": \"as above, registration one window ahead on its own thread\", \"GB\": %.1f, \"seconds\": %.3f, \"GBps\": %.2f, \"ok\": %s}\n", total / 1e9, t1 - t0, total / 1e9 / (t1 - t0), ok ? "true" : "false");
		munmap(map, total); close(fd); unlink("/dev/shm/kart_probe_c");
	}
	// ---- D: the registration itself by T threads (disjoint 256 MB windows each): is pinning fresh pages of ONE file as parallel as copying into them? ----
	for (int T : {1, 2, 4, 7, 12}) {
		char *map; int fd;
		if (!fresh("/dev/shm/kart_probe_d", map, fd)) return 0;
		const size_t piece = (size_t)256 << 20, n = total / piece;
		std::vector<char> okv((size_t)T, 1);
		const double t0 = now();
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back([&, t]() { for (size_t i = (size_t)t; i < n; i += (size_t)T) if (hipHostRegister(map + i * piece, piece, hipHostRegisterDefault) != hipSuccess) { okv[(size_t)t] = 0; return; } });
		for (std::thread &x : th) x.join();
		const double t1 = now();
		bool ok = true;
		for (char c : okv) ok = ok && c;
		double d2h = 0;
		if (ok) {
			const double a = now();
			for (size_t off = 0; off + window <= total; off += window) { if (hipMemcpyAsync(map + off, dev, window, hipMemcpyDeviceToHost, st) != hipSuccess) ok = false; }
			if (hipStreamSynchronize(st) != hipSuccess) ok = false;
			d2h = now() - a;
		}
		for (size_t i = 0; i < n; ++i) (void)hipHostUnregister(map + i * piece);
		printf("{\"way\": \"registration by %d threads, then the copies\", \"GB\": %.1f, \"register_s\": %.3f, \"register_GBps\": %.2f, \"d2h_s\": %.3f, \"d2h_GBps\": %.1f, \"ok\": %s}\n", T, total / 1e9, t1 - t0, total / 1e9 / (t1 - t0), d2h,
		       d2h > 0 ? total / 1e9 / d2h : 0.0, ok ? "true" : "false");
		munmap(map, total); close(fd); unlink("/dev/shm/kart_probe_d");
	}
	return 0;
}
