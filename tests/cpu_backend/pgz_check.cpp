// tests/cpu_backend/pgz_check.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The several-thread reader of ordinary gzip files (kart_amd/csrc/host/detail/pgzip.inc) against zlib's gzread() on the same file:
// every byte the parallel reader delivers must be the byte zlib delivers at that offset, and when it finishes a file it must have
// delivered all of it.  Usage: pgz_check file.gz [threads] [chunk_kb]; prints one JSON line
// {"bytes": text bytes of the parallel reader, "zlib_bytes": ..., "all_done": 0|1, "reads": n, "equal": 0|1, "pgz_s": seconds, "zlib_s": seconds};
// exit code 1 when a delivered byte differs.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace kart {
namespace {
#include "../../kart_amd/csrc/host/detail/pgzip.inc"
}
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: pgz_check file.gz [threads] [chunk_kb]\n"); return 2; }
	const int threads = argc > 2 ? atoi(argv[2]) : 8;
	if (argc > 3) setenv("KART_AMD_PGZ_CHUNK_KB", argv[3], 1);
	std::vector<char> ref;
	const double z0 = now();
	{
		gzFile g = gzopen(argv[1], "rb");
		if (!g) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
		gzbuffer(g, 1 << 20);
		std::vector<char> buf((size_t)16 << 20);
		bool damaged = false;
		for (;;) { int n = gzread(g, buf.data(), (unsigned)buf.size()); if (n < 0) damaged = true; if (n <= 0) break; ref.insert(ref.end(), buf.data(), buf.data() + n); }
		gzclose(g);
		if (damaged) {
			// a damaged stream: a large gzread() fails as a whole; what zlib could still inflate is what small reads deliver before the error
			ref.clear();
			g = gzopen(argv[1], "rb");
			for (;;) { int n = gzread(g, buf.data(), 1); if (n <= 0) break; ref.push_back(buf[0]); }
			gzclose(g);
		}
	}
	const double zlib_s = now() - z0;
	kart::Pgz z;
	std::vector<char> text(ref.size() + ((size_t)64 << 20), 'x');      // (touched: the reader is timed, not this buffer's page faults)
	size_t have = 0;
	int rounds = 0;
	const double p0 = now();
	const bool opened = z.open(argv[1], threads);
	if (opened)
		for (;;) {                                                      // (the size asked for is not a multiple of anything inside the reader)
			const size_t ask = std::min<size_t>(((size_t)48 << 20) - 12345, text.size() - have);
			const size_t n = z.read(text.data() + have, ask);
			have += n; rounds++;
			if (n < ask || have == text.size()) break;
		}
	const double pgz_s = now() - p0;
	text.resize(have);
	const bool prefix = text.size() <= ref.size() && memcmp(text.data(), ref.data(), text.size()) == 0;
	const bool equal = prefix && (!z.all_done || text.size() == ref.size());
	printf("{\"opened\": %d, \"bytes\": %zu, \"zlib_bytes\": %zu, \"all_done\": %d, \"reads\": %d, \"equal\": %d, \"pgz_s\": %.3f, \"zlib_s\": %.3f}\n",
	       opened ? 1 : 0, text.size(), ref.size(), z.all_done ? 1 : 0, rounds, equal ? 1 : 0, pgz_s, zlib_s);
	return equal ? 0 : 1;
}
