// tests/cpu_backend/gzproducer_check.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A gz library on its way to the device stream (kart_amd/csrc/host/detail/batch_reader.inc: GzText::fill_direct, GzProducer) without a device:
// a consumer plays the stream's part -- it waits for the next block of the growing text, checks its bytes against what zlib reads from the
// same file, and releases what lies behind it (a small look-ahead makes the writer wait for that) -- and, in a second pass, stops half-way the
// way Source::gz_stream_end does: what exists and was not consumed becomes the gz reader's carry, GzText::fill() delivers the rest, and the
// three parts together must again be zlib's text.  Usage: gzproducer_check file.gz [threads] [block_kb] [ahead_mb]; one JSON line, exit 1 on
// a difference.
#include <fcntl.h>
#include <immintrin.h>
#include <sched.h>
#include <signal.h>
#include <pthread.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <sys/vfs.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <functional>
#include <future>
#include <malloc.h>
#include <memory>
#include <mutex>
#include <string_view>
#include <thread>

#include "../../kart_amd/csrc/host/mapper.hpp"

namespace kart {
namespace {
#include "../../kart_amd/csrc/host/detail/types.inc"
#include "../../kart_amd/csrc/host/detail/normal_pairs.inc"
#include "../../kart_amd/csrc/host/detail/kmer.inc"
#include "../../kart_amd/csrc/host/detail/gap_closing.inc"
#include "../../kart_amd/csrc/host/detail/report.inc"
#include "../../kart_amd/csrc/host/detail/pairing.inc"
#include "../../kart_amd/csrc/host/detail/sam.inc"
#include "../../kart_amd/csrc/host/detail/bam.inc"
#include "../../kart_amd/csrc/host/detail/reader.inc"
#include "../../kart_amd/csrc/host/detail/shard.inc"
#include "../../kart_amd/csrc/host/detail/chunk_state.inc"
#include "../../kart_amd/csrc/host/detail/writer.inc"
#include "../../kart_amd/csrc/host/detail/chunk_stages.inc"
#include "../../kart_amd/csrc/host/detail/pgzip.inc"
#include "../../kart_amd/csrc/host/detail/batch_reader.inc"

struct Opened {
	GzText g;
	gzFile in = nullptr;
	bool open(const char *path, int threads)
	{
		in = gzopen(path, "rb");
		if (!in) return false;
		gzbuffer(in, 1 << 20);
		g.f = in;
		g.path = path;
		if (!g.try_bgzf(path, threads)) g.try_pgz(path, threads);
		return true;
	}
	~Opened() { if (in) gzclose(in); }
};

// what zlib reads (small reads once a large one has failed: what gzgets() would still see of a damaged stream)
std::vector<char> zlib_text(const char *path)
{
	std::vector<char> ref;
	gzFile g = gzopen(path, "rb");
	if (!g) return ref;
	gzbuffer(g, 1 << 20);
	std::vector<char> buf((size_t)16 << 20);
	bool damaged = false;
	for (;;) { int n = gzread(g, buf.data(), (unsigned)buf.size()); if (n < 0) damaged = true; if (n <= 0) break; ref.insert(ref.end(), buf.data(), buf.data() + n); }
	gzclose(g);
	if (damaged) {
		ref.clear();
		g = gzopen(path, "rb");
		for (;;) { int n = gzread(g, buf.data(), 1); if (n <= 0) break; ref.push_back(buf[0]); }
		gzclose(g);
	}
	return ref;
}

int check(const char *path, int threads, size_t block, bool stop_half, const std::vector<char> &ref, size_t &seen)
{
	struct stat sb;
	if (stat(path, &sb) != 0) return 2;
	Opened o;
	if (!o.open(path, threads)) return 2;
	GzProducer p;
	if (!p.start(&o.g, (size_t)sb.st_size)) return 2;
	size_t pos = 0;
	bool ended = false;
	int bad = 0;
	for (int step = 0;; ++step) {
		const size_t upto = pos + block + (size_t)(step % 7) * 1000;                    // (block sizes that are no multiple of anything inside)
		const size_t have = p.wait_for(upto, ended);
		const size_t end = std::min(have, upto);
		if (end > ref.size() || memcmp(p.base + pos, ref.data() + pos, end - pos) != 0) { bad = 1; break; }
		pos = end;
		if (pos > ((size_t)1 << 20)) p.release(pos - ((size_t)1 << 20));               // (the carry of the next batch may reach back a little)
		if (ended && pos == have) break;
		if (stop_half && pos >= ref.size() / 2) break;
	}
	if (!bad && stop_half) {
		// Source::gz_stream_end(): the writer stops; what exists behind `pos` is the gz reader's carry, fill() goes on behind it
		p.stop();
		std::vector<char> rest(p.base + pos, p.base + p.produced);
		if (o.g.delivered != p.produced) bad = 1;
		while (!o.g.eof) o.g.fill(rest, (size_t)8 << 20);
		if (pos + rest.size() != ref.size() || memcmp(rest.data(), ref.data() + pos, rest.size()) != 0) bad = 1;
		pos += rest.size();
	}
	seen = pos;
	if (!bad && pos != ref.size()) bad = 1;
	return bad;
}

}  // namespace
}  // namespace kart

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: gzproducer_check file.gz [threads] [block_kb] [ahead_mb]\n"); return 2; }
	const int threads = argc > 2 ? atoi(argv[2]) : 4;
	const size_t block = (size_t)(argc > 3 ? atoll(argv[3]) : 300) << 10;
	if (argc > 4) setenv("KART_AMD_GZ_AHEAD_MB", argv[4], 1);
	const std::vector<char> ref = kart::zlib_text(argv[1]);
	size_t seen_all = 0, seen_half = 0;
	const int a = kart::check(argv[1], threads, block, false, ref, seen_all);
	const int b = kart::check(argv[1], threads, block, true, ref, seen_half);
	printf("{\"zlib_bytes\": %zu, \"through_the_producer\": %zu, \"with_a_hand_over_half_way\": %zu, \"equal\": %d}\n", ref.size(), seen_all, seen_half, a == 0 && b == 0 ? 1 : 0);
	return a == 0 && b == 0 ? 0 : 1;
}
