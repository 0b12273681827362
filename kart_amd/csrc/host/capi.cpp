// capi.cpp -- extern "C" boundary of the host pipeline (include/kart_host.h): index resident across mapping runs.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/kart_host.h"
#include "mapper.hpp"

namespace kart {
KernelBackend *make_hip_backend(const Options &opt, std::string &err);
}

struct kh_session {
	kart::Options base;
	kart::RefData ref;
	std::unique_ptr<kart::KernelBackend> kern;
};

namespace {
thread_local char g_err[512] = "";
int fail(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return 1;
}
}  // namespace

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace {
// KART_AMD_BACKTRACE (diagnostics): a fault inside the library prints its frames as library offsets (addr2line -e libkart_host.so <offset>) before it ends the process
void fault_backtrace(int sig)
{
	void *frames[48];
	const int n = backtrace(frames, 48);
	const char msg[] = "kart-amd: fatal signal, frames:\n";
	(void)!write(2, msg, sizeof(msg) - 1);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
}

extern "C" {

const char *kh_last_error(void) { return g_err; }

int kh_open(const char *index_prefix, int device, int threads, kh_session **out)
{
	if (getenv("KART_AMD_BACKTRACE")) { signal(SIGSEGV, fault_backtrace); signal(SIGABRT, fault_backtrace); }
	if (!index_prefix || !out) return fail("kh_open: null argument");
	*out = nullptr;
	std::unique_ptr<kh_session> s(new kh_session());
	s->base.index_prefix = index_prefix;
	s->base.device = device;
	s->base.threads = threads > 0 ? threads : 4;
	std::string ref_err, err;
	bool ref_ok = false;
	std::thread loader([&]() { ref_ok = s->ref.load(s->base.index_prefix, ref_err, s->base.threads); });
	s->kern.reset(kart::make_hip_backend(s->base, err));
	loader.join();
	if (!s->kern) return fail("kh_open: %s", err.c_str());
	if (!ref_ok) return fail("kh_open: %s", ref_err.c_str());
	*out = s.release();
	return 0;
}

int kh_map(kh_session *s, int argc, const char *const *argv, kh_stats_t *stats)
{
	if (!s || argc < 0 || (argc > 0 && !argv)) return fail("kh_map: bad argument");
	std::vector<std::string> store{"kh_map", "-i", s->base.index_prefix};
	for (int i = 0; i < argc; ++i) {
		if (!argv[i]) return fail("kh_map: null flag");
		if (!strcmp(argv[i], "-i") || !strcmp(argv[i], "-t") || !strcmp(argv[i], "-gpu")) return fail("kh_map: %s is fixed by the session", argv[i]);
		store.push_back(argv[i]);
	}
	std::vector<char *> av;
	for (std::string &x : store) av.push_back(&x[0]);
	kart::Options opt = s->base;
	int rc = kart::parse_cli((int)av.size(), av.data(), opt);
	if (rc < 0) return fail("kh_map: bad command line (status %d)", -rc - 1);
	opt.device = s->base.device;
	opt.threads = s->base.threads;
	if (opt.shard_count > 1 && opt.rendezvous.empty()) return fail("kh_map: -shard needs -rendezvous FILE");
	FILE *out = nullptr;
	if (opt.shard_rank == 0 || opt.parts) {
		out = kart::open_output(opt.parts && opt.shard_count > 1 ? opt.out_name + "." + std::to_string(opt.shard_rank) : opt.out_name);
		if (!out) {
			if (opt.shard_count > 1) kart::shard_mark_failed(opt.rendezvous, opt.shard_rank);
			return fail("kh_map: cannot open [%s]", opt.out_name.c_str());
		}
	}
	kart::Stats st;
	auto now = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
	double t0 = now();
	rc = kart::run_mapping(opt, s->ref, *s->kern, out, st);
	if (rc != 0 && opt.shard_count > 1) kart::shard_mark_failed(opt.rendezvous, opt.shard_rank);     // the other shards stop waiting for this one
	double t1 = now();
	if (out) fclose(out);
	if (getenv("KART_AMD_VERBOSE")) fprintf(stdout, "kh_map: run_mapping %.3f s (its own mapping seconds %.3f), closing the output %.3f s\n", t1 - t0, st.map_seconds, now() - t1);
	if (stats) {
		stats->total_reads = st.total_reads; stats->unmapped = st.unmapped; stats->unique = st.unique; stats->paired = st.paired;
		stats->distance = st.distance; stats->respeculated = st.respeculated; stats->map_seconds = st.map_seconds; stats->sharded = st.sharded ? 1 : 0;
		stats->pad = 0;
		stats->stream_reads = st.stream_reads; stats->stream_batches = st.device.batches;
		const double ms[6] = {st.device.parse_ms, st.device.seed_ms, st.device.chain_ms, st.device.align_ms, st.device.format_ms, st.device.copy_ms};
		for (int i = 0; i < 6; ++i) stats->stage_ms[i] = ms[i];
		stats->search_kernel_ms = st.device.search_kernel_ms; stats->search_kernel_launches = st.device.search_kernel_launches;
		stats->search_useful_bytes = st.device.search_useful_bytes;
		stats->text_in_bytes = st.device.text_in_bytes; stats->text_out_bytes = st.device.text_out_bytes;
		stats->candidates = st.device.candidates; stats->candidate_seeds = st.device.candidate_seeds;
		for (int i = 0; i < 16; ++i) { stats->kernel_ms[i] = st.device.kernel_ms[i]; stats->kernel_launches[i] = st.device.kernel_launches[i]; }
		for (int i = 0; i < 8; ++i) stats->aln_counts[i] = st.device.aln_counts[i];
		for (int i = 0; i < 6; ++i) stats->lane_seconds[i] = st.lane_seconds[i];
		stats->lanes = st.lanes; stats->pad2 = 0;
		stats->text_checksum[0] = st.device.text_checksum[0]; stats->text_checksum[1] = st.device.text_checksum[1];
	}
	if (rc == 0) return 0;
	const std::string why = kart::run_error_message();
	return why.empty() ? fail("kh_map: mapping failed") : fail("kh_map: %s", why.c_str());
}

void kh_close(kh_session *s) { delete s; }

}  // extern "C"
