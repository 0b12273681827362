// hip_backend.cpp -- the product's only KernelBackend: libkart_amd.so through its C ABI.
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <ctime>

#include "mapper.hpp"

namespace kart {

int cli_main(int argc, char **argv, KernelBackend *(*make_backend)(const Options &, std::string &));

namespace {

[[noreturn]] void die(const char *what)
{
	// same convention as the reference: fatal errors terminate the run (src/Mapping.cpp:658-662)
	fprintf(stderr, "Error! %s: %s\n", what, kg_last_error());
	exit(1);
}

static double now_sec()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

class HipStream : public StreamBackend {
public:
	HipStream(kg_stream *s, const kg_stream_config &cfg) : s_(s), cfg_(cfg) {}
	~HipStream() override { kg_stream_close(s_); }
	int lanes() const override { return cfg_.lanes; }
	int64_t max_reads() const override { return cfg_.max_reads; }
	int64_t max_window() const override { return cfg_.max_window; }
	char *staging(int lane, int file) override { return kg_stream_staging(s_, lane, file, nullptr); }
	void upload(int lane, int file, int64_t from, int64_t to) override
	{
		if (kg_stream_upload(s_, lane, file, from, to) != KG_OK) die("kg_stream_upload");
	}
	bool parse(int lane, const kg_stream_window &w, kg_stream_parsed &out) override
	{
		int rc = kg_stream_parse(s_, lane, &w, &out);
		if (rc == KG_ERR_CAPACITY) return false;
		if (rc != KG_OK) die("kg_stream_parse");
		return true;
	}
	void map(int lane, const kg_stream_params &p, kg_stream_result &out) override
	{
		if (kg_stream_map(s_, lane, &p, &out) != KG_OK) die("kg_stream_map");
	}
	void fetch(int lane, int64_t first, int64_t count) override
	{
		if (kg_stream_fetch(s_, lane, first, count) != KG_OK) die("kg_stream_fetch");
	}
	bool timing(kg_stream_timing_t &t, bool reset) override { return kg_stream_timing(s_, &t, reset ? 1 : 0) == KG_OK; }
	int seed_group() const override { return cfg_.seed_group > 1 ? cfg_.seed_group : 0; }
	void group_absent(int lane, int rounds) override
	{
		if (kg_stream_group_absent(s_, lane, rounds) != KG_OK) die("kg_stream_group_absent");
	}
	kg_stream *handle() const { return s_; }

private:
	kg_stream *s_;
	kg_stream_config cfg_;
};

// a page-locked array that keeps its capacity (the batched calls below stage their inputs and outputs here: copies to and from the
// device then run at the link's rate instead of through the runtime's own staging)
template <class T> struct PinnedBuf {
	T *p = nullptr;
	size_t cap = 0;
	~PinnedBuf() { if (p) kg_host_free(p); }
	T *get(size_t n)
	{
		if (n > cap) {
			if (p) kg_host_free(p);
			cap = n + n / 4 + 1024;
			p = (T *)kg_host_alloc(cap * sizeof(T));
			if (!p) { fprintf(stderr, "Error! out of page-locked memory\n"); exit(1); }
		}
		return p;
	}
};

class HipBackend : public KernelBackend {
public:
	double t_frag_in = 0, t_frag_call = 0; int64_t n_frag_calls = 0;
	double t_seed = 0, t_cands = 0, t_copy = 0, t_align = 0, t_reccopy = 0;   // KART_AMD_VERBOSE: where the per-batch device stage spends its time
	HipBackend(kg_index *ix, const Options &opt) : ix_(ix), threads_(std::max(1, std::min(opt.threads, 16)))
	{
		kg_index_info(ix_, &info_);
		int64_t max_reads = std::max<int64_t>(opt.batch_reads, 4000ll * 4 * std::max(1, opt.threads)) + 8192;
		// (KART_AMD_TINY_WORKSPACE: test aid -- start far too small, so that the workspace is outgrown and retired several times
		//  while results of earlier batches are still being read)
		if (getenv("KART_AMD_TINY_WORKSPACE")) reserve(2048, 1 << 16);
		else reserve(max_reads, max_reads * 256);
	}
	// long reads: the batches alternate between two workspaces, so that the report of batch k (kg_longread_batch: the fragment and NW
	// kernels, instruction-bound) runs beside the seeding of batch k + 1 (the FM-index search, latency-bound) on streams of their own
	int long_slot() const override { return cur_; }
	bool long_overlap() const override { return !getenv("KART_AMD_LONG_NO_OVERLAP"); }
	~HipBackend() override
	{
		stream_.reset();
		for (Slot &sl : slots_) {
			for (std::pair<kg_workspace *, int> &r : sl.retired) kg_workspace_destroy(r.first);
			if (sl.ws) kg_workspace_destroy(sl.ws);
		}
		kg_index_destroy(ix_);
	}
	bool has_stream() const override { return getenv("KART_AMD_NO_STREAM") == nullptr && getenv("KART_AMD_HOST_ALIGN") == nullptr; }
	StreamBackend *stream(int64_t max_reads, int64_t max_window, int lanes, int seed_group) override
	{
		static const bool off = getenv("KART_AMD_NO_STREAM") != nullptr || getenv("KART_AMD_HOST_ALIGN") != nullptr;      // A/B aids: the host parses and prints, as before
		if (off) return nullptr;
		if (seed_group < 2 || lanes % seed_group != 0) seed_group = 0;
		if (stream_ && stream_->max_reads() >= max_reads && stream_->max_window() >= max_window &&
		    (seed_group ? stream_->lanes() == lanes : stream_->lanes() >= lanes) && stream_->seed_group() == seed_group) return stream_.get();
		stream_.reset();
		kg_stream_config cfg;
		cfg.max_reads = max_reads; cfg.max_window = max_window; cfg.lanes = lanes; cfg.seed_group = seed_group;
		kg_stream *s = nullptr;
		if (kg_stream_open(ix_, &cfg, &s) != KG_OK) {
			fprintf(stderr, "Warning! no device stream (%s): the host parses and prints\n", kg_last_error());
			return nullptr;
		}
		stream_.reset(new HipStream(s, cfg));
		return stream_.get();
	}
	int min_seed_len() const override { return info_.min_seed_len; }
	void *host_alloc(size_t bytes) override { return kg_host_alloc(bytes); }
	void host_free(void *p) override { kg_host_free(p); }
	void seed_and_chain(int mode, bool pacbio, int max_gaps, const uint8_t *enc, const std::vector<int64_t> &off,
	                    std::vector<int32_t> &n_cands, std::vector<int64_t> &cand_off, std::vector<kg_candidate> &, std::vector<kg_seed> &,
	                    const kg_candidate *&cands, const kg_seed *&cand_seeds) override
	{
		int64_t n = (int64_t)off.size() - 1;
		static const bool overlap = !getenv("KART_AMD_LONG_NO_OVERLAP");
		cur_ = (pacbio && overlap) ? (cur_ ^ 1) : 0;
		reserve(n, off[(size_t)n]);
		kg_workspace *const ws_ = slots_[cur_].ws;
		seed_off_.assign(off.size(), 0);
		double t0 = now_sec();
		// the seeds stay on the device (seeds = NULL); only the chained candidates come back, packed
		if (kg_seed_batch(ws_, mode | KG_INPUT_ASCII, info_.min_seed_len, KG_OCC_THR_DEFAULT, enc, off.data(), n, seed_off_.data(), nullptr) != KG_OK) die("kg_seed_batch");
		double t1 = now_sec();
		n_cands.assign((size_t)n + 1, 0);
		const kg_candidate *c = nullptr;
		const kg_seed *cs = nullptr;
		int64_t nc = 0, ns = 0;
		if (kg_candidates_batch(ws_, pacbio ? 1 : 0, max_gaps, n, seed_off_[(size_t)n], n_cands.data(), &c, &nc, &cs, &ns) != KG_OK) die("kg_candidates_batch");
		double t2 = now_sec();
		cands = c;                        // the library rotates its pinned arrays: valid while the next three batches pass
		cand_seeds = cs;
		cand_off.resize((size_t)n + 1);
		int64_t at = 0;
		for (int64_t r = 0; r < n; ++r) { cand_off[(size_t)r] = at; at += n_cands[(size_t)r]; }
		cand_off[(size_t)n] = at;
		double t3 = now_sec();
		t_seed += t1 - t0; t_cands += t2 - t1; t_copy += t3 - t2;
	}
	bool align(const std::vector<int64_t> &chunk_off, const std::vector<uint8_t> &chunk_paired, int est, int max_insert, int max_gaps,
	           bool multi_hit, int unset_flag, const kg_aln_record *&records, std::vector<kg_chunk_stats> &chunk_stats) override
	{
		static const bool off = getenv("KART_AMD_HOST_ALIGN") != nullptr;      // A/B aid: the whole report on the host, as before
		if (off) return false;
		int n_chunks = (int)chunk_paired.size();
		chunk_stats.resize((size_t)n_chunks);
		const kg_aln_record *rec = nullptr;
		double t0 = now_sec();
		kg_workspace *const ws_ = slots_[cur_].ws;
		if (kg_align_batch(ws_, chunk_off.data(), chunk_paired.data(), n_chunks, est, max_insert, max_gaps, multi_hit ? 1 : 0, unset_flag, &rec, chunk_stats.data()) != KG_OK) die("kg_align_batch");
		double t1 = now_sec();
		records = rec;                    // (rotating pinned arrays, as above)
		t_align += t1 - t0; t_reccopy += now_sec() - t1;
		return true;
	}
	bool long_enabled() const override
	{
		static const bool off = getenv("KART_AMD_HOST_ALIGN") != nullptr || getenv("KART_AMD_HOST_LONG") != nullptr;      // A/B aids: the whole report on the host, as before
		return !off;
	}
	bool align_long(int slot, const std::vector<int64_t> &chunk_off, const kg_aln_record *&records, const char *&cigar_pool, std::vector<kg_chunk_stats> &chunk_stats) override
	{
		if (!long_enabled()) return false;
		kg_workspace *const ws_ = slots_[slot & 1].ws;
		const kg_aln_record *rec = nullptr;
		const char *pool = nullptr;
		int64_t bytes = 0, n_host = 0;
		double t0 = now_sec();
		if (kg_longread_batch(ws_, &rec, &pool, &bytes, &n_host) != KG_OK) {
			// (device memory for a very large batch beside a 168 GB index, ...: this batch's report is the host's, the run goes on)
			static std::atomic<int> warned{0};
			if (warned++ == 0) fprintf(stderr, "Warning! kg_longread_batch: %s -- the host maps this batch\n", kg_last_error());
			return false;
		}
		double t1 = now_sec();
		const size_t n_chunks = chunk_off.size() - 1;
		chunk_stats.assign(n_chunks, kg_chunk_stats());
		for (size_t c = 0; c < n_chunks; ++c) {
			kg_chunk_stats &cs = chunk_stats[c];
			cs.paired = 0; cs.distance = 0; cs.lo = -1; cs.hi = INT64_MAX; cs.unmapped = 0; cs.unique = 0; cs.host_pairs = 0; cs.rescue_wanted = 0;
			for (int64_t r = chunk_off[c]; r < chunk_off[c + 1]; ++r) {
				if (rec[r].kind == KG_ALN_HOST) cs.host_pairs++;
				else if (rec[r].kind == KG_ALN_UNMAPPED) cs.unmapped++;          // what OutputSingledAlignments counts (src/Mapping.cpp:278,288)
				else if (rec[r].mapq == 60) cs.unique++;
			}
		}
		records = rec;
		cigar_pool = pool;
		{
			std::lock_guard<std::mutex> tl(frag_mu_);
			t_align += t1 - t0; t_reccopy += now_sec() - t1;
			long_used_ = true;
		}
		return true;
	}
	std::string align_diagnostics() override
	{
		uint64_t w[16];
		kg_workspace *const ws_ = slots_[0].ws;
		if (kg_align_reasons(ws_, w) != KG_OK) return std::string();
		static const char *const name[13] = {"candidate product", "mate-2 window", "window length", "mate characters/length", "runs per window", "rescued pairs", "seeds",
		                                     "gap pairs", "8-mer partition", "list capacity", "CIGAR length", "score", "read length"};
		char tb[768];
		snprintf(tb, sizeof(tb), "device stage seconds: seed (H2D + kernels) %.3f | chain + D2H %.3f | candidate copies %.3f | align (kernels + D2H) %.3f | record copy %.3f || ", t_seed, t_cands, t_copy, t_align, t_reccopy);
		std::string s(tb);
		if (long_used_) {
			uint64_t lw[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, l1[16];
			bool ok = false;
			for (const Slot &sl : slots_)
				if (sl.ws && kg_longread_reasons(sl.ws, l1) == KG_OK) { ok = true; for (int i = 0; i < 16; ++i) lw[i] += l1[i]; }
			if (ok) {
				snprintf(tb, sizeof(tb), "long-read report on the device: %llu reads, %llu handed back (candidates: literal '-' %llu, fragment envelope %llu, seed order %llu, element pool %llu; sequential overlap check %llu; fragment tasks handed back: length %llu, characters %llu, matches %llu, pairs %llu, lists %llu, depth %llu) || ",
				         (unsigned long long)lw[0], (unsigned long long)lw[1], (unsigned long long)lw[2], (unsigned long long)lw[3], (unsigned long long)lw[4], (unsigned long long)lw[5], (unsigned long long)lw[6],
				         (unsigned long long)lw[8], (unsigned long long)lw[9], (unsigned long long)lw[10], (unsigned long long)lw[11], (unsigned long long)lw[12], (unsigned long long)lw[13]);
				s += tb;
			}
		}
		if (n_frag_calls) {
			snprintf(tb, sizeof(tb), "fragment service: %lld calls, copying the parts together %.3f s, kg_fragments_batch %.3f s || ", (long long)n_frag_calls, t_frag_in, t_frag_call);
			s += tb;
		}
		for (int i = 0; i < 13; ++i)
			if (w[i]) s += std::string(name[i]) + " " + std::to_string((unsigned long long)w[i]) + ", ";
		s += "| before the last batch: parked candidates " + std::to_string((unsigned long long)w[13]) + ", NW jobs " + std::to_string((unsigned long long)w[14]) +
		     ", partition plans " + std::to_string((unsigned long long)w[15]);
		return s;
	}
	bool has_fragments() const override { static const bool off = getenv("KART_AMD_HOST_FRAGMENTS") != nullptr; return !off; }
	bool fragments_batch(std::vector<FragJobs *> &parts, bool pacbio, int max_gaps) override
	{
		if (!has_fragments()) return false;
		// one of kFragSets sets of page-locked arrays, in rotation: up to three calls may be in flight (pipeline.inc keeps at most
		// frag_depth <= 3 slices pending) while the results of the call before them are still being read
		int64_t n = 0, b1 = 0, cols = 0;
		for (FragJobs *p : parts) { n += (int64_t)p->size(); b1 += (int64_t)p->f1.size(); cols += p->cols; }
		if (n == 0) return true;
		const double t0 = now_sec();
		FragIo &io = fr_[(size_t)(fr_next_.fetch_add(1) % kFragSets)];
		std::lock_guard<std::mutex> lk(io.mu);
		char *f1 = io.f1.get((size_t)b1 + 64);
		int64_t *o1 = io.o1.get((size_t)n + 1), *g = io.g.get((size_t)n), *oo = io.oo.get((size_t)n);
		int32_t *gl = io.gl.get((size_t)n), *len = io.len.get((size_t)n);
		uint8_t *ops = io.ops.get((size_t)cols + 64), *status = io.status.get((size_t)n);
		// where every part starts in the combined arrays, then the parts are copied side by side (1.3 GB of characters per 200 k long reads)
		std::vector<int64_t> p_at(parts.size() + 1, 0), p_a1(parts.size() + 1, 0), p_ac(parts.size() + 1, 0);
		for (size_t k = 0; k < parts.size(); ++k) {
			p_at[k + 1] = p_at[k] + (int64_t)parts[k]->size(); p_a1[k + 1] = p_a1[k] + (int64_t)parts[k]->f1.size(); p_ac[k + 1] = p_ac[k] + parts[k]->cols;
		}
		o1[0] = 0;
		auto copy_parts = [&](size_t k0, size_t k1) {
			for (size_t k = k0; k < k1; ++k) {
				const FragJobs *p = parts[k];
				const int64_t at = p_at[k], a1 = p_a1[k], ac = p_ac[k];
				memcpy(f1 + a1, p->f1.data(), p->f1.size());
				for (size_t j = 0; j < p->size(); ++j) {
					o1[(size_t)at + j + 1] = a1 + p->o1[j + 1];
					g[(size_t)at + j] = p->g[j]; gl[(size_t)at + j] = p->gl[j]; oo[(size_t)at + j] = ac + p->oo[j];
				}
			}
		};
		{
			const size_t nt = std::min<size_t>((size_t)std::max(1, threads_ / 2), std::max<size_t>(1, parts.size() / 8));
			std::vector<std::thread> th;
			for (size_t t = 1; t < nt; ++t) th.emplace_back(copy_parts, parts.size() * t / nt, parts.size() * (t + 1) / nt);
			copy_parts(0, parts.size() / nt);
			for (std::thread &x : th) x.join();
		}
		const double t1 = now_sec();
		if (kg_fragments_batch(ix_, f1, o1, g, gl, n, pacbio ? 1 : 0, max_gaps, ops, oo, len, status) != KG_OK) die("kg_fragments_batch");
		{
			std::lock_guard<std::mutex> tl(frag_mu_);
			t_frag_in += t1 - t0; t_frag_call += now_sec() - t1; n_frag_calls++;
		}
		// the results stay where the device wrote them: every part gets its window (valid until this set comes round again, kFragSets calls later)
		for (size_t k = 0; k < parts.size(); ++k) {
			FragJobs *p = parts[k];
			p->ops = ops + p_ac[k]; p->len = len + p_at[k]; p->status = status + p_at[k];
		}
		return true;
	}
	void nw_batch(std::vector<NwJobs *> &parts) override
	{
		std::lock_guard<std::mutex> lk(nw_mu_);   // the staging vectors below are shared; calls from the commit path are rare
		// concatenate the parts (a few large copies), one kernel call, scatter the op strings back
		int64_t n = 0, b1 = 0, b2 = 0;
		for (NwJobs *p : parts) { n += (int64_t)p->size(); b1 += (int64_t)p->f1.size(); b2 += (int64_t)p->f2.size(); }
		f1_.resize((size_t)b1 + 1); f2_.resize((size_t)b2 + 1);
		o1_.resize((size_t)n + 1); o2_.resize((size_t)n + 1);
		ops_.resize((size_t)(b1 + b2) + 1); len_.resize((size_t)n);
		int64_t at = 0, a1 = 0, a2 = 0;
		o1_[0] = o2_[0] = 0;
		for (NwJobs *p : parts) {
			memcpy(&f1_[(size_t)a1], p->f1.data(), p->f1.size());
			memcpy(&f2_[(size_t)a2], p->f2.data(), p->f2.size());
			for (size_t j = 1; j <= p->size(); ++j) { o1_[(size_t)at + j] = a1 + p->o1[j]; o2_[(size_t)at + j] = a2 + p->o2[j]; }
			at += (int64_t)p->size(); a1 += (int64_t)p->f1.size(); a2 += (int64_t)p->f2.size();
		}
		if (kg_nw_batch(ix_, f1_.data(), o1_.data(), f2_.data(), o2_.data(), n, ops_.data(), len_.data()) != KG_OK) die("kg_nw_batch");
		at = 0; a1 = 0; a2 = 0;
		for (NwJobs *p : parts) {
			// the ops of a part are contiguous in the combined array: ops offset of job j = o1 + o2
			p->ops.assign(ops_.begin() + (a1 + a2), ops_.begin() + (a1 + a2) + (int64_t)(p->f1.size() + p->f2.size()));
			p->len.assign(len_.begin() + at, len_.begin() + at + (int64_t)p->size());
			at += (int64_t)p->size(); a1 += (int64_t)p->f1.size(); a2 += (int64_t)p->f2.size();
		}
	}

private:
	// the seeding workspace grows with the largest batch seen (long-read batches are far larger in bases)
	void reserve(int64_t reads, int64_t bases)
	{
		Slot &sl = slots_[cur_];
		// a workspace that was outgrown is retired, not destroyed: its page-locked result arrays (candidates, records) are still
		// being read by the stages of up to kRing - 1 earlier batches; it is freed once that many further batches have passed
		for (size_t i = 0; i < sl.retired.size();) {
			if (--sl.retired[i].second <= 0) { kg_workspace_destroy(sl.retired[i].first); sl.retired.erase(sl.retired.begin() + (std::ptrdiff_t)i); }
			else ++i;
		}
		if (sl.ws && reads <= sl.cap_reads && bases <= sl.cap_bases) return;
		if (sl.ws) sl.retired.emplace_back(sl.ws, 4);
		sl.ws = nullptr;
		sl.cap_reads = std::max(sl.cap_reads, reads + reads / 4 + 1024);
		sl.cap_bases = std::max(sl.cap_bases, bases + bases / 4 + 65536);
		if (kg_workspace_create(ix_, sl.cap_reads, sl.cap_bases, &sl.ws) != KG_OK) die("kg_workspace_create");
	}
	struct Slot {
		kg_workspace *ws = nullptr;
		int64_t cap_reads = 0, cap_bases = 0;
		std::vector<std::pair<kg_workspace *, int>> retired;   // outgrown workspaces and the batches left until their arrays are unused
	};
	Slot slots_[2];
	int cur_ = 0;                   // the slot of the batch seeded last
	bool long_used_ = false;
	kg_index *ix_;
	int threads_;
	std::mutex nw_mu_, frag_mu_;
	static constexpr int kFragSets = 5;        // 3 calls in flight + the one whose results are being read + one spare
	struct FragIo {
		std::mutex mu;
		PinnedBuf<char> f1;
		PinnedBuf<int64_t> o1, g, oo;
		PinnedBuf<int32_t> gl, len;
		PinnedBuf<uint8_t> ops, status;
	};
	FragIo fr_[kFragSets];
	std::atomic<uint64_t> fr_next_{0};
	std::vector<char> f1_, f2_;
	std::vector<int64_t> o1_, o2_;
	std::vector<uint8_t> ops_;
	std::vector<int32_t> len_;
	std::vector<int64_t> seed_off_;
	std::unique_ptr<HipStream> stream_;
	kg_index_info_t info_;
};

}  // namespace

KernelBackend *make_hip_backend(const Options &opt, std::string &err)
{
	kg_index *ix = nullptr;
	// KART_AMD_SA = auto (default: full below 2^32 text symbols; above, wide where the device has room, else compact) | full | compact | wide | dense4 | dense8 | sampled: how much of the suffix array
	// stays on the device (include/kart_amd.h); the results are the same, the compact index takes 67 GB instead of 168 GB for a human genome
	int sa_mode = opt.sa_mode;
	if (const char *e = getenv("KART_AMD_SA")) {
		const std::string v(e);
		if (v == "auto") sa_mode = KG_SA_AUTO;
		else if (v == "full") sa_mode = KG_SA_FULL;
		else if (v == "compact") sa_mode = KG_SA_FULL40;
		else if (v == "wide") sa_mode = KG_SA_FULL40_WIDE;
		else if (v == "dense4") sa_mode = KG_SA_DENSE4;
		else if (v == "dense8") sa_mode = KG_SA_DENSE8;
		else if (v == "sampled") sa_mode = KG_SA_SAMPLED;
		else { err = "KART_AMD_SA must be one of auto, full, compact, wide, dense4, dense8, sampled"; return nullptr; }
	}
	if (kg_index_load(opt.index_prefix.c_str(), opt.device, sa_mode, &ix) != KG_OK) {
		err = kg_last_error();
		return nullptr;
	}
	return new HipBackend(ix, opt);
}

}  // namespace kart

#ifndef KART_NO_MAIN
int main(int argc, char **argv) { return kart::cli_main(argc, argv, kart::make_hip_backend); }
#endif
