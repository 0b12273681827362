// hip_backend.cpp -- the product's only KernelBackend: libkart_amd.so through its C ABI.
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <thread>

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

class HipBackend : public KernelBackend {
public:
	HipBackend(kg_index *ix, const Options &opt) : ix_(ix), threads_(std::max(1, std::min(opt.threads, 16)))
	{
		kg_index_info(ix_, &info_);
		int64_t max_reads = opt.batch_reads + 8192, max_bases = max_reads * 512;
		if (opt.pacbio) max_bases = std::max<int64_t>(max_bases, 64ll << 20);
		if (kg_workspace_create(ix_, max_reads, max_bases, &ws_) != KG_OK) die("kg_workspace_create");
	}
	~HipBackend() override
	{
		kg_workspace_destroy(ws_);
		kg_index_destroy(ix_);
	}
	int min_seed_len() const override { return info_.min_seed_len; }
	void seed_batch(int mode, const std::vector<uint8_t> &enc, const std::vector<int64_t> &off, std::vector<int64_t> &seed_off,
	                std::vector<kg_seed> &seeds) override
	{
		int64_t n = (int64_t)off.size() - 1;
		seed_off.assign(off.size(), 0);
		const kg_seed *out = nullptr;
		if (kg_seed_batch(ws_, mode, info_.min_seed_len, KG_OCC_THR_DEFAULT, enc.data(), off.data(), n, seed_off.data(), &out) != KG_OK) die("kg_seed_batch");
		seeds.assign(out, out + seed_off[(size_t)n]);
	}
	void nw_batch(std::vector<NwJob> &jobs) override
	{
		int64_t n = (int64_t)jobs.size();
		std::vector<int64_t> o1((size_t)n + 1, 0), o2((size_t)n + 1, 0);
		for (int64_t i = 0; i < n; ++i) {
			o1[(size_t)i + 1] = o1[(size_t)i] + (int64_t)jobs[(size_t)i].a.size();
			o2[(size_t)i + 1] = o2[(size_t)i] + (int64_t)jobs[(size_t)i].b.size();
		}
		std::vector<char> f1((size_t)o1[(size_t)n] + 1), f2((size_t)o2[(size_t)n] + 1);
		const int T = threads_;
		auto span = [&](int t, int64_t &lo, int64_t &hi) { lo = n * t / T; hi = n * (t + 1) / T; };
		auto fan = [&](const std::function<void(int)> &fn) {
			std::vector<std::thread> th;
			for (int t = 1; t < T; ++t) th.emplace_back(fn, t);
			fn(0);
			for (std::thread &x : th) x.join();
		};
		fan([&](int t) {
			int64_t lo, hi;
			span(t, lo, hi);
			for (int64_t i = lo; i < hi; ++i) {
				memcpy(f1.data() + o1[(size_t)i], jobs[(size_t)i].a.data(), jobs[(size_t)i].a.size());
				memcpy(f2.data() + o2[(size_t)i], jobs[(size_t)i].b.data(), jobs[(size_t)i].b.size());
			}
		});
		std::vector<uint8_t> ops(f1.size() + f2.size());
		std::vector<int32_t> len((size_t)n);
		if (kg_nw_batch(ix_, f1.data(), o1.data(), f2.data(), o2.data(), n, ops.data(), len.data()) != KG_OK) die("kg_nw_batch");
		fan([&](int t) {
			int64_t lo, hi;
			span(t, lo, hi);
			for (int64_t i = lo; i < hi; ++i) {
				NwJob &j = jobs[(size_t)i];
				const uint8_t *op = ops.data() + o1[(size_t)i] + o2[(size_t)i];
				int L = len[(size_t)i];
				j.ra.resize((size_t)L); j.rb.resize((size_t)L);
				size_t x = 0, y = 0;
				for (int q = 0; q < L; ++q) {
					if (op[q] == KG_OP_DIAG) { j.ra[(size_t)q] = j.a[x++]; j.rb[(size_t)q] = j.b[y++]; }
					else if (op[q] == KG_OP_GAP1) { j.ra[(size_t)q] = '-'; j.rb[(size_t)q] = j.b[y++]; }
					else { j.ra[(size_t)q] = j.a[x++]; j.rb[(size_t)q] = '-'; }
				}
			}
		});
	}

private:
	kg_index *ix_;
	int threads_;
	kg_workspace *ws_ = nullptr;
	kg_index_info_t info_;
};

KernelBackend *make_hip_backend(const Options &opt, std::string &err)
{
	kg_index *ix = nullptr;
	if (kg_index_load(opt.index_prefix.c_str(), opt.device, opt.sa_mode, &ix) != KG_OK) {
		err = kg_last_error();
		return nullptr;
	}
	return new HipBackend(ix, opt);
}

}  // namespace
}  // namespace kart

int main(int argc, char **argv) { return kart::cli_main(argc, argv, kart::make_hip_backend); }
