// tests/cpu_backend/oracle_backend.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Binds the product's HOST pipeline (kart_amd/csrc/host/mapper.cpp, cli.cpp) to the CPU oracle
// (oracle/liboracle.so) instead of the HIP library, so the host-side control flow (chaining, pairing,
// rescue, report, SAM text) can be checked against the reference's golden SAM in a container without a
// GPU.  The resulting binary (tests/_build/kart-host-oracle) is never shipped: the product binary
// kart_amd/bin/kart-amd links hip_backend.cpp and nothing else.
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>

#include "../../kart_amd/csrc/host/mapper.hpp"
#include "../../oracle/kart_oracle.h"

namespace kart {
int cli_main(int argc, char **argv, KernelBackend *(*make_backend)(const Options &, std::string &));

// The op string of an alignment the oracle returns as two gapped strings.  A '-' in the gapped read string is a gap the
// alignment inserted -- or a literal '-' of the read itself (the reference takes whatever the file holds).  Within a run of
// dashes of the gapped read string the read's own characters say how many are literal; the columns where the text string has
// a gap must be among those (two gaps never face each other); which of the others are is immaterial to either string.
static void ops_from_strings(const char *read, int m, const char *g1, const char *g2, int L, uint8_t *op)
{
	int i = 0;
	for (int t = 0; t < L;) {
		if (g1[t] != '-') { op[t] = g2[t] == '-' ? KG_OP_GAP2 : KG_OP_DIAG; ++i; ++t; continue; }
		int e = t;
		while (e < L && g1[e] == '-') ++e;               // the run [t, e)
		int literal = 0;
		while (i + literal < m && literal < e - t && read[i + literal] == '-') ++literal;
		if (e == L) literal = m - i;                    // (at the end every remaining read character is one)
		int forced = 0;
		for (int c = t; c < e; ++c) forced += g2[c] == '-';
		int spare = literal - forced;                   // literal dashes that face a text character
		for (int c = t; c < e; ++c) {
			if (g2[c] == '-') { op[c] = KG_OP_GAP2; ++i; }
			else if (spare > 0) { op[c] = KG_OP_DIAG; ++i; --spare; }
			else op[c] = KG_OP_GAP1;
		}
		t = e;
	}
}

class OracleBackend : public KernelBackend {
public:
	explicit OracleBackend(ko_index *ix) : ix_(ix) {}
	~OracleBackend() override { ko_index_free(ix_); }
	int min_seed_len() const override { return ko_min_seed_len(ix_); }
	void seed_and_chain(int mode, bool pacbio, int max_gaps, const uint8_t *enc, const std::vector<int64_t> &off,
	                    std::vector<int32_t> &n_cands, std::vector<int64_t> &cand_off, std::vector<kg_candidate> &cands,
	                    std::vector<kg_seed> &cand_seeds, const kg_candidate *&cands_out, const kg_seed *&seeds_out) override
	{
		int64_t n = (int64_t)off.size() - 1;
		const size_t n_enc = (size_t)off[(size_t)n];
		std::vector<int64_t> seed_off(off.size(), 0);
		std::vector<ko_seed> buf((size_t)(64 * n + 65536));
		std::vector<uint8_t> codes(n_enc);               // EnCodeReadSeq (src/Mapping.cpp:482-485): nst_nt4_table
		for (size_t i = 0; i < n_enc; ++i) {
			switch (enc[i]) {
			case 'A': case 'a': codes[i] = 0; break;
			case 'C': case 'c': codes[i] = 1; break;
			case 'G': case 'g': codes[i] = 2; break;
			case 'T': case 't': codes[i] = 3; break;
			default: codes[i] = 4;
			}
		}
		int64_t t = ko_seed_batch(ix_, mode, ko_min_seed_len(ix_), codes.data(), off.data(), n, seed_off.data(), buf.data(), (int64_t)buf.size(), 4);
		if (t < 0) {
			buf.resize((size_t)(-t));
			t = ko_seed_batch(ix_, mode, ko_min_seed_len(ix_), codes.data(), off.data(), n, seed_off.data(), buf.data(), (int64_t)buf.size(), 4);
		}
		n_cands.assign((size_t)n + 1, 0);
		cand_off.assign((size_t)n + 1, 0);
		cands.clear(); cand_seeds.clear();
		for (int64_t r = 0; r < n; ++r) {
			cand_off[(size_t)r] = (int64_t)cands.size();
			int ns = (int)(seed_off[(size_t)r + 1] - seed_off[(size_t)r]);
			if (ns == 0) continue;
			const ko_seed *in = buf.data() + seed_off[(size_t)r];
			std::vector<int> coff((size_t)ns + 2), score((size_t)ns + 1);
			std::vector<int64_t> pd((size_t)ns + 1);
			std::vector<ko_pair> pairs((size_t)ns + 1);
			int rlen = (int)(off[(size_t)r + 1] - off[(size_t)r]);
			int nc = pacbio ? ko_candidates_pacbio(ix_, rlen, in, ns, coff.data(), score.data(), pd.data(), pairs.data(), ns + 1, ns + 1)
			                : ko_candidates_illumina(ix_, rlen, max_gaps, in, ns, coff.data(), score.data(), pd.data(), pairs.data(), ns + 1, ns + 1);
			if (nc < 0) { fprintf(stderr, "oracle backend: candidate buffers too small\n"); exit(1); }
			n_cands[(size_t)r] = nc;
			for (int c = 0; c < nc; ++c) {
				kg_candidate o;
				o.posDiff = pd[(size_t)c]; o.score = score[(size_t)c]; o.count = coff[(size_t)c + 1] - coff[(size_t)c]; o.first = (int64_t)cand_seeds.size();
				for (int q = coff[(size_t)c]; q < coff[(size_t)c + 1]; ++q) {
					kg_seed d;
					d.gPos = pairs[(size_t)q].gPos; d.rPos = pairs[(size_t)q].rPos; d.len = pairs[(size_t)q].rLen;
					cand_seeds.push_back(d);
				}
				cands.push_back(o);
			}
		}
		cand_off[(size_t)n] = (int64_t)cands.size();
		cands_out = cands.data();
		seeds_out = cand_seeds.data();
	}
	// The fragment service (kg_fragments_batch in the product) restated through the oracle's GenerateNormalPairAlignment, so that the
	// host's -pacbio path WITH the service (plan -> one batched call -> stitch) is covered without a GPU.  KART_ORACLE_FRAGMENTS=1
	// turns it on; KART_ORACLE_FRAGMENTS=<k> (k > 1) also hands every k-th job back (status 1), as the kernels do outside their envelope.
	bool has_fragments() const override { return getenv("KART_ORACLE_FRAGMENTS") != nullptr; }
	bool fragments_batch(std::vector<FragJobs *> &parts, bool pacbio, int max_gaps) override
	{
		if (!has_fragments()) return false;
		const int back_every = atoi(getenv("KART_ORACLE_FRAGMENTS"));
		const char *text = ko_ref_sequence(ix_);
		// result sets in rotation like the product's page-locked ones (valid for the next few calls)
		FragSet &io = sets_[(size_t)(next_set_.fetch_add(1) % kSets)];
		std::lock_guard<std::mutex> lk(io.mu);
		int64_t n = 0, cols = 0;
		std::vector<int64_t> p_at(parts.size() + 1, 0), p_ac(parts.size() + 1, 0);
		for (size_t k = 0; k < parts.size(); ++k) { p_at[k + 1] = p_at[k] + (int64_t)parts[k]->size(); p_ac[k + 1] = p_ac[k] + parts[k]->cols; }
		n = p_at[parts.size()]; cols = p_ac[parts.size()];
		if (n == 0) return true;
		io.ops.assign((size_t)cols + 64, 0); io.len.assign((size_t)n, 0); io.status.assign((size_t)n, 0);
		std::atomic<size_t> next{0};
		auto work = [&]() {
			std::vector<char> g1, g2;
			for (size_t k; (k = next.fetch_add(1)) < parts.size();) {
				const FragJobs *p = parts[k];
				for (size_t j = 0; j < p->size(); ++j) {
					const int64_t at = p_at[k] + (int64_t)j;
					if (back_every > 1 && at % back_every == back_every - 1) { io.status[(size_t)at] = 1; continue; }
					const int m = (int)(p->o1[j + 1] - p->o1[j]), g_n = p->gl[j];
					g1.resize((size_t)(m + g_n + 2)); g2.resize((size_t)(m + g_n + 2));
					int L = ko_normal_pair_alignment(pacbio ? 1 : 0, max_gaps, p->f1.data() + p->o1[j], m, text + p->g[j], g_n, g1.data(), g2.data());
					ops_from_strings(p->f1.data() + p->o1[j], m, g1.data(), g2.data(), L, io.ops.data() + p_ac[k] + p->oo[j]);
					io.len[(size_t)at] = L;
				}
			}
		};
		std::vector<std::thread> th;
		for (int t = 1; t < 8; ++t) th.emplace_back(work);
		work();
		for (std::thread &x : th) x.join();
		for (size_t k = 0; k < parts.size(); ++k) {
			FragJobs *p = parts[k];
			p->ops = io.ops.data() + p_ac[k]; p->len = io.len.data() + p_at[k]; p->status = io.status.data() + p_at[k];
		}
		return true;
	}
	void nw_batch(std::vector<NwJobs *> &parts) override
	{
		for (NwJobs *p : parts) {
			p->ops.assign(p->f1.size() + p->f2.size() + 1, 0);
			p->len.assign(p->size(), 0);
			for (size_t j = 0; j < p->size(); ++j) {
				int m = (int)(p->o1[j + 1] - p->o1[j]), n = (int)(p->o2[j + 1] - p->o2[j]);
				std::vector<char> g1((size_t)(m + n + 2)), g2((size_t)(m + n + 2));
				int L = ko_nw(p->f1.data() + p->o1[j], m, p->f2.data() + p->o2[j], n, g1.data(), g2.data());
				ops_from_strings(p->f1.data() + p->o1[j], m, g1.data(), g2.data(), L, p->ops.data() + p->o1[j] + p->o2[j]);
				p->len[j] = L;
			}
		}
	}

private:
	ko_index *ix_;
	static constexpr int kSets = 5;
	struct FragSet {
		std::mutex mu;
		std::vector<uint8_t> ops, status;
		std::vector<int32_t> len;
	};
	FragSet sets_[kSets];
	std::atomic<uint64_t> next_set_{0};
};

static KernelBackend *make_oracle_backend(const Options &opt, std::string &err)
{
	ko_index *ix = ko_index_load(opt.index_prefix.c_str());
	if (!ix) { err = "oracle: cannot load index"; return nullptr; }
	return new OracleBackend(ix);
}
}  // namespace kart

int main(int argc, char **argv) { return kart::cli_main(argc, argv, kart::make_oracle_backend); }
