// tests/cpu_backend/ops_selftest.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Pass 2 of the long-read report reads CIGAR runs, identical bases, CheckLocalAlignmentQuality's counts and the head / tail trimming
// off (read fragment, text fragment, op string) -- add_cigar_ops / local_quality_ok_ops / finish_head_ops / finish_tail_ops in
// kart_amd/csrc/host/detail/gap_closing.inc -- where rounds 2-3 re-built the two gapped strings and scanned those (add_cigar /
// local_quality_ok / finish_head / finish_tail, which restate src/tools.cpp:49-104,255-290,314-394 and are kept).  This program
// holds the two forms against each other on random alignments: same CIGAR elements, same score, same trimmed pair.
// Usage: ops_selftest [cases] [seed]; prints the number of cases and exits 1 at the first difference.
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <string_view>
#include <vector>

#include "../../kart_amd/csrc/host/mapper.hpp"

namespace kart {
namespace {
#include "../../kart_amd/csrc/host/detail/types.inc"
#include "../../kart_amd/csrc/host/detail/normal_pairs.inc"
#include "../../kart_amd/csrc/host/detail/kmer.inc"
#include "../../kart_amd/csrc/host/detail/gap_closing.inc"

// the gapped strings an op string stands for (what stitch_frag() writes)
void strings_of(const std::string &a, const std::string &b, const std::vector<uint8_t> &op, std::string &s1, std::string &s2)
{
	s1.clear(); s2.clear();
	size_t i = 0, j = 0;
	for (uint8_t o : op) {
		if (o == KG_OP_DIAG) { s1 += a[i++]; s2 += b[j++]; }
		else if (o == KG_OP_GAP1) { s1 += '-'; s2 += b[j++]; }
		else { s1 += a[i++]; s2 += '-'; }
	}
}

bool same(const CigarVec &x, const CigarVec &y) { return x == y; }

int run(long cases, unsigned seed)
{
	std::mt19937 rng(seed);
	long n_ok = 0, n_trim = 0, n_long = 0;
	const char alphabet[] = "ACGTACGTACGTACGTNacgtRY";      // no '-': a fragment with a literal dash keeps the string path (ops_usable)
	for (long c = 0; c < cases; ++c) {
		// an op string: runs of random kinds and lengths (short ones, and ones longer than the 16-byte steps of the scans)
		std::vector<uint8_t> op;
		const int n_runs = (int)(rng() % 12);
		int m = 0, n = 0;
		uint8_t last = 255;
		for (int r = 0; r < n_runs; ++r) {
			uint8_t k = (uint8_t)(rng() % 3);
			if (rng() % 3 == 0) k = KG_OP_DIAG;
			int len = 1 + (int)(rng() % (rng() % 4 == 0 ? 70 : 6));
			if (k == KG_OP_DIAG) { m += len; n += len; } else if (k == KG_OP_GAP1) n += len; else m += len;
			op.insert(op.end(), (size_t)len, k);
			last = k;
		}
		(void)last;
		std::string a((size_t)m, 'A'), b((size_t)n, 'A');
		const int err = (int)(rng() % 3);                   // identical / a few / many mismatches
		for (char &ch : a) ch = alphabet[rng() % (sizeof(alphabet) - 1)];
		for (char &ch : b) ch = "ACGT"[rng() % 4];
		{   // make the diagonal columns mostly agree, so that both outcomes of the quality check occur
			size_t i = 0, j = 0;
			for (uint8_t o : op) {
				if (o == KG_OP_DIAG) { if (err == 0 || (int)(rng() % 10) >= (err == 1 ? 1 : 5)) a[i] = b[j]; ++i; ++j; }
				else if (o == KG_OP_GAP1) ++j; else ++i;
			}
		}
		if (!ops_usable(a.data(), m)) continue;
		std::string s1, s2;
		strings_of(a, b, op, s1, s2);
		const int L = (int)op.size();
		// AddNewCigarElements
		{
			CigarVec x, y;
			int sx = add_cigar(s1, s2, x), sy = add_cigar_ops(a.data(), b.data(), op.data(), L, y);
			if (sx != sy || !same(x, y)) { fprintf(stderr, "add_cigar differs in case %ld\n", c); return 1; }
		}
		// CheckLocalAlignmentQuality
		n_ok += local_quality_ok(s1, s2);
		n_long += L >= 64;
		if (local_quality_ok(s1, s2) != local_quality_ok_ops(a.data(), b.data(), op.data(), L)) { fprintf(stderr, "local_quality_ok differs in case %ld\n", c); return 1; }
		// ProcessHeadSequencePair / ProcessTailSequencePair after the alignment
		for (int tail = 0; tail < 2; ++tail) {
			Pair p1, p2;
			p1.rPos = p2.rPos = 100; p1.gPos = p2.gPos = 5000; p1.rLen = p2.rLen = m; p1.gLen = p2.gLen = n;
			CigarVec x, y;
			std::string t1 = s1, t2 = s2;
			int sx = tail ? finish_tail(p1, t1, t2, x) : finish_head(p1, t1, t2, x);
			int sy = tail ? finish_tail_ops(p2, a.data(), b.data(), op.data(), L, y) : finish_head_ops(p2, a.data(), b.data(), op.data(), L, y);
			n_trim += p1.rLen != m || p1.gLen != n;
			if (sx != sy || !same(x, y) || p1.rPos != p2.rPos || p1.rLen != p2.rLen || p1.gPos != p2.gPos || p1.gLen != p2.gLen) {
				fprintf(stderr, "%s differs in case %ld\n", tail ? "finish_tail" : "finish_head", c);
				return 1;
			}
		}
	}
	printf("%ld cases identical (quality check passed in %ld, a head or tail trimmed in %ld, %ld alignments of 64 columns or more)\n", cases, n_ok, n_trim, n_long);
	return 0;
}
}  // namespace
}  // namespace kart

int main(int argc, char **argv) { return kart::run(argc > 1 ? atol(argv[1]) : 200000, argc > 2 ? (unsigned)atol(argv[2]) : 1u); }
