// oracle/kart_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see kart_oracle.h).
//
// Plain single-file C++ restatement, on the CPU, of the reference algorithm for
// the Kart hot path.  Every function names the reference file:line it follows
// (paths relative to /root/reference).  It is the checker for the HIP kernels
// and the "port" CPU baseline; it is never part of the product.
//
// Parity status: PINNED.  oracle/pin_against_ref.py drives this library and the
// unmodified reference (oracle/_ref/libkartref.so, built by oracle/Makefile) on
// the same seeded inputs and asserts identical results function by function; the
// inputs/outputs of that run are committed under tests/golden/ so the pin can be
// re-checked where /root/reference does not exist.

#include "kart_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// index image (reference bwt_t / bntseq_t, src/structure.h:31-70)
// ---------------------------------------------------------------------------
struct Contig {
	std::string name;
	int64_t fwd_start;  // Chromosome_t::FowardLocation
	int64_t rev_start;  // Chromosome_t::ReverseLocation
	int64_t len;
};

}  // namespace

struct ko_index {
	uint64_t primary = 0;
	uint64_t L2[5] = {0, 0, 0, 0, 0};
	uint64_t seq_len = 0;
	std::vector<uint32_t> bwt;   // interleaved Occ/BWT words exactly as on disk
	uint64_t sa_intv = 32;
	std::vector<uint64_t> sa;    // sa[0] = (uint64_t)-1
	int64_t l_pac = 0;
	std::vector<Contig> contigs;
	std::map<int64_t, int> chr_end;  // ChrLocMap: last coordinate of each strand copy -> contig
	std::vector<char> ref;           // RefSequence: fwd + revcomp, upper-case ACGT
};

namespace {

thread_local ko_counters tl_cnt = {0, 0, 0, 0, 0, 0, 0};
ko_counters g_cnt = {0, 0, 0, 0, 0, 0, 0};
std::atomic_flag g_cnt_lock = ATOMIC_FLAG_INIT;

void flush_counters()
{
	while (g_cnt_lock.test_and_set(std::memory_order_acquire)) {}
	g_cnt.searches += tl_cnt.searches; g_cnt.lf1 += tl_cnt.lf1; g_cnt.lf2 += tl_cnt.lf2;
	g_cnt.inv += tl_cnt.inv; g_cnt.sa += tl_cnt.sa; g_cnt.seeds += tl_cnt.seeds; g_cnt.bases += tl_cnt.bases;
	g_cnt_lock.clear(std::memory_order_release);
	tl_cnt = ko_counters{0, 0, 0, 0, 0, 0, 0};
}

// nst_nt4_table (src/BWT_Index/bntseq.c:40-57): A/a 0, C/c 1, G/g 2, T/t 3, everything else 4
inline int nt4(unsigned char ch)
{
	switch (ch) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

bool slurp(const std::string &path, std::vector<unsigned char> &buf)
{
	FILE *fp = fopen(path.c_str(), "rb");
	if (!fp) return false;
	fseek(fp, 0, SEEK_END);
	long sz = ftell(fp);
	fseek(fp, 0, SEEK_SET);
	buf.resize((size_t)sz);
	size_t got = sz ? fread(buf.data(), 1, (size_t)sz, fp) : 0;
	fclose(fp);
	return got == (size_t)sz;
}

// ---------------------------------------------------------------------------
// rank / LF / SA   (src/bwt_search.cpp:31-138)
// ---------------------------------------------------------------------------
// One Occ block = 16 u32 words per 128 symbols: words 0..7 are four u64 running
// counts (A,C,G,T before the block), words 8..15 hold 16 symbols each, first
// symbol in bits 31:30 (src/BWT_Index/bwtindex.c:53-75; macros src/bwt_search.cpp:31-33).
inline const uint32_t *block_of(const ko_index &ix, uint64_t k) { return ix.bwt.data() + ((k >> 7) << 4); }

inline int symbol_at(const ko_index &ix, uint64_t k)  // bwt_B0, src/bwt_search.cpp:33
{
	uint32_t w = block_of(ix, k)[8 + ((k & 0x7f) >> 4)];
	return (int)((w >> ((~k & 0xf) << 1)) & 3u);
}

// number of 2-bit fields equal to c in a 64-bit word (src/bwt_search.cpp:35-42)
inline int count_base64(uint64_t y, int c)
{
	y = ((c & 2) ? y : ~y) >> 1 & ((c & 1) ? y : ~y) & 0x5555555555555555ull;
	return __builtin_popcountll(y);
}

// bwt_occ, src/bwt_search.cpp:44-66: occurrences of c in BWT[0..k] ($-less coordinates)
uint64_t occ1(const ko_index &ix, uint64_t k, int c)
{
	if (k == ix.seq_len) return ix.L2[c + 1] - ix.L2[c];
	if (k == (uint64_t)-1) return 0;
	k -= (k >= ix.primary);
	const uint32_t *p = block_of(ix, k);
	uint64_t n;
	memcpy(&n, p + 2 * c, 8);
	p += 8;
	uint64_t in_block = k & 0x7f;              // symbols 0..in_block of this block are counted
	uint64_t full = in_block >> 5;             // complete 32-symbol words
	for (uint64_t i = 0; i < full; ++i, p += 2) n += count_base64((uint64_t)p[0] << 32 | p[1], c);
	uint64_t w = ((uint64_t)p[0] << 32 | p[1]) & ~((1ull << ((~k & 31) << 1)) - 1);
	n += count_base64(w, c);
	if (c == 0) n -= ~k & 31;                  // masked-out fields read as 'A'
	return n;
}

// bwt_occ4, src/bwt_search.cpp:68-85
void occ4(const ko_index &ix, uint64_t k, uint64_t cnt[4])
{
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= ix.primary);
	const uint32_t *p = block_of(ix, k);
	memcpy(cnt, p, 32);
	p += 8;
	uint64_t in_block = k & 0x7f;  // symbols 0..in_block of the block are counted
	uint64_t full = in_block >> 5;
	uint64_t c1 = 0, c2 = 0, c3 = 0;
	for (uint64_t i = 0; i < full; ++i, p += 2) {
		uint64_t w = (uint64_t)p[0] << 32 | p[1];
		c1 += count_base64(w, 1); c2 += count_base64(w, 2); c3 += count_base64(w, 3);
	}
	// fields past k are forced to 0 ('A'), so they cannot disturb the C/G/T counts
	uint64_t w = ((uint64_t)p[0] << 32 | p[1]) & ~((1ull << ((~k & 31) << 1)) - 1);
	c1 += count_base64(w, 1); c2 += count_base64(w, 2); c3 += count_base64(w, 3);
	cnt[0] += in_block + 1 - c1 - c2 - c3;
	cnt[1] += c1; cnt[2] += c2; cnt[3] += c3;
}

// bwt_2occ4, src/bwt_search.cpp:87-118 (only the block-sharing test matters for the
// result; the counters record which branch the reference would take)
void occ4_pair(const ko_index &ix, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
	uint64_t k2 = k - (k >= ix.primary), l2 = l - (l >= ix.primary);
	if ((l2 >> 7) != (k2 >> 7) || k == (uint64_t)-1 || l == (uint64_t)-1) tl_cnt.lf2++;
	else tl_cnt.lf1++;
	occ4(ix, k, ck);
	occ4(ix, l, cl);
}

// bwt_invPsi, src/bwt_search.cpp:120-126
inline uint64_t inv_psi(const ko_index &ix, uint64_t k)
{
	uint64_t x = k - (k > ix.primary);
	int c = symbol_at(ix, x);
	x = ix.L2[c] + occ1(ix, k, c);
	return k == ix.primary ? 0 : x;
}

// bwt_sa, src/bwt_search.cpp:128-138
uint64_t sa_lookup(const ko_index &ix, uint64_t k)
{
	uint64_t steps = 0, mask = ix.sa_intv - 1;
	while (k & mask) {
		++steps;
		k = inv_psi(ix, k);
		tl_cnt.inv++;
	}
	tl_cnt.sa++;
	return steps + ix.sa[k / ix.sa_intv];
}

// BWT_Search, src/bwt_search.cpp:140-184.  Bi-interval {x0 (pattern), x1 (revcomp), size}.
int bwt_search(const ko_index &ix, const uint8_t *seq, int start, int stop, int min_seed_len, int *len_out,
               uint64_t *locs)
{
	const int OCC_THR = 50;  // src/bwt_search.cpp:3
	tl_cnt.searches++;
	int p = seq[start];
	uint64_t x0 = ix.L2[p] + 1, x1 = ix.L2[3 - p] + 1, xs = ix.L2[p + 1] - ix.L2[p];
	int pos;
	for (pos = start + 1; pos < stop; ++pos) {
		if (seq[pos] > 3) break;
		uint64_t tk[4], tl[4];
		occ4_pair(ix, x1 - 1, x1 - 1 + xs, tk, tl);
		uint64_t ok1[4], oks[4], ok0[4];
		for (int i = 0; i < 4; ++i) {
			ok1[i] = ix.L2[i] + 1 + tk[i];
			oks[i] = tl[i] - tk[i];
		}
		ok0[3] = x0 + (x1 <= ix.primary && x1 + xs - 1 >= ix.primary);
		ok0[2] = ok0[3] + oks[3];
		ok0[1] = ok0[2] + oks[2];
		ok0[0] = ok0[1] + oks[1];
		int c = 3 - seq[pos];
		if (oks[c] == 0) break;
		x0 = ok0[c]; x1 = ok1[c]; xs = oks[c];
	}
	int len = pos - start, freq = 0;
	*len_out = len;
	if (len >= min_seed_len) {
		freq = (int)xs;
		if (freq <= OCC_THR) {
			for (int i = 0; i < freq; ++i) locs[i] = sa_lookup(ix, x0 + i);
		} else freq = 0;
	}
	return freq;
}

// comparators, src/AlignmentCandidates.cpp:11-21
inline bool by_posdiff(const ko_seed &a, const ko_seed &b)
{
	int64_t da = a.gPos - a.rPos, db = b.gPos - b.rPos;
	if (da == db) return a.rPos < b.rPos;
	return da < db;
}
inline bool by_gpos(const ko_seed &a, const ko_seed &b)
{
	if (a.gPos == b.gPos) return a.rPos < b.rPos;
	return a.gPos < b.gPos;
}
inline bool pair_by_gpos(const ko_pair &a, const ko_pair &b)
{
	if (a.gPos == b.gPos) return a.rPos < b.rPos;
	return a.gPos < b.gPos;
}

// IdentifySeedPairs_FastMode, src/AlignmentCandidates.cpp:49-80
void seed_fast(const ko_index &ix, int min_seed_len, const uint8_t *enc, int rlen, std::vector<ko_seed> &out)
{
	uint64_t locs[64];
	int pos = 0, end_pos = rlen - min_seed_len;
	while (pos < end_pos) {
		if (enc[pos] > 3) { pos++; continue; }
		int len;
		int freq = bwt_search(ix, enc, pos, rlen, min_seed_len, &len, locs);
		for (int i = 0; i < freq; ++i) out.push_back(ko_seed{(int64_t)locs[i], pos, len});
		pos += len + 1;
	}
	std::sort(out.begin(), out.end(), by_posdiff);
}

// IdentifySeedPairs_SensitiveMode, src/AlignmentCandidates.cpp:132-169
void seed_sensitive(const ko_index &ix, int min_seed_len, const uint8_t *enc, int rlen, std::vector<ko_seed> &out)
{
	uint64_t locs[64];
	int pos = 0, stop_pos = 30, end_pos = rlen - min_seed_len;
	while (pos < end_pos) {
		if (enc[pos] > 3) { pos++; stop_pos++; continue; }
		int len;
		// NB: like the reference, stop_pos may exceed rlen here after an N run;
		// the caller guarantees enc[] has a >3 sentinel or enough room (see ko_seed_read).
		int freq = bwt_search(ix, enc, pos, stop_pos, min_seed_len, &len, locs);
		if (freq > 0) {
			for (int i = 0; i < freq; ++i) out.push_back(ko_seed{(int64_t)locs[i], pos, len});
			pos += len; stop_pos += len;
		} else {
			pos += min_seed_len; stop_pos += min_seed_len;
		}
		if (stop_pos > rlen) stop_pos = rlen;
	}
	std::sort(out.begin(), out.end(), by_gpos);
}

// ---------------------------------------------------------------------------
// nw_alignment, src/nw_alignment.cpp:3-80 (float, half-integer scores)
// ---------------------------------------------------------------------------
int nw_align(const char *s1, int m0, const char *s2, int n0, char *out1, char *out2)
{
	const float MaxPenalty = -65536, OPEN_GAP = -1, EXTEND_GAP = -0.5f, NEW_GAP = -1.5f;
	int m = m0 + 1, n = n0 + 1;
	std::vector<float> R((size_t)m * n), T((size_t)m * n), S((size_t)m * n);
	auto at = [n](int i, int j) { return (size_t)i * n + j; };
	R[0] = T[0] = S[0] = 0;
	for (int i = 1; i < m; ++i) { R[at(i, 0)] = MaxPenalty; S[at(i, 0)] = T[at(i, 0)] = OPEN_GAP + i * EXTEND_GAP; }
	for (int j = 1; j < n; ++j) { T[at(0, j)] = MaxPenalty; S[at(0, j)] = R[at(0, j)] = OPEN_GAP + j * EXTEND_GAP; }
	for (int i = 1; i < m; ++i)
		for (int j = 1; j < n; ++j) {
			float r = std::max(R[at(i, j - 1)] + EXTEND_GAP, S[at(i, j - 1)] + NEW_GAP);
			float t = std::max(T[at(i - 1, j)] + EXTEND_GAP, S[at(i - 1, j)] + NEW_GAP);
			float d = S[at(i - 1, j - 1)] + (nt4((unsigned char)s1[i - 1]) == nt4((unsigned char)s2[j - 1]) ? 1.5f : -1.5f);
			R[at(i, j)] = r; T[at(i, j)] = t;
			S[at(i, j)] = std::max(d, std::max(r, t));
		}
	// traceback (src/nw_alignment.cpp:59-72): r first, then t, then diagonal; built back-to-front
	std::string a1, a2;
	int i = m - 1, j = n - 1;
	while (i > 0 || j > 0) {
		if (S[at(i, j)] == R[at(i, j)]) { a1.push_back('-'); a2.push_back(s2[j - 1]); j--; }
		else if (S[at(i, j)] == T[at(i, j)]) { a1.push_back(s1[i - 1]); a2.push_back('-'); i--; }
		else { a1.push_back(s1[i - 1]); a2.push_back(s2[j - 1]); i--; j--; }
	}
	std::reverse(a1.begin(), a1.end());
	std::reverse(a2.begin(), a2.end());
	memcpy(out1, a1.data(), a1.size()); out1[a1.size()] = 0;
	memcpy(out2, a2.data(), a2.size()); out2[a2.size()] = 0;
	return (int)a1.size();
}

// ---------------------------------------------------------------------------
// chaining (src/AlignmentCandidates.cpp:82-130, 171-224) and normal pairs (:226-490)
// ---------------------------------------------------------------------------
int64_t boundary(const ko_index &ix, int64_t g)  // GetAlignmentBoundary, src/tools.cpp:399-404
{
	auto it = ix.chr_end.lower_bound(g);
	return it->first;
}

ko_pair as_pair(const ko_seed &s)
{
	return ko_pair{s.gPos, s.gPos - s.rPos, s.rPos, s.len, s.len, 1};
}

struct Cand {
	int score;
	int64_t posdiff;
	std::vector<ko_pair> v;
};

// GenerateAlignmentCandidateForIlluminaSeq
void cands_illumina(const ko_index &ix, int rlen, int max_gaps, const ko_seed *s, int num, std::vector<Cand> &out)
{
	int thr = (int)(rlen * 0.2);
	if (thr > 50) thr = 50;
	int i = 0;
	while (i < num && s[i].gPos - s[i].rPos < 0) i++;
	while (i < num) {
		int score = s[i].len;
		int64_t g_end = boundary(ix, s[i].gPos);
		int j = i, k = i + 1;
		for (; k < num; ++k) {
			int64_t dk = s[k].gPos - s[k].rPos, dj = s[j].gPos - s[j].rPos;
			if (s[k].gPos > g_end || dk - dj > max_gaps) break;
			score += s[k].len;
			j = k;
		}
		if (score > thr) {
			Cand c;
			c.score = score;
			for (int q = i; q < k; ++q) c.v.push_back(as_pair(s[q]));
			if (score - 50 > thr) thr = score - 50;
			c.posdiff = c.v[0].PosDiff < 0 ? 0 : c.v[0].PosDiff;
			std::sort(c.v.begin(), c.v.end(), pair_by_gpos);
			out.push_back(std::move(c));
		}
		i = k;
	}
}

// GenerateAlignmentCandidateForPacBioSeq
void cands_pacbio(const ko_index &, int, const ko_seed *s, int num, std::vector<Cand> &out)
{
	if (num <= 0) return;
	int thr = 0;
	std::vector<char> taken((size_t)num, 0);
	int i = 0;
	while (i < num && s[i].gPos - s[i].rPos < 0) i++;
	for (; i < num; ++i) {
		if (taken[i]) continue;
		Cand c;
		c.score = s[i].len;
		taken[i] = 1;
		c.v.push_back(as_pair(s[i]));
		int j = i;
		for (int k = i + 1; k < num; ++k) {
			if (taken[k]) continue;
			int64_t dk = s[k].gPos - s[k].rPos, dj = s[j].gPos - s[j].rPos;
			if (std::llabs(dk - dj) < 300) {
				if (s[k].rPos > s[j].rPos) {
					c.score += s[k].len;
					c.v.push_back(as_pair(s[k]));
					taken[k] = 1;
					j = k;
				}
			} else if (s[k].gPos - s[j].gPos > 1000) break;
		}
		if (c.score >= thr) {
			thr = c.score;
			int64_t d = s[i].gPos - s[i].rPos;
			c.posdiff = d < 0 ? 0 : d;
			out.push_back(std::move(c));
		}
	}
}

void drop_null(std::vector<ko_pair> &v)  // RemoveNullSeeds, :226-233
{
	v.erase(std::remove_if(v.begin(), v.end(), [](const ko_pair &p) { return p.rLen == 0; }), v.end());
}

void drop_tandem(std::vector<ko_pair> &v)  // RemoveTandemRepeatSeeds, :235-260
{
	int num = (int)v.size();
	if (num < 2) return;
	std::map<int, int> mult;
	for (auto &p : v) mult[p.rPos]++;
	bool any = false;
	for (auto &p : v)
		if (mult[p.rPos] > 1) { p.rLen = p.gLen = 0; any = true; }
	if (any) drop_null(v);
}

void drop_translocated(std::vector<ko_pair> &v)  // RemoveTranslocatedSeeds, :262-321
{
	int num = (int)v.size();
	if (num < 2) return;
	std::vector<std::pair<int, int>> ord((size_t)num);
	for (int i = 0; i < num; ++i) ord[i] = {v[i].rPos, i};
	std::stable_sort(ord.begin(), ord.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
	bool any = false;
	for (int i = 0; i < num; ++i) {
		if (ord[i].first == v[i].rPos) continue;
		any = true;
		int hi = ord[i].second;  // IdentifyTranslocationRange
		for (int j = i + 1; j <= hi; ++j)
			if (ord[j].second > hi) hi = ord[j].second;
		int s1 = 0, s2 = 0;
		for (int k = i; k <= hi; ++k) {
			if (k < ord[k].second) s1 += v[ord[k].second].rLen;
			else s2 += v[ord[k].second].rLen;
		}
		for (int k = i; k <= hi; ++k) {
			bool kill = (s1 > s2) ? (k > ord[k].second) : (k < ord[k].second);
			if (kill) v[ord[k].second].rLen = v[ord[k].second].gLen = 0;
		}
		i = hi;
	}
	if (any) drop_null(v);
}

bool trim_overlap(ko_pair &p1, ko_pair &p2)  // CheckSeedOverlapping, :323-373
{
	bool master = true;
	int ov;
	if ((ov = p1.rPos + p1.rLen - p2.rPos) > 0) {
		if (p1.rLen < p2.rLen) {
			master = false;
			if (p1.rLen > ov) p1.gLen = (p1.rLen -= ov);
			else p1.rLen = p1.gLen = 0;
		} else {
			if (p2.rLen > ov) { p2.rPos += ov; p2.gPos += ov; p2.gLen = (p2.rLen -= ov); }
			else p2.rLen = p2.gLen = 0;
		}
	}
	if ((p1.rLen > 0 && p2.rLen > 0) && (ov = (int)(p1.gPos + p1.gLen - p2.gPos)) > 0) {
		if (p1.gLen < p2.gLen) {
			master = false;
			if (p1.rLen > ov) p1.gLen = (p1.rLen -= ov);
			else p1.rLen = p1.gLen = 0;
		} else {
			if (p2.rLen > ov) { p2.rPos += ov; p2.gPos += ov; p2.gLen = (p2.rLen -= ov); }
			else p2.rLen = p2.gLen = 0;
		}
	}
	return master;
}

void resolve_overlaps(std::vector<ko_pair> &v)  // CheckOverlappingSeeds, :375-418
{
	int num = (int)v.size();
	if (num < 2) return;
	bool any = false;
	for (int i = 0; i < num;) {
		if (v[i].rLen > 0) {
			int r_end = v[i].rPos + v[i].rLen - 1;
			int64_t g_end = v[i].gPos + v[i].gLen - 1;
			for (int j = i + 1; j < num; ++j) {
				if (v[j].rLen == 0) continue;
				if (r_end < v[j].rPos && g_end < v[j].gPos) break;
				if (!trim_overlap(v[i], v[j])) break;
			}
			if (v[i].rLen == 0) {
				any = true;
				int q = i - 1;  // LocateThePreviousSeedIdx
				while (q > 0 && v[q].rLen == 0) q--;
				i = q < 0 ? 0 : q;
			} else i++;
		} else { any = true; i++; }
	}
	if (any) drop_null(v);
}

// IdentifyNormalPairs, :420-490
void normal_pairs(int rlen, int glen, std::vector<ko_pair> &v)
{
	ko_pair np;
	memset(&np, 0, sizeof(np));
	if (v.size() > 1) {
		drop_tandem(v);
		drop_translocated(v);
		resolve_overlaps(v);
		int num = (int)v.size();
		for (int i = 0, j = 1; j < num; ++i, ++j) {
			int r_gap = v[j].rPos - (v[i].rPos + v[i].rLen);
			if (r_gap < 0) r_gap = 0;
			int g_gap = (int)(v[j].gPos - (v[i].gPos + v[i].gLen));
			if (g_gap < 0) g_gap = 0;
			if (r_gap > 0 || g_gap > 0) {
				np.bSimple = 0;
				np.rPos = v[i].rPos + v[i].rLen;
				np.gPos = v[i].gPos + v[i].gLen;
				np.PosDiff = np.gPos - np.rPos;
				np.rLen = r_gap; np.gLen = g_gap;
				v.push_back(np);
			}
		}
		if ((int)v.size() > num) std::inplace_merge(v.begin(), v.begin() + num, v.end(), pair_by_gpos);
	}
	if (!v.empty()) {
		int r_gap = v[0].rPos > 0 ? v[0].rPos : 0;
		int g_gap = glen > 0 ? (int)v[0].gPos : r_gap;
		if (r_gap > 0 || g_gap > 0) {
			np.rPos = 0;
			np.gPos = v[0].gPos - g_gap;
			if (np.gPos < 0) np.gPos = 0;  // the reference's "gGaps += gPos" after zeroing is a no-op (:464)
			np.PosDiff = np.gPos;
			np.bSimple = 0;
			np.rLen = r_gap; np.gLen = g_gap;
			v.insert(v.begin(), np);
		}
		size_t last = v.size() - 1;
		r_gap = rlen - (v[last].rPos + v[last].rLen);
		g_gap = glen > 0 ? (int)(glen - (v[last].gPos + v[last].gLen)) : r_gap;
		if (r_gap > 0 || g_gap > 0) {
			np.bSimple = 0;
			np.rPos = v[last].rPos + v[last].rLen;
			np.gPos = v[last].gPos + v[last].gLen;
			np.rLen = r_gap; np.gLen = g_gap;  // PosDiff is stale in the reference (:479-484); unused afterwards
			v.push_back(np);
		}
	}
}

// ---- GenerateNormalPairAlignment and the 8-mer partition behind it --------------------------------------------------
// CreateKmerVecFromReadSeq, src/KmerAnalysis.cpp:56-102 (KmerSize 8, KmerPower 0x3FFF, src/structure.h:23-24): the 8-mers of a
// fragment as (id, position), ids rolled through nst_nt4_table -- so a character other than A/C/G/T/N adds 4 and carries into
// the next base's bits, exactly as there; an 'N' restarts the window, and the character right behind the restarted window is
// never shifted in (the loop header advances both ends once more, :93-98): both quirks are part of the observable behaviour.
struct KmerAt { uint32_t wid, pos; };

uint32_t kmer_id(const char *seq, uint32_t pos)   // CreateKmerID, :28-35
{
	uint32_t id = 0;
	for (uint32_t i = pos; i < pos + 8; ++i) id = (id << 2) + (uint32_t)nt4((unsigned char)seq[i]);
	return id;
}

void kmer_list(int len, const char *seq, std::vector<KmerAt> &out)
{
	out.clear();
	uint32_t tail = 0, count = 0;
	auto window = [&]() {        // advance tail until 8 consecutive non-'N' characters end right before it
		while (count < 8 && tail < (uint32_t)len) {
			if (seq[tail++] != 'N') count++;
			else count = 0;
		}
		return count == 8;
	};
	if (!window()) return;
	uint32_t head = tail - 8;
	KmerAt k{kmer_id(seq, head), head};
	out.push_back(k);
	for (head += 1; tail < (uint32_t)len; ++head, ++tail) {
		if (seq[tail] != 'N') {
			k.pos = head;
			k.wid = ((k.wid & 0x3FFFu) << 2) + (uint32_t)nt4((unsigned char)seq[tail]);
			out.push_back(k);
		} else {
			count = 0;
			tail++;
			if (!window()) break;
			head = tail - 8;
			k.pos = head; k.wid = kmer_id(seq, head);
			out.push_back(k);
		}
	}
	std::stable_sort(out.begin(), out.end(), [](const KmerAt &a, const KmerAt &b) { return a.wid < b.wid; });   // (:100; ties do not reach the result: the pairs are re-sorted below)
}

// GenerateSimplePairsFromFragmentPair, :164-179 = IdentifyCommonKmers (:104-131) + GenerateSimplePairsFromCommonKmers (:133-162, at least 8 long) + sort by genome position
void fragment_simple_pairs(int max_shift, int len1, const char *f1, int len2, const char *f2, std::vector<ko_pair> &out)
{
	out.clear();
	std::vector<KmerAt> v1, v2;
	kmer_list(len1, f1, v1);
	kmer_list(len2, f2, v2);
	struct Common { int diff; uint32_t r, g; };
	std::vector<Common> common;
	for (const KmerAt &a : v1) {
		auto it = std::lower_bound(v2.begin(), v2.end(), a, [](const KmerAt &x, const KmerAt &y) { return x.wid < y.wid; });
		for (; it != v2.end() && it->wid == a.wid; ++it) {
			const bool near = it->pos >= a.pos ? it->pos - a.pos < (uint32_t)max_shift : a.pos - it->pos < (uint32_t)max_shift;    // (:118, unsigned compare as there)
			if (near) common.push_back(Common{(int)(it->pos - a.pos), a.pos, it->pos});
		}
	}
	std::sort(common.begin(), common.end(), [](const Common &a, const Common &b) { return a.diff == b.diff ? a.r < b.r : a.diff < b.diff; });
	for (size_t i = 0; i < common.size();) {
		size_t j = i + 1;
		for (uint32_t next = common[i].r + 1; j < common.size() && common[j].r == next && common[j].diff == common[i].diff; ++j) next++;
		const int l = 8 + (int)(j - 1 - i);
		if (l >= 8) {
			ko_pair p;
			memset(&p, 0, sizeof(p));
			p.bSimple = 1; p.rPos = (int32_t)common[i].r; p.gPos = common[i].g; p.PosDiff = common[i].diff; p.rLen = p.gLen = l;
			out.push_back(p);
		}
		i = j;
	}
	std::sort(out.begin(), out.end(), pair_by_gpos);
}

// GenerateNormalPairAlignment, src/tools.cpp:142-223: both strings are replaced by their aligned forms
void normal_pair_alignment(bool pacbio, int max_gaps, std::string &frag1, std::string &frag2)
{
	const int rlen = (int)frag1.size(), glen = (int)frag2.size();
	bool run_nw = true;
	if (rlen > 30 && glen > 30) {
		int max_shift;
		if (pacbio) {
			max_shift = rlen > glen ? (int)(rlen * 0.2) : (int)(glen * 0.2);
			if (max_shift > 50) max_shift = 50;
		} else max_shift = max_gaps;
		std::vector<ko_pair> part;
		fragment_simple_pairs(max_shift, rlen, frag1.c_str(), glen, frag2.c_str(), part);
		if (!part.empty()) normal_pairs(rlen, glen, part);
		if (!part.empty()) {
			run_nw = false;
			std::string a1, a2;
			for (const ko_pair &p : part) {
				if (p.rLen <= 0 && p.gLen <= 0) continue;
				if (p.gLen == 0) { a1 += frag1.substr((size_t)p.rPos, (size_t)p.rLen); a2.append((size_t)p.rLen, '-'); }
				else if (p.rLen == 0) { a1.append((size_t)p.gLen, '-'); a2 += frag2.substr((size_t)p.gPos, (size_t)p.gLen); }
				else {
					std::string s1 = frag1.substr((size_t)p.rPos, (size_t)p.rLen), s2 = frag2.substr((size_t)p.gPos, (size_t)p.gLen);
					if (!(p.rLen == 1 && p.gLen == 1) && !p.bSimple) {
						if (pacbio && (p.rLen > 300 || p.gLen > 300)) normal_pair_alignment(pacbio, max_gaps, s1, s2);   // :197
						else {
							std::vector<char> o1((size_t)(p.rLen + p.gLen) + 1), o2((size_t)(p.rLen + p.gLen) + 1);
							int l = nw_align(s1.data(), p.rLen, s2.data(), p.gLen, o1.data(), o2.data());
							s1.assign(o1.data(), (size_t)l); s2.assign(o2.data(), (size_t)l);
						}
					}
					a1 += s1; a2 += s2;
				}
			}
			frag1.swap(a1); frag2.swap(a2);
		}
	}
	if (run_nw) {
		std::vector<char> o1((size_t)(rlen + glen) + 1), o2((size_t)(rlen + glen) + 1);
		int l = nw_align(frag1.data(), rlen, frag2.data(), glen, o1.data(), o2.data());
		frag1.assign(o1.data(), (size_t)l); frag2.assign(o2.data(), (size_t)l);
	}
}

int emit_cands(std::vector<Cand> &cands, int *cand_off, int *score, int64_t *posdiff, ko_pair *out_pairs, int cand_cap,
               int pair_cap)
{
	int total = 0;
	for (auto &c : cands) total += (int)c.v.size();
	if ((int)cands.size() > cand_cap || total > pair_cap) return -1;
	int off = 0;
	for (size_t i = 0; i < cands.size(); ++i) {
		cand_off[i] = off;
		score[i] = cands[i].score;
		posdiff[i] = cands[i].posdiff;
		for (auto &p : cands[i].v) out_pairs[off++] = p;
	}
	cand_off[cands.size()] = off;
	return (int)cands.size();
}

}  // namespace

// ---------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------
extern "C" {

// bwa_idx_load + RestoreReferenceInfo, src/bwt_index.cpp:16-36,47-71,103-122,194-259
ko_index *ko_index_load(const char *prefix)
{
	std::string pre(prefix);
	std::vector<unsigned char> buf;
	ko_index *ix = new ko_index();
	if (!slurp(pre + ".bwt", buf) || buf.size() < 40) { delete ix; return nullptr; }
	memcpy(&ix->primary, buf.data(), 8);
	memcpy(&ix->L2[1], buf.data() + 8, 32);
	ix->seq_len = ix->L2[4];
	ix->bwt.resize((buf.size() - 40) / 4);
	memcpy(ix->bwt.data(), buf.data() + 40, ix->bwt.size() * 4);

	if (!slurp(pre + ".sa", buf) || buf.size() < 56) { delete ix; return nullptr; }
	memcpy(&ix->sa_intv, buf.data() + 40, 8);
	uint64_t n_sa = (ix->seq_len + ix->sa_intv) / ix->sa_intv;
	ix->sa.assign(n_sa, 0);
	ix->sa[0] = (uint64_t)-1;
	size_t avail = (buf.size() - 56) / 8;
	memcpy(ix->sa.data() + 1, buf.data() + 56, std::min<size_t>(avail, n_sa - 1) * 8);

	FILE *fp = fopen((pre + ".ann").c_str(), "r");
	if (!fp) { delete ix; return nullptr; }
	long long l_pac; int n_seqs; unsigned seed;
	if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(fp); delete ix; return nullptr; }
	ix->l_pac = l_pac;
	int64_t total = 0;
	for (int i = 0; i < n_seqs; ++i) {
		unsigned gi; char name[1024];
		if (fscanf(fp, "%u%1023s", &gi, name) != 2) break;
		int ch;
		while ((ch = fgetc(fp)) != '\n' && ch != EOF) {}
		long long off; int len, n_ambs;
		if (fscanf(fp, "%lld%d%d", &off, &len, &n_ambs) != 3) break;
		Contig c;
		c.name = name; c.len = len;
		c.fwd_start = total; total += len;
		c.rev_start = 2 * ix->l_pac - total;
		ix->chr_end[c.fwd_start + c.len - 1] = i;
		ix->chr_end[c.rev_start + c.len - 1] = i;
		ix->contigs.push_back(c);
	}
	fclose(fp);

	if (!slurp(pre + ".pac", buf)) { delete ix; return nullptr; }
	int64_t L = ix->l_pac;
	ix->ref.assign((size_t)(2 * L + 1), 0);
	static const char fw[4] = {'A', 'C', 'G', 'T'}, rc[4] = {'T', 'G', 'C', 'A'};
	for (int64_t f = 0; f < L; ++f) {
		int b = buf[(size_t)(f >> 2)] >> ((~f & 3) << 1) & 3;
		ix->ref[(size_t)f] = fw[b];
		ix->ref[(size_t)(2 * L - 1 - f)] = rc[b];
	}
	return ix;
}

void ko_index_free(ko_index *ix) { delete ix; }
int64_t ko_genome_size(const ko_index *ix) { return ix->l_pac; }
uint64_t ko_seq_len(const ko_index *ix) { return ix->seq_len; }
uint64_t ko_primary(const ko_index *ix) { return ix->primary; }
int ko_n_contigs(const ko_index *ix) { return (int)ix->contigs.size(); }
const char *ko_ref_sequence(const ko_index *ix) { return ix->ref.data(); }

int ko_min_seed_len(const ko_index *ix)  // src/Mapping.cpp:645
{
	int k;
	double two_l = (double)(2 * ix->l_pac);
	for (k = 13; k < 16; ++k)
		if (two_l < std::pow(4.0, k)) break;
	return k;
}

uint64_t ko_occ(const ko_index *ix, uint64_t k, int c) { return occ1(*ix, k, c); }
void ko_occ4(const ko_index *ix, uint64_t k, uint64_t cnt[4]) { occ4(*ix, k, cnt); }
uint64_t ko_sa(const ko_index *ix, uint64_t k) { return sa_lookup(*ix, k); }

int ko_bwt_search(const ko_index *ix, const uint8_t *seq, int start, int stop, int min_seed_len, int *len, uint64_t *locs)
{
	return bwt_search(*ix, seq, start, stop, min_seed_len, len, locs);
}

int ko_seed_read(const ko_index *ix, int mode, int min_seed_len, const uint8_t *enc, int rlen, ko_seed *out, int cap)
{
	std::vector<ko_seed> v;
	// SensitiveMode may look up to 30 positions past rlen after an N run (App. B-10): give it a padded copy
	std::vector<uint8_t> padded(enc, enc + rlen);
	padded.resize((size_t)rlen + 64, 4);
	if (mode == 0) seed_fast(*ix, min_seed_len, padded.data(), rlen, v);
	else seed_sensitive(*ix, min_seed_len, padded.data(), rlen, v);
	tl_cnt.seeds += v.size();
	tl_cnt.bases += (uint64_t)rlen;
	flush_counters();
	if ((int)v.size() > cap) return -(int)v.size();
	std::copy(v.begin(), v.end(), out);
	return (int)v.size();
}

int64_t ko_seed_batch(const ko_index *ix, int mode, int min_seed_len, const uint8_t *enc, const int64_t *offsets,
                      int64_t n_reads, int64_t *seed_offsets, ko_seed *out, int64_t cap, int threads)
{
	if (threads < 1) threads = 1;
	std::vector<std::vector<ko_seed>> per_read((size_t)n_reads);
	auto work = [&](int64_t lo, int64_t hi) {
		std::vector<uint8_t> padded;
		for (int64_t r = lo; r < hi; ++r) {
			int rlen = (int)(offsets[r + 1] - offsets[r]);
			padded.assign(enc + offsets[r], enc + offsets[r + 1]);
			padded.resize((size_t)rlen + 64, 4);
			if (mode == 0) seed_fast(*ix, min_seed_len, padded.data(), rlen, per_read[(size_t)r]);
			else seed_sensitive(*ix, min_seed_len, padded.data(), rlen, per_read[(size_t)r]);
			tl_cnt.seeds += per_read[(size_t)r].size();
			tl_cnt.bases += (uint64_t)rlen;
		}
		flush_counters();
	};
	if (threads == 1) work(0, n_reads);
	else {
		std::vector<std::thread> pool;
		int64_t per = (n_reads + threads - 1) / threads;
		for (int t = 0; t < threads; ++t) {
			int64_t lo = t * per, hi = std::min<int64_t>(n_reads, lo + per);
			if (lo < hi) pool.emplace_back(work, lo, hi);
		}
		for (auto &th : pool) th.join();
	}
	int64_t total = 0;
	for (int64_t r = 0; r < n_reads; ++r) {
		seed_offsets[r] = total;
		total += (int64_t)per_read[(size_t)r].size();
	}
	seed_offsets[n_reads] = total;
	if (total > cap) return -total;
	for (int64_t r = 0; r < n_reads; ++r)
		std::copy(per_read[(size_t)r].begin(), per_read[(size_t)r].end(), out + seed_offsets[r]);
	return total;
}

void ko_counters_get(ko_counters *c) { *c = g_cnt; }
void ko_counters_reset(void) { g_cnt = ko_counters{0, 0, 0, 0, 0, 0, 0}; }

int ko_nw(const char *s1, int m, const char *s2, int n, char *out1, char *out2) { return nw_align(s1, m, s2, n, out1, out2); }

int64_t ko_alignment_boundary(const ko_index *ix, int64_t gPos) { return boundary(*ix, gPos); }

int ko_candidates_illumina(const ko_index *ix, int rlen, int max_gaps, const ko_seed *seeds, int n, int *cand_off,
                           int *score, int64_t *posdiff, ko_pair *out_pairs, int cand_cap, int pair_cap)
{
	std::vector<Cand> c;
	cands_illumina(*ix, rlen, max_gaps, seeds, n, c);
	return emit_cands(c, cand_off, score, posdiff, out_pairs, cand_cap, pair_cap);
}

int ko_candidates_pacbio(const ko_index *ix, int rlen, const ko_seed *seeds, int n, int *cand_off, int *score,
                         int64_t *posdiff, ko_pair *out_pairs, int cand_cap, int pair_cap)
{
	std::vector<Cand> c;
	cands_pacbio(*ix, rlen, seeds, n, c);
	return emit_cands(c, cand_off, score, posdiff, out_pairs, cand_cap, pair_cap);
}

int ko_normal_pair_alignment(int pacbio, int max_gaps, const char *s1, int m, const char *s2, int n, char *out1, char *out2)
{
	std::string a(s1, (size_t)m), b(s2, (size_t)n);
	normal_pair_alignment(pacbio != 0, max_gaps, a, b);
	memcpy(out1, a.c_str(), a.size() + 1);
	memcpy(out2, b.c_str(), b.size() + 1);
	return (int)a.size();
}

int ko_identify_normal_pairs(int rlen, int glen, ko_pair *pairs, int n, int cap)
{
	std::vector<ko_pair> v(pairs, pairs + n);
	normal_pairs(rlen, glen, v);
	if ((int)v.size() > cap) return -(int)v.size();
	std::copy(v.begin(), v.end(), pairs);
	return (int)v.size();
}

}  // extern "C"
