// mapper.cpp -- host pipeline: reads in, SAM out, kernels through KernelBackend (see mapper.hpp).
//
// Every block names the reference code whose observable behaviour it reproduces (paths relative
// to the reference tree).  The aim is byte-identical output to `kart -t 1`; where the reference
// has undefined behaviour the choice made here is stated next to the code.
#include "mapper.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <climits>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <malloc.h>
#include <memory>
#include <mutex>
#include <string_view>
#include <thread>

namespace kart {

namespace {

// ----------------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------------
inline int nt4(unsigned char ch)  // nst_nt4_table, src/BWT_Index/bntseq.c:40-57
{
	switch (ch) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

// GetComplementaryBase, src/tools.cpp:3-17, as a table (it runs over every base of every mate 2 and of every reverse-strand
// record): A/a -> T, C/c -> G, G/g -> C, T/t -> A, anything else -> N
struct CompTable {
	char t[256];
	constexpr CompTable() : t()
	{
		for (int i = 0; i < 256; ++i) t[i] = 'N';
		t['A'] = t['a'] = 'T'; t['C'] = t['c'] = 'G'; t['G'] = t['g'] = 'C'; t['T'] = t['t'] = 'A';
	}
};
constexpr CompTable kComp;
inline char comp_base(char c) { return kComp.t[(unsigned char)c]; }

std::string revcomp(std::string_view s)  // GetComplementarySeq, src/tools.cpp:19-29
{
	std::string r(s.size(), 'N');
	for (size_t i = 0, n = s.size(); i < n; ++i) r[i] = comp_base(s[n - 1 - i]);
	return r;
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

// ----------------------------------------------------------------------------------------------
// data carried per read (ReadItem_t, SeedPair_t, AlignmentCandidate_t, AlignmentReport_t;
// src/structure.h:106-154)
// ----------------------------------------------------------------------------------------------
struct Pair {
	bool simple;
	int rPos;
	int64_t gPos;
	int rLen, gLen;
	int64_t posDiff;
};

struct Candidate {
	int score = 0;
	int64_t posDiff = 0;
	int mate = -1;   // PairedAlnCanIdx
	std::vector<Pair> pairs;
};

struct Report {
	int score = 0;          // AlnScore
	int flag = 0;           // SamFlag (the reference leaves it uninitialised where it never sets it; here 0)
	int mate = -1;          // PairedAlnCanIdx
	bool fwd = true;        // coor.bDir
	std::string cigar;
	int64_t gPos = 0;
	int chr = 0;            // coor.ChromosomeIdx (uninitialised in the reference when never assigned; here 0)
};

struct Read {
	// views into the mapped input file, or into the batch's own storage (reverse-complemented mates, the
	// getline()/gzgets() readers); no per-read allocation
	std::string_view name, seq, qual;
	int rlen = 0;
	int mapq = 0, score = 0, sub_score = 0, can_num = 0, best = 0;
	std::vector<Report> rep;
};

typedef std::vector<std::pair<int, char>> CigarVec;

bool pair_by_gpos(const Pair &a, const Pair &b)  // CompByGenomePos, src/AlignmentCandidates.cpp:17-21
{
	if (a.gPos == b.gPos) return a.rPos < b.rPos;
	return a.gPos < b.gPos;
}

struct Ctx {
	const Options &opt;
	const RefData &ref;
	KernelBackend &kern;
	int min_seed_len;
	bool fastq = true;
	const char *refseq() const { return ref.seq.get(); }
};

// What one 4000-read chunk contributes to the run-wide pairing statistics (iPaired / iDistance,
// src/Mapping.cpp:13,20,209-213), and for which EstDistance values its pairing decisions hold.
struct PairStats {
	int64_t paired = 0, distance = 0;
	// every "dist < EstiDistance" test of CheckPairedAlignmentCandidates (:372) narrows the interval
	// (lo, hi] of EstDistance values that would have produced the same outcome
	int64_t lo = -1, hi = INT64_MAX;
	bool rescue_used = false;      // rescue windows depend on min(EstDistance, MaxInsertSize)
};

// ----------------------------------------------------------------------------------------------
// chaining (GenerateAlignmentCandidateForIlluminaSeq / ForPacBioSeq, src/AlignmentCandidates.cpp:82-130,
// 171-224) runs on the device: KernelBackend::candidates_batch -> kg_candidates_batch (chain_kernel in
// seed_kernels.hip); chunk_stage_a unpacks its output with from_seed()
// ----------------------------------------------------------------------------------------------
Pair from_seed(const kg_seed &s)
{
	Pair p;
	p.simple = true; p.rPos = s.rPos; p.gPos = s.gPos; p.rLen = p.gLen = s.len; p.posDiff = s.gPos - s.rPos;
	return p;
}

// ----------------------------------------------------------------------------------------------
// normal pairs: IdentifyNormalPairs and its helpers (src/AlignmentCandidates.cpp:226-490)
// ----------------------------------------------------------------------------------------------
void erase_empty(std::vector<Pair> &v)
{
	v.erase(std::remove_if(v.begin(), v.end(), [](const Pair &p) { return p.rLen == 0; }), v.end());
}

void remove_tandem_repeats(std::vector<Pair> &v)  // :235-260 -- every read position hit more than once goes
{
	int num = (int)v.size();
	if (num < 2) return;
	std::vector<std::pair<int, int>> byr((size_t)num);
	for (int i = 0; i < num; ++i) byr[i] = std::make_pair(v[i].rPos, i);
	std::sort(byr.begin(), byr.end());
	bool any = false;
	for (int i = 0; i < num;) {
		int j = i + 1;
		while (j < num && byr[j].first == byr[i].first) j++;
		if (j - i > 1) {
			any = true;
			for (int k = i; k < j; ++k) v[byr[k].second].rLen = v[byr[k].second].gLen = 0;
		}
		i = j;
	}
	if (any) erase_empty(v);
}

void remove_translocated(std::vector<Pair> &v)  // :262-321
{
	int num = (int)v.size();
	if (num < 2) return;
	std::vector<std::pair<int, int>> byr((size_t)num);
	for (int i = 0; i < num; ++i) byr[i] = std::make_pair(v[i].rPos, i);
	std::sort(byr.begin(), byr.end());   // read positions are distinct here (tandem repeats already removed)
	bool any = false;
	for (int i = 0; i < num; ++i) {
		if (byr[i].first == v[i].rPos) continue;
		any = true;
		int hi = byr[i].second;
		for (int j = i + 1; j <= hi; ++j)
			if (byr[j].second > hi) hi = byr[j].second;
		int s1 = 0, s2 = 0;
		for (int k = i; k <= hi; ++k) {
			if (k < byr[k].second) s1 += v[byr[k].second].rLen;
			else s2 += v[byr[k].second].rLen;
		}
		for (int k = i; k <= hi; ++k) {
			bool drop = s1 > s2 ? k > byr[k].second : k < byr[k].second;
			if (drop) v[byr[k].second].rLen = v[byr[k].second].gLen = 0;
		}
		i = hi;
	}
	if (any) erase_empty(v);
}

bool resolve_overlap(Pair &p1, Pair &p2)  // CheckSeedOverlapping, :323-373
{
	bool master = true;
	int ov;
	if ((ov = p1.rPos + p1.rLen - p2.rPos) > 0) {
		if (p1.rLen < p2.rLen) {
			master = false;
			if (p1.rLen > ov) p1.gLen = (p1.rLen -= ov);
			else p1.rLen = p1.gLen = 0;
		} else if (p2.rLen > ov) {
			p2.rPos += ov; p2.gPos += ov; p2.gLen = (p2.rLen -= ov);
		} else p2.rLen = p2.gLen = 0;
	}
	if (p1.rLen > 0 && p2.rLen > 0 && (ov = (int)(p1.gPos + p1.gLen - p2.gPos)) > 0) {
		if (p1.gLen < p2.gLen) {
			master = false;
			if (p1.rLen > ov) p1.gLen = (p1.rLen -= ov);
			else p1.rLen = p1.gLen = 0;
		} else if (p2.rLen > ov) {
			p2.rPos += ov; p2.gPos += ov; p2.gLen = (p2.rLen -= ov);
		} else p2.rLen = p2.gLen = 0;
	}
	return master;
}

void check_overlaps(std::vector<Pair> &v)  // CheckOverlappingSeeds, :375-418
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
				if (!resolve_overlap(v[i], v[j])) break;
			}
			if (v[i].rLen == 0) {
				any = true;
				int q = i - 1;
				while (q > 0 && v[q].rLen == 0) q--;
				i = q < 0 ? 0 : q;
			} else i++;
		} else {
			any = true;
			i++;
		}
	}
	if (any) erase_empty(v);
}

void identify_normal_pairs(int rlen, int glen, std::vector<Pair> &v)  // :420-490
{
	Pair np;
	np.simple = false; np.rPos = 0; np.gPos = 0; np.rLen = np.gLen = 0; np.posDiff = 0;
	if (v.size() > 1) {
		remove_tandem_repeats(v);
		remove_translocated(v);
		check_overlaps(v);
		int num = (int)v.size();
		for (int i = 0, j = 1; j < num; ++i, ++j) {
			int r_gap = v[j].rPos - (v[i].rPos + v[i].rLen);
			if (r_gap < 0) r_gap = 0;
			int g_gap = (int)(v[j].gPos - (v[i].gPos + v[i].gLen));
			if (g_gap < 0) g_gap = 0;
			if (r_gap > 0 || g_gap > 0) {
				np.simple = false;
				np.rPos = v[i].rPos + v[i].rLen;
				np.gPos = v[i].gPos + v[i].gLen;
				np.posDiff = np.gPos - np.rPos;
				np.rLen = r_gap; np.gLen = g_gap;
				v.push_back(np);
			}
		}
		if ((int)v.size() > num) {
			// the appended gap pairs go between the seeds in (gPos, rPos) order (keys are distinct); a handful per read:
			// insert them by hand, std::inplace_merge allocates a scratch buffer every time
			if (v.size() - (size_t)num <= 8) {
				for (size_t t = (size_t)num; t < v.size(); ++t) {
					Pair x = v[t];
					size_t p = t;
					while (p > 0 && pair_by_gpos(x, v[p - 1])) { v[p] = v[p - 1]; --p; }
					v[p] = x;
				}
			} else std::inplace_merge(v.begin(), v.begin() + num, v.end(), pair_by_gpos);
		}
	}
	if (!v.empty()) {
		int r_gap = v[0].rPos > 0 ? v[0].rPos : 0;
		int g_gap = glen > 0 ? (int)v[0].gPos : r_gap;
		if (r_gap > 0 || g_gap > 0) {
			np.rPos = 0;
			np.gPos = v[0].gPos - g_gap;
			if (np.gPos < 0) np.gPos = 0;          // the reference's follow-up "gGaps += gPos" adds zero (:464)
			np.posDiff = np.gPos;
			np.simple = false;
			np.rLen = r_gap; np.gLen = g_gap;
			v.insert(v.begin(), np);
		}
		size_t last = v.size() - 1;
		r_gap = rlen - (v[last].rPos + v[last].rLen);
		g_gap = glen > 0 ? (int)(glen - (v[last].gPos + v[last].gLen)) : r_gap;
		if (r_gap > 0 || g_gap > 0) {
			np.simple = false;
			np.rPos = v[last].rPos + v[last].rLen;
			np.gPos = v[last].gPos + v[last].gLen;
			np.rLen = r_gap; np.gLen = g_gap;
			v.push_back(np);
		}
	}
}

// ----------------------------------------------------------------------------------------------
// 8-mer matcher (src/KmerAnalysis.cpp)
// ----------------------------------------------------------------------------------------------
struct Kmer { uint32_t wid, pos; };
struct KmerHit { int posDiff; uint32_t rPos, gPos; };

uint32_t kmer_id(const char *seq, uint32_t pos)  // CreateKmerID, :25-32
{
	uint32_t id = 0;
	for (uint32_t i = pos; i < pos + 8; ++i) id = (id << 2) + (uint32_t)nt4((unsigned char)seq[i]);
	return id;
}

void kmers_of(int len, const char *seq, std::vector<Kmer> &vec, bool sorted = true)  // CreateKmerVecFromReadSeq, :56-102
{
	vec.clear();
	uint32_t count = 0, head, tail = 0, ulen = (uint32_t)(len < 0 ? 0 : len);
	while (count < 8 && tail < ulen) {
		if (seq[tail++] != 'N') count++;
		else count = 0;
	}
	if (count != 8) return;
	Kmer km;
	km.pos = (head = tail - 8);
	km.wid = kmer_id(seq, head);
	vec.push_back(km);
	for (head += 1; tail < ulen; head++, tail++) {
		if (seq[tail] != 'N') {
			km.pos = head;
			km.wid = ((km.wid & 0x3FFF) << 2) + (uint32_t)nt4((unsigned char)seq[tail]);
			vec.push_back(km);
		} else {
			count = 0;
			tail++;
			while (count < 8 && tail < ulen) {
				if (seq[tail++] != 'N') count++;
				else count = 0;
			}
			if (count != 8) break;
			km.pos = (head = tail - 8);
			km.wid = kmer_id(seq, head);
			vec.push_back(km);
		}
	}
	if (sorted) std::sort(vec.begin(), vec.end(), [](const Kmer &a, const Kmer &b) { return a.wid < b.wid; });
}

void common_kmers(int max_shift, const std::vector<Kmer> &v1, const std::vector<Kmer> &v2, std::vector<KmerHit> &out)  // :104-130
{
	out.clear();
	for (size_t i = 0; i < v1.size(); ++i) {
		uint32_t wid = v1[i].wid;
		std::vector<Kmer>::const_iterator it =
		    std::lower_bound(v2.begin(), v2.end(), v1[i], [](const Kmer &a, const Kmer &b) { return a.wid < b.wid; });
		for (; it != v2.end() && it->wid == wid; ++it) {
			if ((it->pos >= v1[i].pos && it->pos - v1[i].pos < (uint32_t)max_shift) ||
			    (it->pos < v1[i].pos && v1[i].pos - it->pos < (uint32_t)max_shift)) {
				KmerHit h;
				h.rPos = v1[i].pos;
				h.gPos = it->pos;
				h.posDiff = (int)(h.gPos - h.rPos);
				out.push_back(h);
			}
		}
	}
	std::sort(out.begin(), out.end(), [](const KmerHit &a, const KmerHit &b) {
		if (a.posDiff == b.posDiff) return a.rPos < b.rPos;
		return a.posDiff < b.posDiff;
	});
}

// Mate rescue joins the 8-mers of one read with those of many reference windows (src/AlignmentRescue.cpp:108-111,
// 142-145 -> CreateKmerVecFromReadSeq + IdentifyCommonKmers).  The reference sorts every window's k-mers to
// binary-search them; the join itself is just "all (rPos, gPos) with equal k-mers", and the hit list is sorted by a
// total order afterwards, so here the READ's k-mers go into a direct-address table once and each window is streamed
// past it: same hits, no per-window sort.
struct KmerTable {
	std::vector<uint16_t> first;     // k-mer id -> 1 + index of the last read k-mer with that id (0 = none)
	std::vector<uint16_t> next;      // read k-mer -> 1 + index of the previous one with the same id
	// ids are sums of eight codes 0..4 (a lower-case 'n' is not skipped like 'N' and encodes as 4, as in the reference):
	// up to 4 * (4^8 - 1) / 3 = 87380, so 2^17 slots
	KmerTable() : first((size_t)1 << 17, 0) {}
	void set(const std::vector<Kmer> &kr)      // kr in any order
	{
		if (kr.size() >= 65535) return;
		next.resize(kr.size());
		for (size_t i = 0; i < kr.size(); ++i) {
			next[i] = first[kr[i].wid];
			first[kr[i].wid] = (uint16_t)(i + 1);
		}
	}
	void clear(const std::vector<Kmer> &kr)
	{
		if (kr.size() >= 65535) return;
		for (size_t i = 0; i < kr.size(); ++i) first[kr[i].wid] = 0;
	}
};

void window_hits(int max_shift, const std::vector<Kmer> &kr, const KmerTable &tab, int len, const char *seq, std::vector<KmerHit> &out)
{
	out.clear();
	auto probe = [&](uint32_t wid, uint32_t gpos) {
		for (size_t i = tab.first[wid]; i != 0; i = tab.next[i - 1]) {
			uint32_t rpos = kr[i - 1].pos;
			if ((gpos >= rpos && gpos - rpos < (uint32_t)max_shift) || (gpos < rpos && rpos - gpos < (uint32_t)max_shift)) {
				KmerHit h;
				h.rPos = rpos;
				h.gPos = gpos;
				h.posDiff = (int)(gpos - rpos);
				out.push_back(h);
			}
		}
	};
	// the window's k-mers in position order, with CreateKmerVecFromReadSeq's treatment of 'N' (see kmers_of)
	uint32_t count = 0, head, tail = 0, ulen = (uint32_t)(len < 0 ? 0 : len);
	while (count < 8 && tail < ulen) {
		if (seq[tail++] != 'N') count++;
		else count = 0;
	}
	if (count == 8) {
		uint32_t wid = kmer_id(seq, head = tail - 8);
		probe(wid, head);
		for (head += 1; tail < ulen; head++, tail++) {
			if (seq[tail] != 'N') {
				wid = ((wid & 0x3FFF) << 2) + (uint32_t)nt4((unsigned char)seq[tail]);
				probe(wid, head);
			} else {
				count = 0;
				tail++;
				while (count < 8 && tail < ulen) {
					if (seq[tail++] != 'N') count++;
					else count = 0;
				}
				if (count != 8) break;
				wid = kmer_id(seq, head = tail - 8);
				probe(wid, head);
			}
		}
	}
	std::sort(out.begin(), out.end(), [](const KmerHit &a, const KmerHit &b) {
		if (a.posDiff == b.posDiff) return a.rPos < b.rPos;
		return a.posDiff < b.posDiff;
	});
}

void simple_pairs_from_kmers(int min_len, const std::vector<KmerHit> &hits, std::vector<Pair> &out)  // :132-162
{
	out.clear();
	int num = (int)hits.size();
	for (int i = 0; i < num;) {
		int pd = hits[i].posDiff, j;
		uint32_t next = hits[i].rPos + 1;
		for (j = i + 1; j < num; ++j) {
			if (hits[j].rPos != next || hits[j].posDiff != pd) break;
			next++;
		}
		int l = 8 + (j - 1 - i);
		if (l >= min_len) {
			Pair p;
			p.simple = true;
			p.rPos = (int)hits[i].rPos;
			p.gPos = hits[i].gPos;
			p.posDiff = hits[i].posDiff;
			p.rLen = p.gLen = l;
			out.push_back(p);
		}
		i = j;
	}
}

void simple_pairs_from_fragments(int max_dist, int len1, const char *f1, int len2, const char *f2, std::vector<Pair> &out)  // :164-179
{
	static thread_local std::vector<Kmer> k1;
	static thread_local std::vector<KmerHit> hits;
	static thread_local KmerTable tab;
	kmers_of(len1, f1, k1, false);
	if (k1.size() < 65535) {         // same hits as the reference's two sorted vectors + binary search (see window_hits)
		tab.set(k1);
		window_hits(max_dist, k1, tab, len2, f2, hits);
		tab.clear(k1);
	} else {
		std::vector<Kmer> k2;
		std::sort(k1.begin(), k1.end(), [](const Kmer &a, const Kmer &b) { return a.wid < b.wid; });
		kmers_of(len2, f2, k2);
		common_kmers(max_dist, k1, k2, hits);
	}
	simple_pairs_from_kmers(8, hits, out);
	std::sort(out.begin(), out.end(), pair_by_gpos);
}

// ----------------------------------------------------------------------------------------------
// gap closing, two passes around one batched NW call
// (GenerateNormalPairAlignment / Process{Head,Normal,Tail}SequencePair, src/tools.cpp:142-397)
// ----------------------------------------------------------------------------------------------
// Pass 1 turns every normal pair into either an immediate result or a "plan": literal pieces and NW
// jobs whose concatenation is what GenerateNormalPairAlignment leaves in frag1/frag2.  Every decision
// up to the NW call depends on the fragment alone, so all jobs of a chunk can be collected first.
struct Piece {
	int job;              // >= 0: index into the chunk's job list; -1: literal
	std::string a, b;
};

struct Plan {
	std::vector<Piece> pieces;
};

void plan_alignment(const Ctx &cx, int rLen, const std::string &frag1, int gLen, const std::string &frag2, Plan &plan,
                    NwJobs &jobs)
{
	if (rLen > 30 && gLen > 30) {
		int max_shift;
		if (cx.opt.pacbio) {
			max_shift = rLen > gLen ? (int)(rLen * 0.2) : (int)(gLen * 0.2);
			if (max_shift > 50) max_shift = 50;
		} else max_shift = cx.opt.max_gaps;
		std::vector<Pair> part;
		simple_pairs_from_fragments(max_shift, rLen, frag1.c_str(), gLen, frag2.c_str(), part);
		if (!part.empty()) identify_normal_pairs(rLen, gLen, part);
		if (!part.empty()) {
			for (size_t i = 0; i < part.size(); ++i) {
				const Pair &p = part[i];
				if (p.rLen <= 0 && p.gLen <= 0) continue;
				Piece pc;
				pc.job = -1;
				if (p.gLen == 0) {
					pc.a = frag1.substr((size_t)p.rPos, (size_t)p.rLen);
					pc.b.assign((size_t)p.rLen, '-');
					plan.pieces.push_back(pc);
				} else if (p.rLen == 0) {
					pc.a.assign((size_t)p.gLen, '-');
					pc.b = frag2.substr((size_t)p.gPos, (size_t)p.gLen);
					plan.pieces.push_back(pc);
				} else if ((p.rLen == 1 && p.gLen == 1) || p.simple) {
					pc.a = frag1.substr((size_t)p.rPos, (size_t)p.rLen);
					pc.b = frag2.substr((size_t)p.gPos, (size_t)p.gLen);
					plan.pieces.push_back(pc);
				} else {
					if (cx.opt.pacbio && (p.rLen > 300 || p.gLen > 300)) {
						std::string s1 = frag1.substr((size_t)p.rPos, (size_t)p.rLen), s2 = frag2.substr((size_t)p.gPos, (size_t)p.gLen);
						plan_alignment(cx, p.rLen, s1, p.gLen, s2, plan, jobs);
					} else {
						pc.job = jobs.add(frag1.data() + p.rPos, p.rLen, frag2.data() + p.gPos, p.gLen);
						plan.pieces.push_back(pc);
					}
				}
			}
			return;
		}
	}
	Piece pc;
	pc.job = jobs.add(frag1.data(), rLen, frag2.data(), gLen);
	plan.pieces.push_back(pc);
}

void stitch(const Plan &plan, const NwJobs &jobs, std::string &aln1, std::string &aln2)
{
	aln1.clear(); aln2.clear();
	for (size_t i = 0; i < plan.pieces.size(); ++i) {
		const Piece &pc = plan.pieces[i];
		if (pc.job < 0) { aln1 += pc.a; aln2 += pc.b; continue; }
		// re-insert the gaps the kernel's op string describes (what nw_alignment does in place)
		size_t j = (size_t)pc.job;
		const char *a = jobs.f1.data() + jobs.o1[j], *b = jobs.f2.data() + jobs.o2[j];
		const uint8_t *op = jobs.ops.data() + jobs.o1[j] + jobs.o2[j];
		for (int t = 0, L = jobs.len[j]; t < L; ++t) {
			if (op[t] == KG_OP_DIAG) { aln1 += *a++; aln2 += *b++; }
			else if (op[t] == KG_OP_GAP1) { aln1 += '-'; aln2 += *b++; }
			else { aln1 += *a++; aln2 += '-'; }
		}
	}
}

int mismatches(int len, const char *a, const char *b)  // CalFragPairMismatchBases, src/tools.cpp:40-47 (raw characters)
{
	int c = 0;
	for (int i = 0; i < len; ++i)
		if (a[i] != b[i]) c++;
	return c;
}

int add_cigar(const std::string &s1, const std::string &s2, CigarVec &cig)  // AddNewCigarElements, src/tools.cpp:49-104
{
	char state = '*';
	int c = 0, score = 0;
	for (size_t i = 0; i < s1.size(); ++i) {
		char st;
		if (s1[i] == '-') st = 'D';
		else if (s2[i] == '-') st = 'I';
		else {
			st = 'M';
			if (s1[i] == s2[i]) score++;
		}
		if (st == state) c++;
		else {
			if (c > 0) cig.push_back(std::make_pair(c, state));
			c = 1;
			state = st;
		}
	}
	if (c > 0) cig.push_back(std::make_pair(c, state));
	return score;
}

bool local_quality_ok(const std::string &a1, const std::string &a2)  // CheckLocalAlignmentQuality, src/tools.cpp:255-290
{
	int type = -1, n = 0, mis = 0, runs = 0;
	for (size_t i = 0; i < a1.size(); ++i) {
		int t;
		if (a1[i] == '-') t = 0;
		else if (a2[i] == '-') t = 1;
		else {
			t = 2;
			n++;
			if (a1[i] != a2[i]) mis++;
		}
		if (t != type) { type = t; runs++; }
	}
	return !(runs >= 4 || (mis >= 3 && mis >= (int)(n * 0.3)));
}

// what pass 1 decided for one pair of a candidate
struct PairWork {
	enum Kind { NONE, SIMPLE, IMMEDIATE, PLANNED } kind = NONE;
	std::pair<int, char> op{0, '\0'};   // IMMEDIATE: always exactly one CIGAR element (or none)
	int score = 0;    // IMMEDIATE
	Plan plan;        // PLANNED
};

struct CandWork {
	bool valid = false;            // reached the pair loop (Score != 0, coordinates valid)
	std::vector<PairWork> pairs;
};

bool quick_match(const Pair &sp, const char *f1, const char *f2, int &n)  // the <=2-mismatch shortcut, src/tools.cpp:240,301,352
{
	if (sp.rLen != sp.gLen) return false;
	n = mismatches(sp.rLen, f1, f2);
	return n <= 2 && n <= (int)(sp.rLen * 0.2);
}

// pass 1 for one pair; role: 0 head, 1 inner, 2 tail
void plan_pair(const Ctx &cx, const Read &rd, const Pair &sp, int role, PairWork &w, NwJobs &jobs)
{
	if (role == 1 && (sp.rLen == 0 || sp.gLen == 0)) {   // ProcessNormalSequencePair :229-233
		w.kind = PairWork::IMMEDIATE;
		if (sp.rLen > 0) w.op = std::make_pair(sp.rLen, 'I');
		else if (sp.gLen > 0) w.op = std::make_pair(sp.gLen, 'D');
		return;
	}
	int n = 0;
	bool shortcut = (role == 1 || !cx.opt.pacbio) && quick_match(sp, rd.seq.data() + sp.rPos, cx.refseq() + sp.gPos, n);
	if (shortcut) {
		w.kind = PairWork::IMMEDIATE;
		w.score = sp.rLen - n;
		w.op = std::make_pair(sp.rLen, 'M');
		return;
	}
	if (!cx.opt.pacbio && ((role == 0 && sp.rLen > 50) || (role == 2 && sp.rLen > 100))) {   // :307-311, :358-362
		w.kind = PairWork::IMMEDIATE;
		w.score = 0;
		w.op = std::make_pair(sp.rLen, 'S');
		return;
	}
	if (sp.rLen == 1 && sp.gLen == 1 && rd.seq[(size_t)sp.rPos] != '-') {   // (a literal '-' in a read is booked as a deletion by the reference's CIGAR scan)
		// one base against one base -- the mismatch right next to a maximal exact match, by far the most common gap:
		// nw_alignment can only answer with the diagonal (+-1.5 against -3 for two gaps), the quality check passes a
		// single column, nothing is trimmed, and AddNewCigarElements (src/tools.cpp:49-104) books 1M with one
		// identical base iff the raw characters are equal -- no job for the kernel
		w.kind = PairWork::IMMEDIATE;
		w.score = rd.seq[(size_t)sp.rPos] == cx.refseq()[sp.gPos] ? 1 : 0;
		w.op = std::make_pair(1, 'M');
		return;
	}
	w.kind = PairWork::PLANNED;
	std::string f1(rd.seq.data() + sp.rPos, (size_t)sp.rLen), f2(cx.refseq() + sp.gPos, (size_t)sp.gLen);
	plan_alignment(cx, sp.rLen, f1, sp.gLen, f2, w.plan, jobs);
}

// pass 2: ProcessHeadSequencePair / ProcessTailSequencePair after the alignment is known
int finish_head(Pair &sp, std::string &a1, std::string &a2, CigarVec &cig)  // src/tools.cpp:314-339
{
	if (!local_quality_ok(a1, a2)) {
		cig.push_back(std::make_pair(sp.rLen, 'S'));
		return 0;
	}
	size_t p = 0;
	while (p < a1.size() && a1[p] == '-') p++;
	if (p > 0) {
		a1.erase(0, p); a2.erase(0, p);
		sp.gPos += (int64_t)p; sp.gLen -= (int)p;
	}
	p = 0;
	while (p < a2.size() && a2[p] == '-') p++;
	if (p > 0) {
		a1.erase(0, p); a2.erase(0, p);
		sp.rPos += (int)p; sp.rLen -= (int)p;
		cig.push_back(std::make_pair((int)p, 'S'));
	}
	return add_cigar(a1, a2, cig);
}

int finish_tail(Pair &sp, std::string &a1, std::string &a2, CigarVec &cig)  // src/tools.cpp:366-394
{
	if (!local_quality_ok(a1, a2)) {
		cig.push_back(std::make_pair(sp.rLen, 'S'));
		return 0;
	}
	int c = 0;
	for (int p = (int)a1.size() - 1; p >= 0 && a1[(size_t)p] == '-'; --p) c++;
	if (c > 0) {
		a1.resize(a1.size() - (size_t)c); a2.resize(a2.size() - (size_t)c);
		sp.gLen -= c;
	}
	c = 0;
	for (int p = (int)a2.size() - 1; p >= 0 && a2[(size_t)p] == '-'; --p) c++;
	if (c > 0) {
		a1.resize(a1.size() - (size_t)c); a2.resize(a2.size() - (size_t)c);
		sp.rLen -= c;
	}
	int score = add_cigar(a1, a2, cig);
	if (c > 0) cig.push_back(std::make_pair(c, 'S'));
	return score;
}

// ----------------------------------------------------------------------------------------------
// report: GenMappingReport and helpers (src/AlignmentCandidates.cpp:492-745)
// ----------------------------------------------------------------------------------------------
std::string cigar_string(const CigarVec &cig)  // GenerateCIGAR, :492-513
{
	std::string out;
	char state = '\0';
	int c = 0;
	auto emit = [&out](int n, char st) {
		char buf[16];
		int k = 0;
		do { buf[k++] = (char)('0' + n % 10); n /= 10; } while (n);
		while (k) out += buf[--k];
		out += st;
	};
	for (size_t i = 0; i < cig.size(); ++i) {
		if (cig[i].second != state) {
			if (c > 0) emit(c, state);
			c = cig[i].first;
			state = cig[i].second;
		} else c += cig[i].first;
	}
	if (c > 0) emit(c, state);
	return out;
}

bool coordinates_valid(const Ctx &cx, const std::vector<Pair> &v)  // CheckCoordinateValidity, :582-610
{
	int64_t g1 = 0, g2 = cx.ref.two_genome_size;
	for (size_t i = 0; i < v.size(); ++i)
		if (v[i].gLen > 0) { g1 = v[i].gPos; break; }
	for (size_t i = v.size(); i-- > 0;)
		if (v[i].gLen > 0) { g2 = v[i].gPos + v[i].gLen - 1; break; }
	int64_t L = cx.ref.genome_size;
	if ((g1 < L && g2 >= L) || (g1 >= L && g2 < L)) return false;
	std::map<int64_t, int>::const_iterator i1 = cx.ref.chr_end.lower_bound(g1), i2 = cx.ref.chr_end.lower_bound(g2);
	if (i1 == cx.ref.chr_end.end() || i2 == cx.ref.chr_end.end() || i1->second != i2->second) return false;
	return true;
}

void make_coordinate(const Ctx &cx, bool first, int64_t gPos, int64_t end_gPos, CigarVec &cig, Report &rp)  // GenCoordinateInfo, :515-562
{
	const RefData &ref = cx.ref;
	int n_chr = (int)ref.contigs.size();
	if (gPos < ref.genome_size) {
		rp.fwd = first;
		if (n_chr == 1) { rp.chr = 0; rp.gPos = gPos + 1; }
		else {
			std::map<int64_t, int>::const_iterator it = ref.chr_end.lower_bound(gPos);
			rp.chr = it->second;
			rp.gPos = gPos + 1 - ref.contigs[(size_t)rp.chr].fwd_start;
		}
	} else {
		rp.fwd = !first;
		std::reverse(cig.begin(), cig.end());
		if (n_chr == 1) { rp.chr = 0; rp.gPos = ref.two_genome_size - end_gPos; }
		else {
			std::map<int64_t, int>::const_iterator it = ref.chr_end.lower_bound(gPos);
			if (it == ref.chr_end.end()) --it;   // beyond the text: undefined in the reference; stay in range
			rp.gPos = it->first - end_gPos + 1;
			rp.chr = it->second;
		}
	}
	rp.cigar = cigar_string(cig);
}

int gap_penalty(const CigarVec &cig)  // GapPenalty, :612-622
{
	int gp = 0;
	for (size_t i = 0; i < cig.size(); ++i)
		if (cig[i].second == 'I' || cig[i].second == 'D') gp += cig[i].first;
	return gp;
}

// pass 1 of GenMappingReport for one read: normal pairs, validity, and the NW jobs of every pair
void report_plan(const Ctx &cx, Read &rd, std::vector<Candidate> &cands, std::vector<CandWork> &work, NwJobs &jobs)
{
	work.assign(cands.size(), CandWork());
	for (size_t i = 0; i < cands.size(); ++i) {
		if (cands[i].score == 0) continue;
		// (PacBio: the reference skips this candidate when an earlier one already scored; that is only
		// known in pass 2, so its jobs are planned anyway and simply never read.)
		identify_normal_pairs(rd.rlen, -1, cands[i].pairs);
		if (!coordinates_valid(cx, cands[i].pairs)) continue;
		CandWork &cw = work[i];
		cw.valid = true;
		std::vector<Pair> &v = cands[i].pairs;
		int num = (int)v.size();
		cw.pairs.assign((size_t)num, PairWork());
		for (int j = 0; j < num; ++j) {
			PairWork &w = cw.pairs[(size_t)j];
			if (v[j].rLen == 0 && v[j].gLen == 0) continue;
			if (v[j].simple) { w.kind = PairWork::SIMPLE; continue; }
			if (j == 0 || j == num - 1) {
				if (v[j].rLen > 3000) {               // :671-676, :690-695
					w.kind = PairWork::IMMEDIATE;
					w.op = std::make_pair(v[j].rLen, 'S');
					w.score = -1;                     // marks the long soft clip (handled like s == 0 but unconditionally)
					continue;
				}
				plan_pair(cx, rd, v[j], j == 0 ? 0 : 2, w, jobs);
			} else plan_pair(cx, rd, v[j], 1, w, jobs);
		}
	}
}

// pass 2 of GenMappingReport
void report_finish(const Ctx &cx, bool first, Read &rd, std::vector<Candidate> &cands, std::vector<CandWork> &work,
                   const NwJobs &jobs)
{
	rd.score = rd.sub_score = rd.best = 0;
	rd.can_num = (int)cands.size();
	if (rd.can_num == 0) {
		rd.can_num = 1;
		rd.best = 0;
		rd.rep.assign(1, Report());
		return;
	}
	rd.rep.assign((size_t)rd.can_num, Report());
	std::string a1, a2;
	for (int i = 0; i < rd.can_num; ++i) {
		Report &rp = rd.rep[(size_t)i];
		rp.score = 0;
		rp.mate = cands[(size_t)i].mate;
		if (cands[(size_t)i].score == 0) continue;
		if (cx.opt.pacbio && rd.score > 0) { rd.sub_score = rd.score; continue; }
		CandWork &cw = work[(size_t)i];
		if (!cw.valid) continue;
		std::vector<Pair> &v = cands[(size_t)i].pairs;
		int num = (int)v.size();
		CigarVec cig;
		cig.reserve((size_t)num + 4);
		for (int j = 0; j < num; ++j) {
			PairWork &w = cw.pairs[(size_t)j];
			if (w.kind == PairWork::NONE) continue;
			if (w.kind == PairWork::SIMPLE) {
				cig.push_back(std::make_pair(v[j].rLen, 'M'));
				rp.score += v[j].rLen;
				continue;
			}
			bool head = j == 0, tail = j == num - 1 && !head;
			int s;
			if (w.kind == PairWork::IMMEDIATE) {
				if (w.op.second != '\0') cig.push_back(w.op);
				s = w.score;
			} else {
				stitch(w.plan, jobs, a1, a2);
				if (head) s = finish_head(v[j], a1, a2, cig);
				else if (tail) s = finish_tail(v[j], a1, a2, cig);
				else s = add_cigar(a1, a2, cig);
			}
			if (head) {
				if (s > 0) rp.score += s;
				if (s <= 0) {   // s == 0, or the > 3000 soft clip (score -1): collapse the genome side, :674-686
					v[0].gPos = v[1].gPos;
					v[0].gLen = 0;
				}
			} else if (tail) {
				if (s > 0) rp.score += s;
				if (s <= 0) {
					v[j].gPos = v[j - 1].gPos + v[j - 1].gLen;
					v[j].gLen = 0;
				}
			} else rp.score += s;
		}
		if (!cx.opt.pacbio && cig.size() > 1) {
			rp.score -= gap_penalty(cig);
			if (rp.score <= 0) { rp.score = 0; continue; }
		}
		if (cig.empty()) rp.score = 0;
		else {
			make_coordinate(cx, first, v[0].gPos, v[(size_t)num - 1].gPos + v[(size_t)num - 1].gLen - 1, cig, rp);
			if (rp.gPos <= 0) rp.score = 0;
		}
		if (rp.score > rd.score) {
			rd.best = i;
			rd.sub_score = rd.score;
			rd.score = rp.score;
		} else if (rp.score == rd.score) {
			rd.sub_score = rd.score;
			if (!cx.opt.multi_hit && cx.ref.contigs[(size_t)rp.chr].len > cx.ref.contigs[(size_t)rd.rep[(size_t)rd.best].chr].len) rd.best = i;
		}
	}
}

// ----------------------------------------------------------------------------------------------
// pairing, filters, rescue (src/Mapping.cpp:317-480, src/AlignmentRescue.cpp)
// ----------------------------------------------------------------------------------------------
void remove_redundant(const Ctx &cx, std::vector<Candidate> &v)  // RemoveRedundantCandidates, src/Mapping.cpp:317-346
{
	if (v.size() <= 1) return;
	int s1 = 0, s2 = 0;
	for (size_t i = 0; i < v.size(); ++i) {
		if (v[i].score > s2) {
			if (v[i].score >= s1) { s2 = s1; s1 = v[i].score; }
			else s2 = v[i].score;
		}
	}
	int thr = (cx.opt.pacbio || s1 == s2 || s1 - s2 > 20) ? s1 : s2;
	for (size_t i = 0; i < v.size(); ++i)
		if (v[i].score < thr) v[i].score = 0;
}

bool pair_candidates(const Ctx &cx, int64_t est, std::vector<Candidate> &v1, std::vector<Candidate> &v2, PairStats &ps)  // CheckPairedAlignmentCandidates, :348-400
{
	bool pairing = false;
	int n1 = (int)v1.size(), n2 = (int)v2.size();
	if (n1 * n2 > 1000) { remove_redundant(cx, v1); remove_redundant(cx, v2); }
	for (int i = 0; i < n1; ++i) {
		if (v1[i].score == 0) continue;
		int best = -1, s = 0;
		for (int j = 0; j < n2; ++j) {
			if (v2[j].score == 0 || v2[j].posDiff < v1[i].posDiff) continue;
			int64_t dist = v2[j].posDiff - v1[i].posDiff;
			if (dist < est) { if (dist > ps.lo) ps.lo = dist; } else if (dist < ps.hi) ps.hi = dist;
			if (dist < est) {
				if (v2[j].score > s) { best = j; s = v2[j].score; }
				else if (v2[j].score == s) best = -1;
			}
		}
		if (s > 0 && best != -1) {
			int j = best;
			if (v2[j].mate == -1) {
				pairing = true;
				v1[i].mate = j;
				v2[j].mate = i;
			} else if (v1[i].score > v1[v2[j].mate].score) {
				v1[v2[j].mate].mate = -1;
				v1[i].mate = j;
				v2[j].mate = i;
			}
		}
	}
	return pairing;
}

void remove_unmated(std::vector<Candidate> &v1, std::vector<Candidate> &v2)  // RemoveUnMatedAlignmentCandidates, :402-427
{
	for (size_t i = 0; i < v1.size(); ++i) {
		if (v1[i].mate == -1) v1[i].score = 0;
		else {
			int j = v1[i].mate;
			v1[i].score = v2[j].score = v1[i].score + v2[j].score;
		}
	}
	for (size_t j = 0; j < v2.size(); ++j)
		if (v2[j].mate == -1) v2[j].score = 0;
}

int max_score(const std::vector<Candidate> &v)
{
	int s = 0;
	for (size_t i = 0; i < v.size(); ++i)
		if (v[i].score > s) s = v[i].score;
	return s;
}

// the k-mer hits between a read (k-mers `kr`, registered in `tab`) and a reference window
void rescue_hits(int slen, const std::vector<Kmer> &kr, const KmerTable &tab, const char *window, std::vector<KmerHit> &hits)
{
	if (kr.size() < 65535) {
		window_hits(slen, kr, tab, slen, window, hits);
	} else {            // more k-mers than the table's 16-bit links address: the reference's sort-and-search join
		std::vector<Kmer> kg;
		kmers_of(slen, window, kg);
		common_kmers(slen, kr, kg, hits);
	}
}

// IdnetifyRescueCandidate, src/AlignmentRescue.cpp:24-69
Candidate rescue_candidate(const Ctx &cx, int64_t gPos, std::vector<Pair> &vec)
{
	Candidate best;
	best.score = 0;
	best.mate = -1;
	int num = (int)vec.size();
	for (int i = 0; i < num;) {
		vec[i].gPos += gPos;
		int s = vec[i].rLen;
		std::vector<Pair> grp(1, vec[i]);
		int j;
		for (j = i + 1; j < num; ++j) {
			if (vec[j].posDiff - vec[i].posDiff < cx.opt.max_gaps) {
				vec[j].gPos += gPos;
				s += vec[j].rLen;
				grp.push_back(vec[j]);
			} else break;
		}
		if (s > best.score) {
			best.score = s;
			best.posDiff = grp[0].posDiff + gPos;
			best.pairs = grp;
		}
		i = j;
	}
	std::sort(best.pairs.begin(), best.pairs.end(), pair_by_gpos);
	for (size_t i = 0; i < best.pairs.size(); ++i) best.pairs[i].posDiff += gPos;
	return best;
}

// RescueUnpairedAlignment, src/AlignmentRescue.cpp:71-168.  The reference can index RefSequence
// before its start or dereference ChrLocMap.end() here; those windows are
// clamped to the text instead (inputs that trigger it have no defined reference output).
bool rescue_unpaired(const Ctx &cx, int est, const Read &r1, const Read &r2, std::vector<Candidate> &v1, std::vector<Candidate> &v2)
{
	const RefData &ref = cx.ref;
	int score1 = max_score(v1), score2 = max_score(v2);
	int strategy;
	if (score1 == 0 && score2 == 0) return false;
	else if (score1 < (int)(r1.rlen * 0.1) && score2 < (int)(r2.rlen * 0.1)) strategy = 4;
	else if (score1 > score2 && score1 - score2 > 50) strategy = 1;
	else if (score2 > score1 && score2 - score1 > 50) strategy = 2;
	else strategy = 3;
	if (est > cx.opt.max_insert) est = cx.opt.max_insert;
	bool mated = false;
	int num1 = (int)v1.size(), num2 = (int)v2.size();
	std::vector<Kmer> kr;
	std::vector<KmerHit> hits;
	std::vector<Pair> sp;
	static thread_local KmerTable tab;
	if (strategy == 1 || strategy == 3) {
		int thr = max_score(v1) - 30;
		if (thr < 50) thr = 50;
		kmers_of(r2.rlen, r2.seq.data(), kr);
		tab.set(kr);
		for (int j = num2, i = 0; i < num1; ++i) {
			if (v1[i].score < thr) continue;
			int64_t left = v1[i].posDiff, right = v1[i].posDiff + est + r2.rlen;
			std::map<int64_t, int>::const_iterator it = ref.chr_end.lower_bound(left);
			if (it == ref.chr_end.end()) continue;
			int chr = it->second;
			if (right < ref.genome_size && right > ref.contigs[(size_t)chr].fwd_start) right = ref.contigs[(size_t)chr].fwd_start - 1;
			else if (right >= ref.genome_size && right > ref.contigs[(size_t)chr].rev_start) right = ref.contigs[(size_t)chr].rev_start - 1;
			int slen = (int)(right - left);
			if (slen < r2.rlen) continue;
			if (left < 0 || right > ref.two_genome_size) continue;
			rescue_hits(slen, kr, tab, cx.refseq() + left, hits);
			simple_pairs_from_kmers(10, hits, sp);
			Candidate c = rescue_candidate(cx, left, sp);
			if (c.score > score2) {
				mated = true;
				c.mate = i;
				v1[i].mate = j++;
				v2.push_back(c);
			}
		}
		tab.clear(kr);
	}
	if (strategy == 2 || strategy == 3) {
		int thr = max_score(v2) - 30;   // the rescued entries appended above are included, as in the reference
		if (thr < 50) thr = 50;
		kmers_of(r1.rlen, r1.seq.data(), kr);
		tab.set(kr);
		for (int i = num1, j = 0; j < num2; ++j) {
			if (v2[j].score < thr) continue;
			int64_t left = v2[j].posDiff - est, right = v2[j].posDiff + r2.rlen;
			std::map<int64_t, int>::const_iterator it = ref.chr_end.lower_bound(right);
			if (it == ref.chr_end.end()) continue;
			int chr = it->second;
			const Contig &cg = ref.contigs[(size_t)chr];
			if (left < ref.genome_size && left < (cg.fwd_start - cg.len)) left = cg.fwd_start - cg.len + 1;
			else if (right >= ref.genome_size && left < (cg.rev_start - cg.len)) left = cg.rev_start - cg.len + 1;
			int slen = (int)(right - left);
			if (slen < r1.rlen) continue;
			if (left < 0) { left = 0; slen = (int)(right - left); if (slen < r1.rlen) continue; }
			if (right > ref.two_genome_size) continue;
			rescue_hits(slen, kr, tab, cx.refseq() + left, hits);
			simple_pairs_from_kmers(10, hits, sp);
			Candidate c = rescue_candidate(cx, left, sp);
			if (c.score > score1) {
				mated = true;
				c.mate = j;
				v2[j].mate = i++;
				v1.push_back(c);
			}
		}
		tab.clear(kr);
	}
	return mated;
}

void check_final_pair(const Ctx &cx, Read &r1, Read &r2)  // CheckPairedFinalAlignments, src/Mapping.cpp:429-480
{
	bool mated = false;
	if (r1.best != -1 && r2.best != -1) mated = r1.rep[(size_t)r1.best].mate == r2.best;
	if (!cx.opt.multi_hit && mated) return;
	if (!mated && r1.score > 0 && r2.score > 0) {
		int s = 0;
		for (int i = 0; i < r1.can_num; ++i) {
			int j;
			if (r1.rep[(size_t)i].score > 0 && (j = r1.rep[(size_t)i].mate) != -1 && r2.rep[(size_t)j].score > 0) {
				mated = true;
				if (s < r1.rep[(size_t)i].score + r2.rep[(size_t)j].score) {
					s = r1.rep[(size_t)i].score + r2.rep[(size_t)j].score;
					r1.best = i; r1.score = r1.rep[(size_t)i].score;
					r2.best = j; r2.score = r2.rep[(size_t)j].score;
				}
			}
		}
	}
	if (mated) {
		for (int i = 0; i < r1.can_num; ++i) {
			int j;
			if (r1.rep[(size_t)i].score != r1.score || ((j = r1.rep[(size_t)i].mate) != -1 && r2.rep[(size_t)j].score != r2.score)) {
				r1.rep[(size_t)i].score = 0;
				r1.rep[(size_t)i].mate = -1;
			}
		}
	} else {
		for (int i = 0; i < r1.can_num; ++i) {
			r1.rep[(size_t)i].mate = -1;
			if (r1.rep[(size_t)i].score > 0 && r1.rep[(size_t)i].score != r1.score) r1.rep[(size_t)i].score = 0;
		}
		for (int j = 0; j < r2.can_num; ++j) {
			r2.rep[(size_t)j].mate = -1;
			if (r2.rep[(size_t)j].score > 0 && r2.rep[(size_t)j].score != r2.score) r2.rep[(size_t)j].score = 0;
		}
	}
}

void set_single_flag(Read &rd)  // SetSingleAlignmentFlag, src/Mapping.cpp:49-71
{
	if (rd.score > rd.sub_score) rd.rep[(size_t)rd.best].flag = rd.rep[(size_t)rd.best].fwd ? 0 : 0x10;
	else if (rd.score > 0) {
		for (int i = 0; i < rd.can_num; ++i)
			if (rd.rep[(size_t)i].score > 0) rd.rep[(size_t)i].flag = rd.rep[(size_t)i].fwd ? 0 : 0x10;
	} else rd.rep[0].flag = 0x4;
}

void set_one_mate_flags(Read &me, Read &other, int base_flag)  // the per-mate halves of SetPairedAlignmentFlag, :96-156
{
	if (me.score > me.sub_score) {
		Report &rp = me.rep[(size_t)me.best];
		rp.flag = base_flag | (rp.fwd ? 0x20 : 0x10);
		int j = rp.mate;
		if (j != -1 && other.rep[(size_t)j].score > 0) rp.flag |= 0x2;
		else rp.flag |= 0x8;
	} else if (me.score > 0) {
		for (int i = 0; i < me.can_num; ++i) {
			Report &rp = me.rep[(size_t)i];
			if (rp.score <= 0) continue;
			rp.flag = base_flag | (rp.fwd ? 0x20 : 0x10);
			int j = rp.mate;
			if (j != -1 && other.rep[(size_t)j].score > 0) rp.flag |= 0x2;
			else rp.flag |= 0x8;
		}
	} else {
		me.rep[0].flag = base_flag | 0x4;
		if (other.score == 0) me.rep[0].flag |= 0x8;
		else me.rep[0].flag |= (other.rep[(size_t)other.best].fwd ? 0x10 : 0x20);
	}
}

void set_paired_flags(Read &r1, Read &r2)  // SetPairedAlignmentFlag, src/Mapping.cpp:73-158
{
	if (r1.score > r1.sub_score && r2.score > r2.sub_score) {
		Report &a = r1.rep[(size_t)r1.best], &b = r2.rep[(size_t)r2.best];
		a.flag = 0x41;
		b.flag = 0x81;
		if (r2.best == a.mate) { a.flag |= 0x2; b.flag |= 0x2; }
		a.flag |= a.fwd ? 0x20 : 0x10;
		b.flag |= b.fwd ? 0x20 : 0x10;
	} else {
		set_one_mate_flags(r1, r2, 0x41);
		set_one_mate_flags(r2, r1, 0x81);
	}
}

void evaluate_mapq(const Ctx &cx, Read &rd)  // EvaluateMAPQ, src/Mapping.cpp:160-175
{
	if (rd.score == 0 || rd.score == rd.sub_score) { rd.mapq = 0; return; }
	if (cx.opt.pacbio) {
		float scale = 85.0 * (int)(ceil(rd.rlen / 100 + 0.5));
		if (scale > 2000) scale = 2000;
		rd.mapq = (int)(60 * (rd.score / scale));
	} else if (rd.sub_score == 0 || rd.score - rd.sub_score > 5) rd.mapq = 60;
	else rd.mapq = (int)(30 * (1 - (float)(rd.score - rd.sub_score) / rd.score) * log(rd.score) + 0.4999);
	if (rd.mapq > 60) rd.mapq = 60;
}

// ----------------------------------------------------------------------------------------------
// SAM text (src/Mapping.cpp:177-315; record formats in SURVEY.md App. D)
// ----------------------------------------------------------------------------------------------
inline void append_int(std::string &out, long long v)   // what "%d" / "%lld" print
{
	char buf[24];
	int n = 0;
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
	if (v < 0) buf[n++] = '-';
	while (n) out += buf[--n];
}

void sam_unmapped(const Ctx &cx, const Read &rd, std::string &out)
{
	out += rd.name; out += '\t';
	append_int(out, rd.rep[0].flag);
	out += "\t*\t0\t0\t*\t*\t0\t0\t";
	out += rd.seq; out += '\t';
	if (cx.fastq) out += rd.qual; else out += '*';
	out += "\tAS:i:0\tXS:i:0\n";
}

// raw-pointer writers for the record below: one capacity check per record instead of one per field
inline char *put(char *p, std::string_view v) { memcpy(p, v.data(), v.size()); return p + v.size(); }
inline char *put(char *p, const char *lit, size_t n) { memcpy(p, lit, n); return p + n; }
inline char *put_int(char *p, long long v)   // what "%d" / "%lld" print
{
	char buf[24];
	int n = 0;
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
	if (v < 0) buf[n++] = '-';
	while (n) *p++ = buf[--n];
	return p;
}

// flip: the record shows the reverse complement of the read as it is held (and its qualities reversed); written straight
// into the chunk's text, no temporaries
void sam_mapped(const Ctx &cx, const Read &rd, const Report &rp, bool has_mate, long long mate_pos, int tlen, bool flip, std::string &out)
{
	const std::string &chr = cx.ref.contigs[(size_t)rp.chr].name;
	size_t at = out.size();
	out.resize(at + rd.name.size() + chr.size() + rp.cigar.size() + rd.seq.size() + rd.qual.size() + 192);
	char *p = &out[at];
	p = put(p, rd.name); *p++ = '\t';
	p = put_int(p, rp.flag); *p++ = '\t';
	p = put(p, chr); *p++ = '\t';
	p = put_int(p, (long long)rp.gPos); *p++ = '\t';
	p = put_int(p, rd.mapq); *p++ = '\t';
	p = put(p, rp.cigar);
	if (has_mate) { p = put(p, "\t=\t", 3); p = put_int(p, mate_pos); *p++ = '\t'; p = put_int(p, tlen); *p++ = '\t'; }
	else p = put(p, "\t*\t0\t0\t", 7);
	size_t n = rd.seq.size();
	if (!flip) p = put(p, rd.seq);
	else {
		const char *sq = rd.seq.data();
		for (size_t i = 0; i < n; ++i) p[i] = comp_base(sq[n - 1 - i]);      // GetComplementarySeq, src/tools.cpp:19-29
		p += n;
	}
	*p++ = '\t';
	if (!cx.fastq) *p++ = '*';
	else if (!flip) p = put(p, rd.qual);
	else {
		n = rd.qual.size();
		const char *ql = rd.qual.data();
		for (size_t i = 0; i < n; ++i) p[i] = ql[n - 1 - i];
		p += n;
	}
	p = put(p, "\tNM:i:", 6); p = put_int(p, rd.rlen - rd.score);
	p = put(p, "\tAS:i:", 6); p = put_int(p, rd.score);
	p = put(p, "\tXS:i:", 6); p = put_int(p, rd.sub_score);
	*p++ = '\n';
	out.resize((size_t)(p - out.data()));
}

// OutputPairedAlignments, src/Mapping.cpp:177-270.  Mate 2 is held reverse-complemented (App. B-2).
void output_pair(const Ctx &cx, const Read &r1, const Read &r2, Stats &st, PairStats &ps, std::string &out)
{
	if (r1.score == 0) { st.unmapped++; sam_unmapped(cx, r1, out); }
	else {
		if (r1.mapq == 60) st.unique++;
		for (int i = r1.best; i < r1.can_num; ++i) {
			const Report &rp = r1.rep[(size_t)i];
			if (rp.score > 0) {
				int j = rp.mate;
				if (j != -1 && r2.rep[(size_t)j].score > 0) {
					int dist = (int)(r2.rep[(size_t)j].gPos - rp.gPos + (rp.fwd ? r2.rlen : 0 - r1.rlen));
					if (i == r1.best) {
						ps.paired += 2;
						if (abs(dist) < 10000) ps.distance += abs(dist);
					}
					sam_mapped(cx, r1, rp, true, (long long)r2.rep[(size_t)j].gPos, dist, !rp.fwd, out);
				} else sam_mapped(cx, r1, rp, false, 0, 0, !rp.fwd, out);
			}
			if (!cx.opt.multi_hit) break;
		}
	}
	if (r2.score == 0) { st.unmapped++; sam_unmapped(cx, r2, out); }
	else {
		if (r2.mapq == 60) st.unique++;
		// mate 2 is held reverse-complemented (src/GetData.cpp:125-135): a forward report shows it flipped back
		for (int j = r2.best; j < r2.can_num; ++j) {
			const Report &rp = r2.rep[(size_t)j];
			if (rp.score > 0) {
				int i = rp.mate;
				if (i != -1 && r1.rep[(size_t)i].score > 0) {
					int dist = 0 - (int)(rp.gPos - r1.rep[(size_t)i].gPos + (r1.rep[(size_t)i].fwd ? r2.rlen : 0 - r1.rlen));
					sam_mapped(cx, r2, rp, true, (long long)r1.rep[(size_t)i].gPos, dist, rp.fwd, out);
				} else sam_mapped(cx, r2, rp, false, 0, 0, rp.fwd, out);
			}
			if (!cx.opt.multi_hit) break;
		}
	}
}

void output_single(const Ctx &cx, const Read &rd, Stats &st, std::string &out)  // OutputSingledAlignments, src/Mapping.cpp:272-315
{
	if (rd.score == 0) { st.unmapped++; sam_unmapped(cx, rd, out); return; }
	if (rd.mapq == 60) st.unique++;
	for (int i = rd.best; i < rd.can_num; ++i) {
		const Report &rp = rd.rep[(size_t)i];
		if (rp.score == rd.score) {
			sam_mapped(cx, rd, rp, false, 0, 0, !rp.fwd, out);
			if (!cx.opt.multi_hit) break;
		}
	}
}

// ----------------------------------------------------------------------------------------------
// input (src/GetData.cpp)
// ----------------------------------------------------------------------------------------------
struct Input {
	FILE *fp = nullptr;
	gzFile gz = nullptr;
	char *line = nullptr;
	size_t cap = 0;
	std::vector<char> gzbuf;
	~Input() { close(); }
	void close()
	{
		if (fp) fclose(fp);
		if (gz) gzclose(gz);
		fp = nullptr; gz = nullptr;
		free(line); line = nullptr; cap = 0;
	}
};

std::string_view header_view(const char *buf, int len)  // IdentifyHeaderBegPos/EndPos, src/GetData.cpp:29-49
{
	int p1 = len - 1, p2 = len - 1;
	for (int i = 1; i < len; ++i)
		if (buf[i] != '>' && buf[i] != '@') { p1 = i; break; }
	for (int i = 1; i < len; ++i)
		if (buf[i] == ' ' || buf[i] == '/' || buf[i] == '\t') { p2 = i; break; }
	return p2 > p1 ? std::string_view(buf + p1, (size_t)(p2 - p1)) : std::string_view();
}

// a read owned by the reader itself (getline()/gzgets() paths)
struct OwnedRead {
	std::string name, seq, qual;
	int rlen = 0;
};

// GetNextEntry, src/GetData.cpp:51-107.  Like the reference, the last character of every line is taken
// to be the newline.
bool next_entry_plain(Input &in, bool fastq, OwnedRead &rd)
{
	rd = OwnedRead();
	ssize_t len = getline(&in.line, &in.cap, in.fp);
	if (len == -1) return false;
	rd.name.assign(header_view(in.line, (int)len));
	if (fastq) {
		ssize_t sl = getline(&in.line, &in.cap, in.fp);
		if (sl == -1) { rd.rlen = 0; return true; }
		rd.rlen = (int)sl - 1;
		rd.seq.assign(in.line, (size_t)rd.rlen);
		getline(&in.line, &in.cap, in.fp);
		ssize_t ql = getline(&in.line, &in.cap, in.fp);
		if (ql < 0) ql = 0;
		rd.qual.assign(in.line, (size_t)std::min<ssize_t>(ql, rd.rlen));
		rd.qual.resize((size_t)rd.rlen, '\0');
		rd.qual = std::string(rd.qual.c_str());   // the reference treats it as a C string
	} else {
		std::string seq;
		while (true) {
			len = getline(&in.line, &in.cap, in.fp);
			if (len == -1) break;
			if (in.line[0] == '>') { fseek(in.fp, 0 - len, SEEK_CUR); break; }
			in.line[len - 1] = '\0';
			seq += in.line;
		}
		rd.rlen = (int)seq.size();
		rd.seq.swap(seq);
	}
	return true;
}

bool next_entry_gz(Input &in, bool fastq, bool pacbio, OwnedRead &rd)  // gzGetNextEntry, src/GetData.cpp:145-182
{
	rd = OwnedRead();
	int buf_size = pacbio ? 1000000 : 1000;
	in.gzbuf.resize((size_t)buf_size);
	char *buf = in.gzbuf.data();
	if (gzgets(in.gz, buf, buf_size) == NULL) return false;
	int len = (int)strlen(buf);
	std::string name(header_view(buf, len));
	if (!name.empty() && (buf[0] == '@' || buf[0] == '>')) {
		rd.name = name;
		if (gzgets(in.gz, buf, buf_size) == NULL) buf[0] = '\0';
		rd.rlen = (int)strlen(buf) - 1;
		if (rd.rlen < 0) rd.rlen = 0;
		rd.seq.assign(buf, (size_t)rd.rlen);
		if (fastq) {
			gzgets(in.gz, buf, buf_size);
			if (gzgets(in.gz, buf, buf_size) == NULL) buf[0] = '\0';
			rd.qual.assign(buf, std::min<size_t>(strlen(buf), (size_t)rd.rlen));
		}
	}
	return true;
}

bool next_entry(Input &in, bool fastq, bool pacbio, OwnedRead &rd)
{
	return in.gz ? next_entry_gz(in, fastq, pacbio, rd) : next_entry_plain(in, fastq, rd);
}

// GetNextChunk, src/GetData.cpp:109-143 / 184-219
int next_chunk(const Ctx &cx, bool sep, Input &in1, Input &in2, std::deque<OwnedRead> &reads, int limit)
{
	int count = 0;
	OwnedRead rd;
	while (true) {
		if (!next_entry(in1, cx.fastq, cx.opt.pacbio, rd) || rd.rlen == 0) break;
		reads.push_back(rd);
		count++;
		bool ok = sep ? next_entry(in2, cx.fastq, cx.opt.pacbio, rd) : next_entry(in1, cx.fastq, cx.opt.pacbio, rd);
		if (!ok || rd.rlen == 0) break;
		if (cx.opt.paired) {   // mate 2 is stored reverse-complemented, qualities reversed, :125-135
			rd.seq = revcomp(rd.seq);
			if (cx.fastq) std::reverse(rd.qual.begin(), rd.qual.end());
		}
		reads.push_back(rd);
		count++;
		if (count == limit) break;
	}
	return count;
}

// ---- fast path for plain FASTQ: the file is mapped, record boundaries are found with memchr by one
// thread (exactly the line structure GetNextEntry walks with four getline() calls), and the reads are
// materialised (copies, mate-2 reverse complement) by the worker threads.
struct MappedFile {
	const char *data = nullptr;
	size_t size = 0, pos = 0;
	bool mapped = false;
	~MappedFile() { if (data && mapped) munmap(const_cast<char *>(data), size); }
	// a block of text owned by someone else (the inflated part of a gz file that a batch holds)
	void attach(const char *text, size_t n) { data = text; size = n; pos = 0; line_end.clear(); next_line = 0; }
	bool open(const std::string &path)
	{
		int fd = ::open(path.c_str(), O_RDONLY);
		if (fd < 0) return false;
		struct stat sb;
		if (fstat(fd, &sb) != 0 || sb.st_size == 0) { ::close(fd); return false; }
		void *p = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
		::close(fd);
		if (p == MAP_FAILED) return false;
		madvise(p, (size_t)sb.st_size, MADV_SEQUENTIAL);
		data = (const char *)p;
		size = (size_t)sb.st_size;
		mapped = true;
		return true;
	}
	// getline(): returns the line length including the newline when there is one, -1 at end of file
	ssize_t line(const char *&start)
	{
		if (pos >= size) return -1;
		start = data + pos;
		size_t len;
		if (next_line < line_end.size()) len = line_end[next_line++] - pos;      // indexed window
		else {
			const char *nl = (const char *)memchr(start, '\n', size - pos);
			len = nl ? (size_t)(nl - start) + 1 : size - pos;
		}
		pos += len;
		return (ssize_t)len;
	}
	// Index the lines of the next `bytes` of the file with all workers: every worker counts the newlines
	// of its slice, a prefix sum places them, a second sweep records the line ends.
	template <class PoolT>
	void index_ahead(PoolT &pool, size_t bytes)
	{
		line_end.clear(); next_line = 0;
		size_t lo = pos, hi = std::min(size, pos + bytes);
		if (hi <= lo) return;
		int parts = std::max(1, std::min<int>(pool.size() * 4, (int)((hi - lo) >> 16) + 1));
		std::vector<size_t> cnt((size_t)parts + 1, 0);
		auto slice = [&](int t, size_t &a, size_t &b) { a = lo + (hi - lo) * (size_t)t / (size_t)parts; b = lo + (hi - lo) * (size_t)(t + 1) / (size_t)parts; };
		pool.run(parts, [&](int t) {
			size_t a, b, c = 0;
			slice(t, a, b);
			for (const char *p = data + a, *e = data + b; p < e;) {
				const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
				if (!nl) break;
				c++; p = nl + 1;
			}
			cnt[(size_t)t + 1] = c;
		});
		for (int t = 0; t < parts; ++t) cnt[(size_t)t + 1] += cnt[(size_t)t];
		line_end.resize(cnt[(size_t)parts]);
		pool.run(parts, [&](int t) {
			size_t a, b, at = cnt[(size_t)t];
			slice(t, a, b);
			for (const char *p = data + a, *e = data + b; p < e;) {
				const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
				if (!nl) break;
				line_end[at++] = (size_t)(nl - data) + 1;
				p = nl + 1;
			}
		});
	}
	std::vector<size_t> line_end;   // absolute end offsets (one past the newline) of the indexed lines
	size_t next_line = 0;
};

struct RecView {
	const char *hdr, *seq, *qual;
	int hdr_len, rlen, qual_len;
	bool flip;    // mate 2 of a pair: stored reverse-complemented (src/GetData.cpp:125-135)
};

// GetNextEntry on a mapped FASTQ (same arithmetic as next_entry_plain)
bool view_next(MappedFile &f, RecView &v)
{
	const char *p;
	ssize_t len = f.line(p);
	if (len == -1) return false;
	v.hdr = p; v.hdr_len = (int)len;
	ssize_t sl = f.line(p);
	if (sl == -1) { v.rlen = 0; v.seq = v.qual = nullptr; v.qual_len = 0; return true; }
	v.seq = p; v.rlen = (int)sl - 1;
	f.line(p);
	ssize_t ql = f.line(p);
	v.qual = p; v.qual_len = ql < 0 ? 0 : (int)ql;
	return true;
}

// The records whose four lines all lie in the indexed window, as views -- the same arithmetic as view_next(), but for all
// records at once on the pool (the serial walk was the largest serial piece of the reader: ~80 ns per read)
template <class PoolT>
void views_of_indexed_records(PoolT &pool, const MappedFile &f, std::vector<RecView> &out)
{
	size_t n = f.line_end.size() / 4;
	out.resize(n);
	if (n == 0) return;
	const size_t first = f.pos;
	const char *data = f.data;
	const std::vector<size_t> &le = f.line_end;
	int parts = std::max(1, std::min<int>(pool.size() * 4, (int)(n >> 12) + 1));
	pool.run(parts, [&](int t) {
		for (size_t j = n * (size_t)t / (size_t)parts, e = n * (size_t)(t + 1) / (size_t)parts; j < e; ++j) {
			size_t l0 = j == 0 ? first : le[4 * j - 1], l1 = le[4 * j], l2 = le[4 * j + 1], l3 = le[4 * j + 2], l4 = le[4 * j + 3];
			RecView &v = out[j];
			v.hdr = data + l0; v.hdr_len = (int)(l1 - l0);
			v.seq = data + l1; v.rlen = (int)(l2 - l1) - 1;
			v.qual = data + l3; v.qual_len = (int)(l4 - l3);
			v.flip = false;
		}
	});
}

// `arena` receives the reverse-complemented copy of a flipped mate (2 * rlen bytes)
void materialise(const RecView &v, Read &rd, char *&arena)
{
	rd = Read();
	rd.name = header_view(v.hdr, v.hdr_len);
	rd.rlen = v.rlen;
	int ql = std::min(v.qual_len, v.rlen);
	const char *z = ql > 0 ? (const char *)memchr(v.qual, '\0', (size_t)ql) : nullptr;
	if (z) ql = (int)(z - v.qual);
	if (!v.flip) {
		rd.seq = std::string_view(v.seq, (size_t)v.rlen);
		rd.qual = std::string_view(v.qual ? v.qual : "", (size_t)ql);
		return;
	}
	char *sq = arena, *qq = arena + v.rlen;
	arena += 2 * (size_t)v.rlen;
	for (int i = 0; i < v.rlen; ++i) sq[i] = comp_base(v.seq[v.rlen - 1 - i]);
	for (int i = 0; i < ql; ++i) qq[i] = v.qual[ql - 1 - i];
	rd.seq = std::string_view(sq, (size_t)v.rlen);
	rd.qual = std::string_view(qq, (size_t)ql);
}

// GetNextChunk over mapped files: the same loop as next_chunk(), producing views
int next_chunk_views(const Ctx &cx, bool sep, MappedFile &f1, MappedFile &f2, std::vector<RecView> &views, int limit)
{
	int count = 0;
	RecView v;
	while (true) {
		if (!view_next(f1, v) || v.rlen == 0) break;
		v.flip = false;
		views.push_back(v);
		count++;
		bool ok = sep ? view_next(f2, v) : view_next(f1, v);
		if (!ok || v.rlen == 0) break;
		v.flip = cx.opt.paired;
		views.push_back(v);
		count++;
		if (count == limit) break;
	}
	return count;
}

bool is_fastq(const std::string &path)  // CheckReadFormat, src/GetData.cpp:8-16
{
	gzFile f = gzopen(path.c_str(), "rb");
	if (!f) return true;
	char c = 0;
	gzread(f, &c, 1);
	gzclose(f);
	return c == '@';
}

// ----------------------------------------------------------------------------------------------
// one library: batches of chunks (ReadMapping, src/Mapping.cpp:488-637)
// ----------------------------------------------------------------------------------------------
// The reference is only deterministic at -t 1, where chunk k sees EstDistance computed from the final
// alignments of chunks < k (:533-540).  To run chunks concurrently and still write exactly that output,
// every chunk is mapped with a SPECULATED EstDistance (the latest committed value) and records the
// interval of values for which its decisions hold; a commit step walks the chunks in input order,
// derives the true value from the committed totals and re-runs the (rare) chunk that falls outside.
struct ChunkState {
	int begin = 0, count = 0;
	bool paired = false;
	int est_used = 0;
	PairStats ps;
	Stats st;
	std::vector<std::vector<Candidate>> cands;
	std::vector<std::vector<CandWork>> work;
	NwJobs jobs;
	std::string text;
};

// persistent worker pool: stage after stage reuses the same threads (and their malloc arenas)
class Pool {
public:
	explicit Pool(int n) : n_(std::max(1, n))
	{
		for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t]() { loop(t); });
	}
	~Pool()
	{
		{
			std::lock_guard<std::mutex> lk(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		for (std::thread &th : workers_) th.join();
	}
	int size() const { return n_; }
	// two item sets in one parallel phase, both with the item -> worker (i mod n) mapping
	void run2(int n_a, const std::function<void(int)> &fa, int n_b, const std::function<void(int)> &fb)
	{
		int n = std::max(n_a, n_b);
		run(n, [&](int i) {
			if (i < n_a) fa(i);
			if (i < n_b) fb(i);
		});
	}
	// the same two sets, but every worker first does ALL its items of the first set, then calls `between` (which may
	// block), then does its items of the second set
	void run_ordered(int n_first, const std::function<void(int)> &f_first, const std::function<void()> &between, int n_second,
	                 const std::function<void(int)> &f_second)
	{
		if (n_first <= 0 && n_second <= 0) return;
		run(n_, [&](int w) {
			for (int i = w; i < n_first; i += n_) f_first(i);
			between();
			for (int i = w; i < n_second; i += n_) f_second(i);
		});
	}
	void run(int n_items, const std::function<void(int)> &fn)
	{
		if (n_items <= 0) return;
		if (n_ == 1 || n_items == 1) {
			for (int i = 0; i < n_items; ++i) fn(i);
			return;
		}
		{
			std::lock_guard<std::mutex> lk(mu_);
			fn_ = &fn; items_ = n_items; pending_ = n_ - 1; gen_++;
		}
		cv_.notify_all();
		// static assignment (item i -> worker i mod n): the chunk a worker built in one stage is the chunk it
		// finishes and frees in the next, so allocations never cross threads
		for (int i = 0; i < n_items; i += n_) fn(i);
		std::unique_lock<std::mutex> lk(mu_);
		done_cv_.wait(lk, [this]() { return pending_ == 0; });
		fn_ = nullptr;
	}

private:
	void loop(int me)
	{
		uint64_t seen = 0;
		for (;;) {
			const std::function<void(int)> *fn;
			int items;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&]() { return stop_ || gen_ != seen; });
				if (stop_) return;
				seen = gen_;
				fn = fn_; items = items_;
			}
			for (int i = me; i < items; i += n_) (*fn)(i);
			{
				std::lock_guard<std::mutex> lk(mu_);
				if (--pending_ == 0) done_cv_.notify_one();
			}
		}
	}
	int n_;
	std::vector<std::thread> workers_;
	std::mutex mu_;
	std::condition_variable cv_, done_cv_;
	const std::function<void(int)> *fn_ = nullptr;
	int items_ = 0, pending_ = 0;
	uint64_t gen_ = 0;
	bool stop_ = false;
};

// Asynchronous SAM output.  The chunks arrive in order, so each one's file offset is known when it is pushed: on a
// seekable file several threads copy into the page cache at once (pwrite) -- one thread's ~2.5 GB/s was the end of the
// run for E. coli-sized inputs (2.6 GB of SAM per 8 M reads) -- otherwise (pipe) one thread writes sequentially.
class Writer {
public:
	explicit Writer(FILE *out, int n_threads = 4) : fd_(fileno(out))
	{
		fflush(out);
		off_ = lseek(fd_, 0, SEEK_CUR);
		seekable_ = off_ >= 0;
		if (!seekable_) n_threads = 1;
		for (int t = 0; t < n_threads; ++t) th_.emplace_back([this]() { loop(); });
	}
	~Writer() { finish(); }
	void push(std::string &&text)
	{
		std::lock_guard<std::mutex> lk(mu_);
		Item it;
		it.off = off_;
		if (seekable_) off_ += (off_t)text.size();
		it.text = std::move(text);
		q_.push_back(std::move(it));
		cv_.notify_one();
	}
	void finish()
	{
		if (th_.empty()) return;
		{
			std::lock_guard<std::mutex> lk(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		for (std::thread &t : th_) t.join();
		th_.clear();
		if (seekable_) lseek(fd_, off_, SEEK_SET);
	}

private:
	struct Item { std::string text; off_t off; };
	void loop()
	{
		for (;;) {
			Item it;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [this]() { return stop_ || !q_.empty(); });
				if (q_.empty()) return;
				it = std::move(q_.front());
				q_.pop_front();
			}
			const char *p = it.text.data();
			size_t left = it.text.size();
			off_t at = it.off;
			while (left > 0) {
				ssize_t w = seekable_ ? ::pwrite(fd_, p, left, at) : ::write(fd_, p, left);
				if (w <= 0) { perror("write"); exit(1); }
				p += w; left -= (size_t)w; at += w;
			}
		}
	}
	int fd_;
	off_t off_ = 0;
	bool seekable_ = false;
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<Item> q_;
	bool stop_ = false;
	std::vector<std::thread> th_;
};

int est_distance(const Ctx &cx, int64_t iPaired, int64_t iDistance)  // src/Mapping.cpp:534-539
{
	if (iPaired < 1000) return cx.opt.max_insert;
	int est = (int)(iDistance / (iPaired >> 2));
	return est + (est >> 1);
}

// KART_AMD_VERBOSE: thread-seconds per section of the two chunk stages (summed over all workers)
std::atomic<int64_t> g_sec_ns[6];
const char *const g_sec_name[6] = {"unpack candidates", "pair+rescue+filters", "report plan", "report finish", "pair check+flags+mapq", "sam text"};
bool g_sections = false;
struct Section {
	int id;
	timespec t0;
	explicit Section(int i) : id(i) { if (g_sections) clock_gettime(CLOCK_MONOTONIC, &t0); }
	~Section()
	{
		if (!g_sections) return;
		timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		g_sec_ns[id] += (int64_t)(t1.tv_sec - t0.tv_sec) * 1000000000 + (t1.tv_nsec - t0.tv_nsec);
	}
};

// stage A: chaining, pairing, rescue, filters, report pass 1 (collects the chunk's NW jobs)
void chunk_stage_a(const Ctx &cx, std::vector<Read> &reads, const std::vector<int64_t> &cand_off, const std::vector<int32_t> &n_cands,
                   const std::vector<kg_candidate> &dev_cands, const std::vector<kg_seed> &cand_seeds, ChunkState &ck, int est)
{
	ck.est_used = est;
	ck.ps = PairStats();
	ck.st = Stats();
	ck.cands.assign((size_t)ck.count, std::vector<Candidate>());
	ck.work.assign((size_t)ck.count, std::vector<CandWork>());
	ck.jobs.clear();
	{ Section sec(0);
	for (int q = 0; q < ck.count; ++q) {
		size_t ri = (size_t)(ck.begin + q);
		// the candidates were chained on the device (kg_candidates_batch); unpack them into the per-read vectors
		std::vector<Candidate> &out = ck.cands[(size_t)q];
		out.resize((size_t)n_cands[ri]);
		for (int c = 0; c < n_cands[ri]; ++c) {
			const kg_candidate &d = dev_cands[(size_t)cand_off[ri] + (size_t)c];
			Candidate &o = out[(size_t)c];
			o.score = d.score;
			o.posDiff = d.posDiff;
			o.pairs.reserve((size_t)d.count * 2 + 3);     // room for the gap pairs identify_normal_pairs adds later
			o.pairs.resize((size_t)d.count);
			for (int k = 0; k < d.count; ++k) o.pairs[(size_t)k] = from_seed(cand_seeds[(size_t)d.first + (size_t)k]);
		}
	}
	}
	{ Section sec(1);
	if (ck.paired) {
		for (int q = 0; q < ck.count; q += 2) {
			std::vector<Candidate> &v1 = ck.cands[(size_t)q], &v2 = ck.cands[(size_t)q + 1];
			Read &r1 = reads[(size_t)(ck.begin + q)], &r2 = reads[(size_t)(ck.begin + q + 1)];
			bool pairing = pair_candidates(cx, est, v1, v2, ck.ps);
			if (!pairing) {
				ck.ps.rescue_used = true;
				pairing = rescue_unpaired(cx, est, r1, r2, v1, v2);
			}
			if (pairing) remove_unmated(v1, v2);
			remove_redundant(cx, v1);
			remove_redundant(cx, v2);
		}
	} else {
		for (int q = 0; q < ck.count; ++q) remove_redundant(cx, ck.cands[(size_t)q]);
	}
	}
	Section sec(2);
	// the plan compares read fragments with the reference at the candidates' positions -- random addresses in 6.2 GB for
	// hg38: ask for the lines of the reads a few places ahead while this one is being planned
	auto prefetch_ref = [&](int q) {
		if (q >= ck.count) return;
		for (const Candidate &c : ck.cands[(size_t)q])
			for (const Pair &p : c.pairs) {
				__builtin_prefetch(cx.refseq() + p.gPos);
				__builtin_prefetch(cx.refseq() + p.gPos + p.gLen + 64);
			}
	};
	for (int q = 0; q < 6 && q < ck.count; ++q) prefetch_ref(q);
	for (int q = 0; q < ck.count; ++q) { prefetch_ref(q + 6); report_plan(cx, reads[(size_t)(ck.begin + q)], ck.cands[(size_t)q], ck.work[(size_t)q], ck.jobs); }
}

// stage C: report pass 2, final pair check, flags, MAPQ, SAM text
void chunk_stage_c(const Ctx &cx, std::vector<Read> &reads, ChunkState &ck)
{
	{ Section sec(3);
	for (int q = 0; q < ck.count; ++q) {
		bool first = ck.paired ? (q % 2 == 0) : true;
		report_finish(cx, first, reads[(size_t)(ck.begin + q)], ck.cands[(size_t)q], ck.work[(size_t)q], ck.jobs);
	}
	}
	ck.text.clear();
	ck.text.reserve((size_t)ck.count * 400);
	if (ck.paired) {
		{ Section sec(4);
		for (int q = 0; q < ck.count; q += 2) {
			Read &r1 = reads[(size_t)(ck.begin + q)], &r2 = reads[(size_t)(ck.begin + q + 1)];
			check_final_pair(cx, r1, r2);
			set_paired_flags(r1, r2);
			evaluate_mapq(cx, r1);
			evaluate_mapq(cx, r2);
		}
		}
		Section sec(5);
		for (int q = 0; q < ck.count; q += 2)
			output_pair(cx, reads[(size_t)(ck.begin + q)], reads[(size_t)(ck.begin + q + 1)], ck.st, ck.ps, ck.text);
	} else {
		for (int q = 0; q < ck.count; ++q) {
			Read &rd = reads[(size_t)(ck.begin + q)];
			set_single_flag(rd);
			evaluate_mapq(cx, rd);
		}
		for (int q = 0; q < ck.count; ++q) output_single(cx, reads[(size_t)(ck.begin + q)], ck.st, ck.text);
	}
	ck.st.total_reads = ck.count;
	ck.cands.clear(); ck.work.clear();
	// the reports are spent once the text exists: release them here, on the worker that allocated them, instead of in the
	// batch destructor on the main thread (400 k small frees per batch, serial)
	for (int q = 0; q < ck.count; ++q) std::vector<Report>().swap(reads[(size_t)(ck.begin + q)].rep);
}

// one NW kernel call for the jobs of many chunks
void run_nw(const Ctx &cx, std::vector<ChunkState> &chunks, size_t from, size_t to)
{
	std::vector<NwJobs *> parts;
	for (size_t c = from; c < to; ++c)
		if (chunks[c].jobs.size() > 0) parts.push_back(&chunks[c].jobs);
	if (!parts.empty()) cx.kern.nw_batch(parts);
}

struct RunTotals {
	int64_t iPaired = 0, iDistance = 0;   // src/Mapping.cpp:13,20
	double t_read = 0, t_encode = 0, t_seed = 0, t_a = 0, t_nw = 0, t_c = 0, t_commit = 0, t_drain = 0, t_lib = 0;   // KART_AMD_VERBOSE stage timers
};

double now_s()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct Batch {
	std::vector<Read> reads;
	std::vector<ChunkState> chunks;
	std::vector<uint8_t> enc;                       // the read characters, concatenated (encoded on the device)
	std::vector<int64_t> off;
	std::vector<int64_t> cand_off;                  // per-read candidate ranges of this batch (filled by the seeding + chaining stage)
	std::vector<int32_t> n_cands;                   // chaining results of this batch (kg_candidates_batch)
	std::vector<kg_candidate> cands;
	std::vector<kg_seed> cand_seeds;
	std::deque<OwnedRead> owned;                    // storage behind the views (getline()/gzgets() readers)
	std::vector<std::unique_ptr<char[]>> arenas;    // storage behind reverse-complemented mates (mapped files)
	std::vector<char> text1, text2;                 // inflated FASTQ text behind the views (gz files)
	bool eof = false;
	double seconds = 0, seed_seconds = 0;
};

// gzipped FASTQ: the text is inflated a batch at a time (both mate files in parallel) into buffers the batch owns and then
// parsed by the same view code as a mapped file; what is left after the last whole chunk is carried into the next batch
struct GzText {
	gzFile f = nullptr;
	std::vector<char> carry;
	bool eof = false;
	double bytes_per_record = 0;
	void fill(std::vector<char> &buf, size_t want)
	{
		size_t at = buf.size(), got = 0;
		buf.resize(at + want);
		while (got < want) {
			int n = gzread(f, buf.data() + at + got, (unsigned)std::min<size_t>(want - got, (size_t)1 << 30));
			if (n <= 0) { eof = true; break; }
			got += (size_t)n;
		}
		buf.resize(at + got);
	}
};

struct Source {
	bool sep = false, fast = false, gzfast = false;
	Input in1, in2;            // getline()/gzgets() readers (FASTA, gz)
	MappedFile m1, m2;         // mapped plain FASTQ, or the inflated text of the current batch
	GzText g1, g2;
};

// reads `batch_chunks` whole chunks (GetNextChunk each); runs on the prefetch thread
void read_batch(const Ctx &cx, Source &src, int64_t batch_chunks, int chunk_limit, Pool &pool, Batch &b)
{
	double t0 = now_s();
	b.reads.clear(); b.chunks.clear(); b.owned.clear(); b.arenas.clear(); b.eof = false;
	std::vector<RecView> views;
	int64_t parse_chunks = batch_chunks;
	if (src.gzfast) {
		const size_t recs_per_file = (size_t)batch_chunks * (size_t)chunk_limit / (src.sep ? 2 : 1);
		b.text1.swap(src.g1.carry); src.g1.carry.clear();
		b.text2.swap(src.g2.carry); src.g2.carry.clear();
		auto want = [&](const GzText &g, const std::vector<char> &t) {
			size_t est = (size_t)((double)recs_per_file * (g.bytes_per_record > 0 ? g.bytes_per_record * 1.03 : 360.0)) + (1 << 16);
			return est > t.size() ? est - t.size() : (size_t)0;
		};
		size_t need1 = want(src.g1, b.text1), need2 = src.sep ? want(src.g2, b.text2) : 0;
		for (;;) {
			std::future<void> other;
			if (src.sep && !src.g2.eof && need2) other = std::async(std::launch::async, [&]() { src.g2.fill(b.text2, need2); });
			if (!src.g1.eof && need1) src.g1.fill(b.text1, need1);
			if (other.valid()) other.get();
			// the mate files end together or not at all: once one is exhausted the rest of the other is needed
			if (src.sep && src.g1.eof != src.g2.eof) {
				GzText &g = src.g1.eof ? src.g2 : src.g1;
				std::vector<char> &t = src.g1.eof ? b.text2 : b.text1;
				while (!g.eof) g.fill(t, (size_t)64 << 20);
			}
			src.m1.attach(b.text1.data(), b.text1.size());
			src.m1.index_ahead(pool, b.text1.size());
			size_t recs1 = src.m1.line_end.size() / 4, recs2 = 0;
			if (src.sep) {
				src.m2.attach(b.text2.data(), b.text2.size());
				src.m2.index_ahead(pool, b.text2.size());
				recs2 = src.m2.line_end.size() / 4;
			}
			bool final = src.g1.eof && (!src.sep || src.g2.eof);
			if (final) break;                               // everything is in memory: parse to the end
			size_t reads_avail = src.sep ? 2 * std::min(recs1, recs2) : (recs1 & ~(size_t)1);
			parse_chunks = std::min<int64_t>(batch_chunks, (int64_t)(reads_avail / (size_t)chunk_limit));
			if (parse_chunks >= 1) break;
			need1 = need2 = std::max<size_t>((size_t)1 << 20, recs_per_file * 64);   // not even one whole chunk yet: more text
		}
	}
	if (src.fast) {
		// roughly the bytes this batch will consume(header + 2 x read + "+"), indexed in parallel
		size_t per_file = (size_t)batch_chunks * (size_t)chunk_limit * 400 / (src.sep ? 2 : 1) + (1 << 20);
		src.m1.index_ahead(pool, per_file);
		if (src.sep) src.m2.index_ahead(pool, per_file);
	}
	const bool by_views = src.fast || src.gzfast;
	if (by_views) {
		// whole chunks straight from the line index (GetNextChunk's loop over precomputed views); whatever the index does
		// not cover -- the tail of a file, a last line without newline -- is left to the line-by-line walk below
		std::vector<RecView> v1, v2;
		views_of_indexed_records(pool, src.m1, v1);
		if (src.sep) views_of_indexed_records(pool, src.m2, v2);
		size_t i1 = 0, i2 = 0;
		const size_t per_chunk = src.sep ? (size_t)(chunk_limit + 1) / 2 : (size_t)chunk_limit + 1;   // records a chunk can take from a file
		while ((int64_t)b.chunks.size() < parse_chunks && i1 + per_chunk <= v1.size() && (!src.sep || i2 + per_chunk <= v2.size())) {
			ChunkState ck;
			ck.begin = (int)views.size();
			int count = 0;
			for (;;) {                                        // next_chunk_views() on the precomputed views
				RecView v = v1[i1++];
				if (v.rlen == 0) break;
				v.flip = false;
				views.push_back(v);
				count++;
				v = src.sep ? v2[i2++] : v1[i1++];
				if (v.rlen == 0) break;
				v.flip = cx.opt.paired;
				views.push_back(v);
				count++;
				if (count == chunk_limit) break;
			}
			ck.count = count;
			if (count == 0) { b.eof = true; break; }          // an empty record first ends the library, as in the walk below
			ck.paired = cx.opt.paired && ck.count % 2 == 0 && !cx.opt.pacbio;
			b.chunks.push_back(std::move(ck));
		}
		auto advance = [](MappedFile &f, size_t recs) {
			if (recs == 0) return;
			f.pos = f.line_end[4 * recs - 1];
			f.next_line = 4 * recs;
		};
		advance(src.m1, i1);
		if (src.sep) advance(src.m2, i2);
	}
	while (!b.eof && (int64_t)b.chunks.size() < parse_chunks) {
		ChunkState ck;
		ck.begin = by_views ? (int)views.size() : (int)b.owned.size();
		ck.count = by_views ? next_chunk_views(cx, src.sep, src.m1, src.m2, views, chunk_limit)
		                    : next_chunk(cx, src.sep, src.in1, src.in2, b.owned, chunk_limit);
		if (ck.count == 0) { b.eof = true; break; }
		ck.paired = cx.opt.paired && ck.count % 2 == 0 && !cx.opt.pacbio;
		b.chunks.push_back(std::move(ck));
	}
	if (src.gzfast) {   // what the parser did not reach belongs to the next batch
		size_t n1 = views.size() / (src.sep ? 2 : 1);
		if (n1 > 0) src.g1.bytes_per_record = (double)src.m1.pos / (double)((views.size() + (src.sep ? 1 : 0)) / (src.sep ? 2 : 1));
		if (src.sep && views.size() > 1) src.g2.bytes_per_record = (double)src.m2.pos / (double)(views.size() / 2);
		src.g1.carry.assign(b.text1.begin() + (std::ptrdiff_t)src.m1.pos, b.text1.end());
		if (src.sep) src.g2.carry.assign(b.text2.begin() + (std::ptrdiff_t)src.m2.pos, b.text2.end());
	}
	if (by_views) {
		b.reads.resize(views.size());
		int blocks = (int)((views.size() + 2047) / 2048);
		b.arenas.resize((size_t)blocks);
		pool.run(blocks, [&](int blk) {
			size_t lo = (size_t)blk * 2048, hi = std::min(views.size(), lo + 2048), need = 0;
			for (size_t i = lo; i < hi; ++i)
				if (views[i].flip) need += 2 * (size_t)views[i].rlen;
			b.arenas[(size_t)blk].reset(new char[need + 1]);
			char *arena = b.arenas[(size_t)blk].get();
			for (size_t i = lo; i < hi; ++i) materialise(views[i], b.reads[i], arena);
		});
	} else {
		b.reads.resize(b.owned.size());
		for (size_t i = 0; i < b.owned.size(); ++i) {
			const OwnedRead &o = b.owned[i];
			Read &rd = b.reads[i];
			rd = Read();
			rd.name = o.name; rd.seq = o.seq; rd.qual = o.qual; rd.rlen = o.rlen;
		}
	}
	// The reads of the batch, concatenated, as characters: EnCodeReadSeq (src/Mapping.cpp:482-485) itself runs on the device
	// (KG_INPUT_ASCII).  The reference encodes mate 2 with mate 1's length (:550, App. B-5); with equal-length mates that is the
	// same thing, otherwise it reads or leaves uninitialised bytes -- here every read is taken over its own length.
	std::vector<Read> &reads = b.reads;
	b.off.assign(reads.size() + 1, 0);
	for (size_t i = 0; i < reads.size(); ++i) b.off[i + 1] = b.off[i] + reads[i].rlen;
	b.enc.resize((size_t)b.off[reads.size()]);
	pool.run((int)((reads.size() + 4095) / 4096), [&](int blk) {
		size_t lo = (size_t)blk * 4096, hi = std::min(reads.size(), lo + 4096);
		for (size_t i = lo; i < hi; ++i) memcpy(b.enc.data() + b.off[i], reads[i].seq.data(), (size_t)reads[i].rlen);
	});
	b.seconds = now_s() - t0;
}

void map_library(Ctx &cx, Source &src, FILE *out, Stats &st, RunTotals &tot)
{
	const int chunk_limit = cx.opt.pacbio ? 10 : 4000;   // ReadChunkSize, src/structure.h:21; src/GetData.cpp:140
	const int mode = cx.opt.pacbio ? KG_MODE_SENSITIVE : KG_MODE_FAST;
	const int nthreads = std::max(1, cx.opt.threads);
	// Three batches in flight:
	//   prefetch thread : read + encode + seed (GPU) of batch k+1
	//   worker pool     : ONE combined phase -- chain/pair/plan the chunks of batch k, then finish/format the chunks of
	//                     batch k-1, chunk c of either batch on worker c mod n, so whatever a worker allocates for a
	//                     chunk it also frees (no cross-thread frees)
	//   helper thread   : NW kernel call of batch k, overlapping the commit of k-1 and the planning of k+1
	//   main thread     : the in-order commit of batch k-1
	// The speculated EstDistance therefore lags the committed totals by up to two batches; the commit's
	// validity check absorbs that.
	// parsing + encoding costs ~0.4x what the mapping stages cost per read: half as many reader threads keep up
	Pool pool(nthreads), read_pool((nthreads + 1) / 2);
	Writer writer(out);
	// small batches first: the estimate moves fastest while the totals are small
	int64_t batch_chunks = 1;
	// enough chunks per batch to keep every worker busy (static chunk -> worker assignment)
	// ... and a whole number of chunks per worker, so that the last round of a phase is not half empty
	// batches are sized in bases: 400 k short reads, or ~10 k long reads (1024 chunks of 10) -- enough to fill the GPU and the
	// workers, small enough that three batches in flight stay within a few GB and that seeding overlaps the mapping
	int64_t want_chunks = cx.opt.pacbio ? std::max<int64_t>(1024, 8 * nthreads)
	                                     : std::max<int64_t>(std::max<int64_t>(1, cx.opt.batch_reads / chunk_limit), 4 * nthreads);
	const int64_t max_batch_chunks = (want_chunks + nthreads - 1) / nthreads * nthreads;
	if (!cx.opt.paired || cx.opt.pacbio) batch_chunks = max_batch_chunks;   // no EstDistance feedback to settle: full batches at once
	std::unique_ptr<Batch> cur(new Batch()), nxt(new Batch()), prev;
	std::shared_future<void> nw_prev;      // the gap-closing kernel call of `prev`, running on its own thread

	auto fetch = [&](Batch *b, int64_t n_chunks) {
		read_batch(cx, src, n_chunks, chunk_limit, read_pool, *b);
		double t = now_s();
		if (!b->reads.empty()) {
			cx.kern.seed_and_chain(mode, cx.opt.pacbio, cx.opt.max_gaps, b->enc, b->off, b->n_cands, b->cand_off, b->cands, b->cand_seeds);
		}
		b->seed_seconds = now_s() - t;
	};
	auto commit = [&](Batch &b) {   // in-order: EstDistance feeds forward (src/Mapping.cpp:533-540)
		double t0 = now_s();
		// A chunk mapped under a speculated EstDistance stands if the true estimate -- a function of the totals of
		// all chunks before it -- would have decided every pair the same way.  Walk the batch with running totals,
		// collect the chunks whose speculation does not hold, re-map those together on the pool, and repeat: a
		// re-mapped chunk can move the estimates after it.  Every round settles at least the first unsettled chunk,
		// so the fixed point is the sequential result.
		for (;;) {
			std::vector<std::pair<size_t, int>> redo;
			int64_t paired = tot.iPaired, distance = tot.iDistance;
			for (size_t c = 0; c < b.chunks.size(); ++c) {
				ChunkState &ck = b.chunks[c];
				if (ck.paired) {
					int est_true = est_distance(cx, paired, distance);
					bool valid = est_true == ck.est_used ||
					             (ck.ps.lo < est_true && est_true <= ck.ps.hi &&
					              (!ck.ps.rescue_used || std::min(est_true, cx.opt.max_insert) == std::min(ck.est_used, cx.opt.max_insert)));
					if (!valid) redo.emplace_back(c, est_true);
				}
				paired += ck.ps.paired;
				distance += ck.ps.distance;
			}
			if (redo.empty()) break;
			st.respeculated += (int64_t)redo.size();
			pool.run((int)redo.size(), [&](int i) {
				chunk_stage_a(cx, b.reads, b.cand_off, b.n_cands, b.cands, b.cand_seeds, b.chunks[redo[(size_t)i].first], redo[(size_t)i].second);
			});
			std::vector<NwJobs *> parts;
			for (const std::pair<size_t, int> &r : redo)
				if (b.chunks[r.first].jobs.size() > 0) parts.push_back(&b.chunks[r.first].jobs);
			if (!parts.empty()) cx.kern.nw_batch(parts);
			pool.run((int)redo.size(), [&](int i) { chunk_stage_c(cx, b.reads, b.chunks[redo[(size_t)i].first]); });
		}
		for (size_t c = 0; c < b.chunks.size(); ++c) {
			ChunkState &ck = b.chunks[c];
			writer.push(std::move(ck.text));
			tot.iPaired += ck.ps.paired;
			tot.iDistance += ck.ps.distance;
			st.total_reads += ck.st.total_reads;
			st.unmapped += ck.st.unmapped;
			st.unique += ck.st.unique;
		}
		tot.t_commit += now_s() - t0;
	};

	fetch(cur.get(), batch_chunks);
	tot.t_read += cur->seconds;
	tot.t_seed += cur->seed_seconds;
	while (!cur->reads.empty() || prev) {
		bool have_cur = !cur->reads.empty();
		batch_chunks = std::min(max_batch_chunks, batch_chunks * 2);
		std::future<void> prefetch;
		bool more = have_cur && !cur->eof;
		Batch *np = nxt.get();
		static const bool inline_fetch = getenv("KART_AMD_NO_PREFETCH") != nullptr;   // profiling aid: everything on the main thread
		if (more) prefetch = std::async(inline_fetch ? std::launch::deferred : std::launch::async, [&, batch_chunks, np]() { fetch(np, batch_chunks); });
		int est_guess = est_distance(cx, tot.iPaired, tot.iDistance);
		Batch *cp = cur.get(), *pp = prev.get();
		double t3 = now_s();
		// one phase on the pool: plan the chunks of batch k, then -- once the gap-closing kernel call of batch k-1, which has
		// been running on its own thread since the last phase, is back -- finish and format the chunks of batch k-1
		std::atomic<int64_t> nw_wait_ns{0};
		pool.run_ordered(have_cur ? (int)cp->chunks.size() : 0,
		                 [&](int c) { chunk_stage_a(cx, cp->reads, cp->cand_off, cp->n_cands, cp->cands, cp->cand_seeds, cp->chunks[(size_t)c], est_guess); },
		                 [&]() {
			                 if (!nw_prev.valid()) return;
			                 double tw = now_s();
			                 nw_prev.wait();
			                 nw_wait_ns += (int64_t)((now_s() - tw) * 1e9);
		                 },
		                 pp ? (int)pp->chunks.size() : 0, [&](int c) { chunk_stage_c(cx, pp->reads, pp->chunks[(size_t)c]); });
		if (nw_prev.valid()) nw_prev.get();
		double t4 = now_s();
		tot.t_nw += 1e-9 * (double)nw_wait_ns.load() / (double)nthreads;   // average time a worker stood waiting for the kernel call
		tot.t_a += t4 - t3;
		if (have_cur) nw_prev = std::async(std::launch::async, [&cx, cp]() { run_nw(cx, cp->chunks, 0, cp->chunks.size()); }).share();
		else nw_prev = std::shared_future<void>();
		if (pp) commit(*pp);
		prev = have_cur ? std::move(cur) : nullptr;
		if (more) {
			double tw = now_s();
			prefetch.get();
			tot.t_read += now_s() - tw;   // only the part of read + encode + seed that was not hidden
			tot.t_seed += nxt->seed_seconds;
			cur = std::move(nxt);
			nxt.reset(new Batch());
		} else {
			cur.reset(new Batch());
		}
	}
	double td = now_s();
	writer.finish();
	tot.t_drain += now_s() - td;
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// public entry points
// ----------------------------------------------------------------------------------------------
bool RefData::load(const std::string &prefix, std::string &err, int threads)
{
	FILE *fp = fopen((prefix + ".ann").c_str(), "r");
	if (!fp) { err = "cannot read " + prefix + ".ann"; return false; }
	long long l_pac;
	int n_seqs;
	unsigned seed;
	if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(fp); err = "bad .ann header"; return false; }
	genome_size = l_pac;
	two_genome_size = 2 * genome_size;
	int64_t total = 0;
	for (int i = 0; i < n_seqs; ++i) {
		unsigned gi;
		char name[1024];
		long long offv;
		int len, n_ambs, ch;
		if (fscanf(fp, "%u%1023s", &gi, name) != 2) { fclose(fp); err = "bad .ann record"; return false; }
		while ((ch = fgetc(fp)) != '\n' && ch != EOF) {}
		if (fscanf(fp, "%lld%d%d", &offv, &len, &n_ambs) != 3) { fclose(fp); err = "bad .ann record"; return false; }
		Contig c;
		c.name = name;
		c.len = len;
		c.fwd_start = total;
		total += len;
		c.rev_start = two_genome_size - total;
		chr_end[c.fwd_start + c.len - 1] = i;
		chr_end[c.rev_start + c.len - 1] = i;
		contigs.push_back(c);
	}
	fclose(fp);
	std::vector<unsigned char> pac;
	if (!slurp(prefix + ".pac", pac) || (int64_t)pac.size() < genome_size / 4 + 1) { err = "cannot read " + prefix + ".pac"; return false; }
	// both strands as characters (src/bwt_index.cpp:242-258), one .pac byte = four bases at a time; the 2L bytes
	// are first touched by the decoding threads themselves (6.2 GB for hg38)
	// random access all over 6.2 GB (hg38): transparent huge pages keep the report stage out of the page walker
	{
		void *mem = nullptr;
		size_t bytes = (((size_t)two_genome_size + 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
		if (posix_memalign(&mem, (size_t)2 << 20, bytes) != 0) { err = "out of memory for the reference sequence"; return false; }
		madvise(mem, bytes, MADV_HUGEPAGE);
		seq.reset((char *)mem);
	}
	seq[(size_t)two_genome_size] = '\0';
	std::vector<uint32_t> fw4(256), rc4(256);
	for (int v = 0; v < 256; ++v) {
		char f[4], r[4];
		for (int j = 0; j < 4; ++j) {
			int b = (v >> ((3 - j) << 1)) & 3;       // base j of the byte (first base in the top bits)
			f[j] = "ACGT"[b];
			r[3 - j] = "TGCA"[b];                    // the reverse strand runs the other way
		}
		memcpy(&fw4[(size_t)v], f, 4);
		memcpy(&rc4[(size_t)v], r, 4);
	}
	int64_t whole = genome_size >> 2;                // bytes whose four bases all exist
	int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, whole >> 20));
	std::vector<std::thread> pool;
	char *out = seq.get();
	for (int t = 0; t < nt; ++t)
		pool.emplace_back([&, t]() {
			for (int64_t i = whole * t / nt, e = whole * (t + 1) / nt; i < e; ++i) {
				memcpy(out + (i << 2), &fw4[pac[(size_t)i]], 4);
				memcpy(out + (two_genome_size - (i << 2) - 4), &rc4[pac[(size_t)i]], 4);
			}
		});
	for (std::thread &th : pool) th.join();
	static const char fw[4] = {'A', 'C', 'G', 'T'}, rc[4] = {'T', 'G', 'C', 'A'};
	for (int64_t f = whole << 2; f < genome_size; ++f) {
		int b = pac[(size_t)(f >> 2)] >> ((~f & 3) << 1) & 3;
		seq[(size_t)f] = fw[b];
		seq[(size_t)(two_genome_size - 1 - f)] = rc[b];
	}
	return true;
}

int run_mapping(const Options &opt, const RefData &ref, KernelBackend &kern, FILE *out, Stats &stats)
{
	// keep freed memory inside the arenas: the per-read strings/vectors of one batch are reused by the next,
	// and handing pages back to the kernel serialises every worker on the address-space lock
	mallopt(M_MMAP_THRESHOLD, 1 << 30);
	mallopt(M_TRIM_THRESHOLD, -1);
	mallopt(M_TOP_PAD, 64 << 20);
	double t_begin = now_s();
	g_sections = getenv("KART_AMD_VERBOSE") != nullptr;
	Ctx cx{opt, ref, kern, kern.min_seed_len()};
	Options &o = const_cast<Options &>(opt);
	RunTotals tot;
	// header: @PG first, then @SQ, no @HD (src/Mapping.cpp:664-675)
	fprintf(out, "@PG\tID:kart\tPN:Kart\tVN:%s\n", "2.5.6");
	for (size_t i = 0; i < ref.contigs.size(); ++i) fprintf(out, "@SQ\tSN:%s\tLN:%lld\n", ref.contigs[i].name.c_str(), (long long)ref.contigs[i].len);
	for (size_t lib = 0; lib < opt.files1.size(); ++lib) {
		const std::string &f1 = opt.files1[lib];
		bool gz = f1.size() >= 2 && f1.substr(f1.find_last_of('.') + 1) == "gz";   // src/Mapping.cpp:688
		cx.fastq = is_fastq(f1);
		Source src;
		Input &in1 = src.in1, &in2 = src.in2;
		bool want_fast = !gz && cx.fastq && !getenv("KART_AMD_NO_MMAP");
		if (gz) in1.gz = gzopen(f1.c_str(), "rb"); else in1.fp = fopen(f1.c_str(), "r");
		bool sep = false;
		if (opt.files1.size() == opt.files2.size()) {
			sep = true;
			o.paired = true;
			const std::string &f2 = opt.files2[lib];
			if (cx.fastq != is_fastq(f2)) {
				fprintf(stdout, "Error! %s and %s are with different format...\n", f1.c_str(), f2.c_str());
				continue;
			}
			if (gz) in2.gz = gzopen(f2.c_str(), "rb"); else in2.fp = fopen(f2.c_str(), "r");
		}
		if (!in1.fp && !in1.gz) continue;
		if (sep && !in2.fp && !in2.gz) continue;
		src.sep = sep;
		src.fast = want_fast && src.m1.open(f1) && (!sep || src.m2.open(opt.files2[lib]));
		src.gzfast = gz && cx.fastq && !getenv("KART_AMD_NO_MMAP");
		if (src.gzfast) { src.g1.f = in1.gz; src.g2.f = in2.gz; gzbuffer(in1.gz, 1 << 20); if (in2.gz) gzbuffer(in2.gz, 1 << 20); }
		double tl = now_s();
		map_library(cx, src, out, stats, tot);
		tot.t_lib += now_s() - tl;
	}
	stats.paired = tot.iPaired;
	stats.distance = tot.iDistance;
	stats.map_seconds = now_s() - t_begin;
	if (getenv("KART_AMD_VERBOSE"))
		fprintf(stdout, "stage seconds: unhidden read+encode+seed %.2f (seed calls %.2f) | finish+format(k-1) with chain+pair+plan(k) %.2f | nw %.2f | commit %.2f | writer drain %.2f | libraries %.2f of %.2f\n",
		        tot.t_read, tot.t_seed, tot.t_a, tot.t_nw, tot.t_commit, tot.t_drain, tot.t_lib, stats.map_seconds);
	if (g_sections) {
		fprintf(stdout, "worker thread-seconds:");
		for (int i = 0; i < 6; ++i) fprintf(stdout, " %s %.2f%s", g_sec_name[i], 1e-9 * (double)g_sec_ns[i].load(), i < 5 ? " |" : "\n");
	}
	return 0;
}

}  // namespace kart
