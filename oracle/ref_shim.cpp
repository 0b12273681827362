// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C wrapper over the UNMODIFIED reference functions in oracle/_ref/libkartref.so so that
// oracle/pin_against_ref.py can call them from Python (they take/return std::vector and
// std::string, which ctypes cannot express).  The reference's own header is included from
// where it lies (-I/root/reference/src); nothing from the reference is copied.  Built by
// `make -C oracle ref` into oracle/_ref/libkartref_shim.so; only usable in a container where
// /root/reference exists at build time.
#include "structure.h"  // reference src/structure.h (included in place, not copied)

#include <cstring>
#include <thread>
#include <vector>

// exported (non-static) functions of src/bwt_search.cpp:44,68,128 that structure.h does not declare
extern bwtint_t bwt_occ(const bwt_t *bwt, bwtint_t k, ubyte_t c);
extern void bwt_occ4(const bwt_t *bwt, bwtint_t k, bwtint_t cnt[4]);
extern bwtint_t bwt_sa(bwtint_t k);
extern void RemoveRedundantCandidates(vector<AlignmentCandidate_t> &AlignmentVec);  // src/Mapping.cpp:317
extern void GenerateNormalPairAlignment(int rLen, string &frag1, int gLen, string &frag2);   // src/tools.cpp:142 (exported, not in structure.h)

extern "C" {

struct shim_seed { int64_t gPos; int32_t rPos; int32_t len; };
struct shim_pair { int64_t gPos; int64_t PosDiff; int32_t rPos; int32_t rLen; int32_t gLen; int32_t bSimple; };

// main.cpp:192-207 without Mapping()
int shim_init(const char *prefix, int threads, int pacbio, int max_gaps)
{
	iThreadNum = threads; bPacBioData = pacbio != 0; MaxGaps = max_gaps; MaxInsertSize = 1500;
	bDebugMode = false; bMultiHit = false; bPairEnd = false;
	RefIdx = bwa_idx_load(prefix);
	if (RefIdx == 0) return -1;
	Refbwt = RefIdx->bwt;
	RestoreReferenceInfo();
	for (MinSeedLength = 13; MinSeedLength < 16; MinSeedLength++) if (TwoGenomeSize < pow(4, MinSeedLength)) break;  // Mapping.cpp:645
	return MinSeedLength;
}

void shim_set_mode(int pacbio, int max_gaps) { bPacBioData = pacbio != 0; MaxGaps = max_gaps; }
int64_t shim_genome_size() { return GenomeSize; }
uint64_t shim_primary() { return Refbwt->primary; }
uint64_t shim_seq_len() { return Refbwt->seq_len; }
const char *shim_ref_sequence() { return RefSequence; }


uint64_t shim_occ(uint64_t k, int c) { return bwt_occ(Refbwt, k, (ubyte_t)c); }
void shim_occ4(uint64_t k, uint64_t *cnt) { bwtint_t c[4]; bwt_occ4(Refbwt, k, c); for (int i = 0; i < 4; i++) cnt[i] = c[i]; }
uint64_t shim_sa(uint64_t k) { return bwt_sa(k); }

int shim_bwt_search(uint8_t *seq, int start, int stop, int *len, uint64_t *locs)
{
	bwtSearchResult_t r = BWT_Search(seq, start, stop);
	*len = r.len;
	for (int i = 0; i < r.freq; i++) locs[i] = r.LocArr[i];
	if (r.LocArr) delete[] r.LocArr;
	return r.freq;
}

int shim_seed_read(int mode, uint8_t *enc, int rlen, shim_seed *out, int cap)
{
	vector<SeedPair_t> v = mode == 0 ? IdentifySeedPairs_FastMode(rlen, enc) : IdentifySeedPairs_SensitiveMode(rlen, enc);
	if ((int)v.size() > cap) return -(int)v.size();
	for (size_t i = 0; i < v.size(); i++) { out[i].gPos = v[i].gPos; out[i].rPos = v[i].rPos; out[i].len = v[i].rLen; }
	return (int)v.size();
}

// IdentifySeedPairs_* (the reference's own object code) over a batch of reads on `threads` threads, the way its workers call
// it (src/Mapping.cpp:546-551): the CPU baseline of bench.py.  Returns the number of seeds found.
int64_t shim_seed_batch(int mode, uint8_t *enc, const int64_t *off, int64_t n, int threads)
{
	std::vector<std::thread> pool;
	std::vector<int64_t> found((size_t)threads, 0);
	for (int t = 0; t < threads; ++t)
		pool.emplace_back([&, t]() {
			int64_t cnt = 0;
			for (int64_t i = n * t / threads, e = n * (t + 1) / threads; i < e; ++i) {
				int rlen = (int)(off[i + 1] - off[i]);
				vector<SeedPair_t> v = mode == 0 ? IdentifySeedPairs_FastMode(rlen, enc + off[i]) : IdentifySeedPairs_SensitiveMode(rlen, enc + off[i]);
				cnt += (int64_t)v.size();
			}
			found[(size_t)t] = cnt;
		});
	for (std::thread &th : pool) th.join();
	int64_t total = 0;
	for (int64_t c : found) total += c;
	return total;
}

int shim_nw(const char *s1, int m, const char *s2, int n, char *out1, char *out2)
{
	string a(s1, m), b(s2, n);
	nw_alignment(m, a, n, b);
	memcpy(out1, a.c_str(), a.size() + 1);
	memcpy(out2, b.c_str(), b.size() + 1);
	return (int)a.size();
}

// GenerateNormalPairAlignment, src/tools.cpp:142 (under the mode shim_set_mode selected)
int shim_normal_pair_alignment(const char *s1, int m, const char *s2, int n, char *out1, char *out2)
{
	string a(s1, m), b(s2, n);
	GenerateNormalPairAlignment(m, a, n, b);
	memcpy(out1, a.c_str(), a.size() + 1);
	memcpy(out2, b.c_str(), b.size() + 1);
	return (int)a.size();
}

static vector<SeedPair_t> to_vec(const shim_seed *s, int n)
{
	vector<SeedPair_t> v((size_t)n);
	for (int i = 0; i < n; i++) {
		v[i].bSimple = true; v[i].rPos = s[i].rPos; v[i].gPos = s[i].gPos;
		v[i].rLen = v[i].gLen = s[i].len; v[i].PosDiff = s[i].gPos - s[i].rPos;
	}
	return v;
}

static int emit(vector<AlignmentCandidate_t> &c, int *cand_off, int *score, int64_t *posdiff, shim_pair *out, int cand_cap, int pair_cap)
{
	int total = 0;
	for (size_t i = 0; i < c.size(); i++) total += (int)c[i].SeedVec.size();
	if ((int)c.size() > cand_cap || total > pair_cap) return -1;
	int off = 0;
	for (size_t i = 0; i < c.size(); i++) {
		cand_off[i] = off; score[i] = c[i].Score; posdiff[i] = c[i].PosDiff;
		for (size_t j = 0; j < c[i].SeedVec.size(); j++) {
			SeedPair_t &p = c[i].SeedVec[j];
			out[off].gPos = p.gPos; out[off].PosDiff = p.PosDiff; out[off].rPos = p.rPos;
			out[off].rLen = p.rLen; out[off].gLen = p.gLen; out[off].bSimple = p.bSimple ? 1 : 0;
			off++;
		}
	}
	cand_off[c.size()] = off;
	return (int)c.size();
}

int shim_candidates(int pacbio, int rlen, const shim_seed *s, int n, int *cand_off, int *score, int64_t *posdiff, shim_pair *out, int cand_cap, int pair_cap)
{
	vector<AlignmentCandidate_t> c = pacbio ? GenerateAlignmentCandidateForPacBioSeq(rlen, to_vec(s, n)) : GenerateAlignmentCandidateForIlluminaSeq(rlen, to_vec(s, n));
	return emit(c, cand_off, score, posdiff, out, cand_cap, pair_cap);
}

int shim_identify_normal_pairs(int rlen, int glen, shim_pair *pairs, int n, int cap)
{
	vector<SeedPair_t> v((size_t)n);
	for (int i = 0; i < n; i++) {
		v[i].bSimple = pairs[i].bSimple != 0; v[i].rPos = pairs[i].rPos; v[i].gPos = pairs[i].gPos;
		v[i].rLen = pairs[i].rLen; v[i].gLen = pairs[i].gLen; v[i].PosDiff = pairs[i].PosDiff;
	}
	IdentifyNormalPairs(rlen, glen, v);
	if ((int)v.size() > cap) return -(int)v.size();
	for (size_t i = 0; i < v.size(); i++) {
		pairs[i].gPos = v[i].gPos; pairs[i].PosDiff = v[i].PosDiff; pairs[i].rPos = v[i].rPos;
		pairs[i].rLen = v[i].rLen; pairs[i].gLen = v[i].gLen; pairs[i].bSimple = v[i].bSimple ? 1 : 0;
	}
	return (int)v.size();
}

}  // extern "C"
