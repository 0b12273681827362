// align_kernels.hip -- the per-read report on the device (gfx950): pairing, normal pairs, gap closing, CIGAR, flags, MAPQ.
//
// Replaces, for the short-read configuration (neither -pacbio nor -m) and per batch, what ReadMapping() does for every read
// between chaining and the SAM text (reference src/Mapping.cpp:542-578):
//   aln_pair_kernel   CheckPairedAlignmentCandidates, RemoveUnMatedAlignmentCandidates, RemoveRedundantCandidates
//                     (src/Mapping.cpp:317-427); one read pair per lane.
//   aln_plan_kernel   GenMappingReport pass 1 (src/AlignmentCandidates.cpp:624-745), one candidate per lane: IdentifyNormalPairs
//                     with RemoveTandemRepeatSeeds / RemoveTranslocatedSeeds / CheckOverlappingSeeds (:226-490),
//                     CheckCoordinateValidity (:582-610), and for every normal pair the decisions of
//                     Process{Head,Normal,Tail}SequencePair that need no alignment (src/tools.cpp:225-253, 292-312, 344-363):
//                     pure insertion / deletion, the <= 2-mismatch shortcut against the 2-bit text, the long-end soft clip, the
//                     1 x 1 gap.  A pair that needs nw_alignment becomes a job descriptor for the NW kernels (nw_kernels.hip read
//                     the read characters and the 2-bit text in place); the candidate is then parked in a spill slot.  A
//                     candidate without jobs is finished right here.
//   aln_finish_kernel pass 2 for the parked candidates: CheckLocalAlignmentQuality, the leading / trailing gap trimming of the head
//                     and tail pairs, AddNewCigarElements (src/tools.cpp:49-104, 255-290, 314-339, 366-394) over the op strings.
//   aln_final_kernel  best / second best (src/AlignmentCandidates.cpp:724-740), CheckPairedFinalAlignments,
//                     Set{Paired,Single}AlignmentFlag, EvaluateMAPQ (src/Mapping.cpp:49-175, 429-480) and what
//                     Output{Paired,Singled}Alignments print (:177-315) as one kg_aln_record per read, plus the chunk's
//                     contribution to iPaired / iDistance and its EstDistance validity interval.
// Integer work throughout; MAPQ's one libm expression comes from a table the host fills with its own log().  No MFMA: there is
// no contraction here.  Anything outside the envelope (mate rescue, 8-mer partition of long fragments, > 12 seeds, > 47 CIGAR
// characters) marks the read pair for the host path instead.
#include "align_kernels.hpp"

namespace kg {

namespace {

// ---- small helpers --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int text_code(const AlnArgs &a, int64_t g)       // base of the indexed text (forward + reverse complement)
{
	return (a.ix.text[(uint64_t)g >> 2] >> (((uint32_t)g & 3) << 1)) & 3;
}
__device__ __forceinline__ char text_char(const AlnArgs &a, int64_t g)      // RefSequence[g]: always upper-case ACGT
{
	int c = text_code(a, g);
	return c == 0 ? 'A' : c == 1 ? 'C' : c == 2 ? 'G' : 'T';
}

struct __attribute__((packed, aligned(1))) AlnU64u { uint64_t v; };

// 32 bases of the 2-bit text from position p (base i in bits 2 i); beyond the end of the text: zeros
__device__ __forceinline__ uint64_t text_word32(const AlnArgs &a, int64_t p)
{
	if (p > a.two_genome_size) return 0;                 // (the text buffer has 16 bytes of slack behind its last base)
	const uint8_t *tp = a.ix.text + ((uint64_t)p >> 2);
	uint64_t lo = reinterpret_cast<const AlnU64u *>(tp)->v, hi = tp[8];
	int sh = ((int)p & 3) << 1;
	return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

__device__ __forceinline__ int chunk_of(const AlnArgs &a, int64_t r)
{
	// (the chunks of a batch are equally long but for the last: the proportional guess is right, two independent loads confirm it;
	//  the search below -- ~8 dependent loads for the 250 chunks of a 1 M-read batch -- is only the fallback)
	if (a.n_chunks > 1 && a.n_reads > 0) {
		int c = (int)((r * (int64_t)a.n_chunks) / a.n_reads);
		c = c < 0 ? 0 : c > a.n_chunks - 1 ? a.n_chunks - 1 : c;
		if (a.chunk_off[c] <= r && r < a.chunk_off[c + 1]) return c;
	}
	int lo = 0, hi = a.n_chunks - 1;
	while (lo < hi) {
		int mid = (lo + hi + 1) >> 1;
		if (a.chunk_off[mid] <= r) lo = mid; else hi = mid - 1;
	}
	return lo;
}

// ChrLocMap.lower_bound(g): index of the first key >= g, n_ends when there is none
__device__ __forceinline__ int end_lower_bound(const AlnArgs &a, int64_t g)
{
	int lo = 0, hi = a.n_ends;
	while (lo < hi) {
		int mid = (lo + hi) >> 1;
		if (a.contig_end[mid] < g) lo = mid + 1; else hi = mid;
	}
	return lo;
}

// A lane's share of a list whose end is a device counter: the wave's requests are summed and ONE returning atomic takes them all.
// (Every lane asking for itself was the stage's hidden cost: the counters are single addresses -- ctl[0..6] --, fifty million returning
// atomics per 100 M-read step queue at one L2 channel at about one per clock, ~25 ms per kernel whatever else the kernel does.)
// EVERY lane of the wave calls it, at a point where the wave has reconverged (need = 0: nothing for this lane).
__device__ __forceinline__ unsigned long long wave_reserve(unsigned long long *counter, unsigned long long need)
{
	const int lane = threadIdx.x & 63;
	unsigned long long incl = need;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const unsigned long long t = ((unsigned long long)(uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), off) << 32) | (uint32_t)__shfl_up((int)(uint32_t)incl, off);
		if (lane >= off) incl += t;
	}
	const unsigned long long total = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(incl >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)incl, 63);
	if (total == 0) return 0;
	unsigned long long base = 0;
	if (lane == 63) base = atomicAdd(counter, total);
	base = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 63);
	return base + incl - need;
}

// why a pair went back to the host (kg_align_reasons)
enum { WHY_PAIR_PRODUCT = 0, WHY_RESCUE_DIR1 = 1, WHY_RESCUE_WINDOW = 2, WHY_RESCUE_READ = 3, WHY_RESCUE_RUNS = 4, WHY_RESCUE_SEEDS = 5, WHY_SEEDS = 6,
       WHY_GAPS = 7, WHY_PARTITION = 8, WHY_CAPACITY = 9, WHY_CIGAR = 10, WHY_SCORE = 11, WHY_READ_LEN = 12 };

__device__ __forceinline__ void flag_host(const AlnArgs &a, int64_t r, int why)       // the pair of read r goes back to the host
{
	atomicAdd(&a.ctl[8 + why], 1ull);
	int c = chunk_of(a, r);
	int64_t base = a.chunk_off[c];
	if (a.chunk_paired[c]) {
		int64_t first = base + (((r - base) >> 1) << 1);
		a.r_host[first] = 1;
		a.r_host[first + 1] = 1;
	} else a.r_host[r] = 1;
}

// vector<AlignmentCandidate_t> of one read: the chained candidates (dense, [c0, c0 + nd)) followed by the slots of the pair's
// rescue windows ([r0, r0 + nr), mate 1 of a rescued pair only; a window that found nothing leaves a slot of score 0, which
// every consumer skips exactly like a candidate of score 0)
struct CandList {
	int64_t c0, r0;
	int nd, nr;
	__device__ __forceinline__ int n() const { return nd + nr; }
	__device__ __forceinline__ int64_t at(int i) const { return i < nd ? c0 + i : r0 + (i - nd); }
};
__device__ __forceinline__ CandList cand_list(const AlnArgs &a, int64_t r)
{
	CandList l;
	l.c0 = a.cand_off[r];
	l.nd = (int)(a.cand_off[r + 1] - l.c0);
	l.nr = a.resc_n[r];
	l.r0 = a.n_cands + (l.nr ? a.resc_off[r] : 0);
	return l;
}

// RemoveRedundantCandidates, src/Mapping.cpp:317-346 (non-PacBio)
__device__ void remove_redundant(const AlnArgs &a, const CandList &l)
{
	const int n = l.n();
	if (n <= 1) return;
	int s1 = 0, s2 = 0;
	for (int i = 0; i < n; ++i) {
		int s = a.c_score[l.at(i)];
		if (s > s2) {
			if (s >= s1) { s2 = s1; s1 = s; }
			else s2 = s;
		}
	}
	int thr = (s1 == s2 || s1 - s2 > 20) ? s1 : s2;
	for (int i = 0; i < n; ++i)
		if (a.c_score[l.at(i)] < thr) a.c_score[l.at(i)] = 0;
}

// RemoveUnMatedAlignmentCandidates, src/Mapping.cpp:402-427
__device__ void remove_unmated(const AlnArgs &a, const CandList &l1, const CandList &l2)
{
	for (int i = 0; i < l1.n(); ++i) {
		int j = a.c_mate[l1.at(i)];
		if (j == -1) a.c_score[l1.at(i)] = 0;
		else { int s = a.c_score[l1.at(i)] + a.c_score[l2.at(j)]; a.c_score[l1.at(i)] = s; a.c_score[l2.at(j)] = s; }
	}
	for (int j = 0; j < l2.n(); ++j)
		if (a.c_mate[l2.at(j)] == -1) a.c_score[l2.at(j)] = 0;
}

}  // namespace

// ---- pairing ---------------------------------------------------------------------------------------------------------------
// one pair (or one single-end read): CheckPairedAlignmentCandidates and what follows it.  out_ck / out_lo / out_hi: the pair's
// contribution to its chunk's EstDistance validity interval (out_ck < 0: none) -- merged per wave by the kernel.
// In two halves: pair_front runs up to the point where the pair knows how many rescue windows it wants (st.nt; 0: it is done), the kernel
// reserves the task slots of the whole wave with one atomic (wave_reserve), pair_back writes the windows and finishes the pair.
struct PairState {
	int64_t a1;
	int ck, n1, n2, sc1, rl1, rl2, est_r, thr, nt;
};

// the windows of mate 1 next to the candidates of mate 2 (src/AlignmentRescue.cpp:127-165): counted (tasks == nullptr) or written from slot `base` on.
// Returns their number; host: a window beyond what the rescue kernel takes
__device__ __forceinline__ int rescue_windows(const AlnArgs &a, int64_t r, const PairState &st, bool write, unsigned long long base, bool &host, int &why)
{
	int k = 0;
	for (int j = 0; j < st.n2; ++j) {
		if (a.c_score[st.a1 + j] < st.thr) continue;
		int64_t pd = a.cands[st.a1 + j].posDiff;
		int64_t left = pd - st.est_r, right = pd + st.rl2;
		int it = end_lower_bound(a, right);
		if (it == a.n_ends) continue;
		int chr = a.end_chr[it];
		int64_t fs = a.chr_fwd_start[chr], rs = a.chr_rev_start[chr], cl = a.chr_len[chr];
		if (left < a.genome_size && left < (fs - cl)) left = fs - cl + 1;
		else if (right >= a.genome_size && left < (rs - cl)) left = rs - cl + 1;
		int slen = (int)(right - left);
		if (slen < st.rl1) continue;
		if (left < 0) { left = 0; slen = (int)(right - left); if (slen < st.rl1) continue; }
		if (right > a.two_genome_size) continue;
		if (slen > kRescueMaxWindow) { host = true; why = WHY_RESCUE_WINDOW; break; }
		if (write) {
			RescueTask t;
			t.left = left; t.read = (int32_t)r; t.j = j; t.slen = slen; t.score1 = st.sc1; t.ordinal = k;
			a.tasks[base + k] = t;
			int64_t slot = a.n_cands + (int64_t)(base + k);
			a.c_score[slot] = 0; a.c_mate[slot] = -1; a.c_read[slot] = (int32_t)r;
		}
		k++;
	}
	return k;
}

__device__ __forceinline__ void pair_front(const AlnArgs &a, const int64_t r, int &out_ck, long long &out_lo, long long &out_hi, PairState &st)
{
		st.nt = 0;
		const int ck = chunk_of(a, r);
		const bool paired = a.chunk_paired[ck] != 0;
		const int64_t in_chunk = r - a.chunk_off[ck];
		if (paired && (in_chunk & 1)) return;                        // the first mate's lane does the pair
		const CandList l1 = cand_list(a, r);                         // (no rescue slots yet: resc_n is zero)
		const int64_t a0 = l1.c0;
		const int n1 = l1.nd;
		for (int i = 0; i < n1; ++i) { a.c_score[a0 + i] = a.cands[a0 + i].score; a.c_mate[a0 + i] = -1; a.c_read[a0 + i] = (int32_t)r; }
		a.records[r].est_lo = -1; a.records[r].est_hi = 0x7fffffff; a.records[r].rescue = 0;
		if (!paired) {
			remove_redundant(a, l1);                                 // src/Mapping.cpp:589
			return;
		}
		a.records[r + 1].est_lo = -1; a.records[r + 1].est_hi = 0x7fffffff; a.records[r + 1].rescue = 0;
		const CandList l2 = cand_list(a, r + 1);
		const int64_t a1 = l2.c0;
		const int n2 = l2.nd;
		for (int j = 0; j < n2; ++j) { a.c_score[a1 + j] = a.cands[a1 + j].score; a.c_mate[a1 + j] = -1; a.c_read[a1 + j] = (int32_t)(r + 1); }
		if ((int64_t)n1 * n2 > kAlnPairProduct) { flag_host(a, r, WHY_PAIR_PRODUCT); return; }
		// CheckPairedAlignmentCandidates, src/Mapping.cpp:348-400
		if (n1 * n2 > 1000) { remove_redundant(a, l1); remove_redundant(a, l2); }
		bool pairing = false;
		long long lo = -1, hi = 0x7fffffffffffffffll;
		const long long est = a.est_distance;
		for (int i = 0; i < n1; ++i) {
			if (a.c_score[a0 + i] == 0) continue;
			const int64_t pd1 = a.cands[a0 + i].posDiff;
			int best = -1, s = 0;
			for (int j = 0; j < n2; ++j) {
				int sj = a.c_score[a1 + j];
				int64_t pd2 = a.cands[a1 + j].posDiff;
				if (sj == 0 || pd2 < pd1) continue;
				long long dist = pd2 - pd1;
				if (dist < est) {
					if (dist > lo) lo = dist;
					if (sj > s) { best = j; s = sj; }
					else if (sj == s) best = -1;
				} else if (dist < hi) hi = dist;
			}
			if (s > 0 && best != -1) {
				int j = best;
				int mj = a.c_mate[a1 + j];
				if (mj == -1) {
					pairing = true;
					a.c_mate[a0 + i] = j;
					a.c_mate[a1 + j] = i;
				} else if (a.c_score[a0 + i] > a.c_score[a0 + mj]) {
					a.c_mate[a0 + mj] = -1;
					a.c_mate[a0 + i] = j;
					a.c_mate[a1 + j] = i;
				}
			}
		}
		out_ck = ck; out_lo = lo; out_hi = hi;                       // (into the chunk's interval by the caller: one atomic pair per wave)
		{
			// the pair's own interval (every distance that matters is far below 2^31: EstDistance never exceeds 1.5 x 10000)
			int32_t plo = (int32_t)lo, phi = hi > 0x7fffffffll ? 0x7fffffff : (int32_t)hi;
			a.records[r].est_lo = plo; a.records[r].est_hi = phi;
			a.records[r + 1].est_lo = plo; a.records[r + 1].est_hi = phi;
		}
		if (pairing) remove_unmated(a, l1, l2);
		else {
			// RescueUnpairedAlignment is due (src/Mapping.cpp:559-560; src/AlignmentRescue.cpp:73-170)
			a.chunk_stats[ck].rescue_wanted = 1;
			a.records[r].rescue = 1; a.records[r + 1].rescue = 1;
			int sc1 = 0, sc2 = 0;
			for (int i = 0; i < n1; ++i) sc1 = max(sc1, a.c_score[a0 + i]);
			for (int j = 0; j < n2; ++j) sc2 = max(sc2, a.c_score[a1 + j]);
			const int rl1 = (int)(a.read_off[r + 1] - a.read_off[r]), rl2 = (int)(a.read_off[r + 2] - a.read_off[r + 1]);
			int strategy;
			if (sc1 == 0 && sc2 == 0) strategy = 0;                                            // :83 returns at once
			else if (sc1 < (int)(rl1 * 0.1) && sc2 < (int)(rl2 * 0.1)) strategy = 4;          // :84: neither direction is tried
			else if (sc1 > sc2 && sc1 - sc2 > 50) strategy = 1;
			else if (sc2 > sc1 && sc2 - sc1 > 50) strategy = 2;
			else strategy = 3;
			const int est_r = a.est_distance > a.max_insert ? a.max_insert : a.est_distance;   // :95
			bool host = false;
			int why = 0;
			if (strategy == 1 || strategy == 3) {
				// mate 2 next to the candidates of mate 1 (:97-125).  The right end of that window is clamped against the START
				// of the contig (:111-112), which collapses it: the "slen < rlen" test skips it.  Verified per window here; a
				// window that would be scanned after all goes to the host.
				int thr = sc1 - 30;
				if (thr < 50) thr = 50;
				for (int i = 0; i < n1 && !host; ++i) {
					if (a.c_score[a0 + i] < thr) continue;
					int64_t left = a.cands[a0 + i].posDiff, right = left + est_r + rl2;
					int it = end_lower_bound(a, left);
					if (it == a.n_ends) continue;
					int chr = a.end_chr[it];
					if (right < a.genome_size && right > a.chr_fwd_start[chr]) right = a.chr_fwd_start[chr] - 1;
					else if (right >= a.genome_size && right > a.chr_rev_start[chr]) right = a.chr_rev_start[chr] - 1;
					int slen = (int)(right - left);
					if (slen < rl2) continue;
					if (left < 0 || right > a.two_genome_size) continue;
					host = true; why = WHY_RESCUE_DIR1;
				}
			}
			int nt = 0;
			if (!host && (strategy == 2 || strategy == 3)) {
				// mate 1 next to the candidates of mate 2 (:127-165): one task per window -- counted here, written by pair_back
				st.a1 = a1; st.ck = ck; st.n1 = n1; st.n2 = n2; st.sc1 = sc1; st.rl1 = rl1; st.rl2 = rl2; st.est_r = est_r;
				st.thr = sc2 - 30;                        // (nothing was appended to mate 2's list above)
				if (st.thr < 50) st.thr = 50;
				nt = rescue_windows(a, r, st, false, 0, host, why);
				if (!host && nt > 200) { host = true; why = WHY_CAPACITY; }
			}
			if (host) { a.resc_n[r] = 0; flag_host(a, r, why); return; }
			if (nt > 0) { st.nt = nt; return; }                    // the windows are written once the wave has its task slots (pair_back)
		}
		remove_redundant(a, l1);                                     // src/Mapping.cpp:563
		remove_redundant(a, l2);
}

// a pair with st.nt rescue windows, task slots [base, base + nt) reserved
__device__ __forceinline__ void pair_back(const AlnArgs &a, const int64_t r, const PairState &st, unsigned long long base)
{
	bool host = false;
	int why = 0;
	if (base + (unsigned long long)st.nt > (unsigned long long)a.task_capacity) { host = true; why = WHY_CAPACITY; }
	else {
		(void)rescue_windows(a, r, st, true, base, host, why);
		a.resc_off[r] = (int32_t)base; a.resc_n[r] = (uint8_t)st.nt;
		// the 8-mer code skips 'N' and maps everything else through nst_nt4_table (src/KmerAnalysis.cpp:25-32, 56-102);
		// the kernel compares 2-bit codes, which is the same thing for reads made of A/C/G/T in either case
		if (st.rl1 > kRescueMaxRead || st.rl1 < 8) { host = true; why = WHY_RESCUE_READ; }
		const uint8_t *rd = a.enc + a.read_off[r];
		for (int i0 = 0; i0 < st.rl1 && !host; i0 += 8) {          // (eight characters per load: the character array has 64 bytes of slack)
			const uint64_t w = reinterpret_cast<const AlnU64u *>(rd + i0)->v;
			const int m = st.rl1 - i0 < 8 ? st.rl1 - i0 : 8;
			for (int i = 0; i < m; ++i) {
				unsigned u = (unsigned)((w >> (8 * i)) & 0xDFu);
				if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T')) { host = true; why = WHY_RESCUE_READ; }
			}
		}
	}
	if (host) { a.resc_n[r] = 0; flag_host(a, r, why); return; }
	a.r_pending[r] = 1;                                              // filters follow once the windows are scanned (aln_post_rescue_kernel)
}

// ---- the same for a pair with MANY candidates, the whole wave on it ----------------------------------------------------------------
// A pair out of a repeat family comes with tens of candidates per mate: CheckPairedAlignmentCandidates is a loop over n1 x n2 of them, the
// filters and the rescue windows loops over each list, every step a dependent trip to the per-candidate arrays -- and a wave costs what its
// heaviest lane costs (64 consecutive pairs of the hg38-sized workload: the heaviest lane carries tens of times the wave's mean, tools/cand_histogram.py).
// Pairs above kPairHeavy candidate pairs are therefore taken out of the lanes' loop and done by the wave together, one after the other:
// lane j holds candidate j of a list (j + 64, ... where a list is longer), the inner loop of :362-391 is one step per candidate of mate 1 --
// the best score among the admissible candidates of mate 2 and whether a single one reaches it, by wave reductions; the order in which the
// reference walks mate 2's list does not matter for that --, the mate book-keeping stays sequential in the candidates of mate 1 as in
// the reference (a later candidate may take an earlier one's mate, :381-388).  Same arrays, same results as pair_front / pair_back.
constexpr int kPairHeavy = 32;

__device__ __forceinline__ void wave_sync_mem()          // the wave's stores to the per-candidate arrays are visible to all its lanes
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ int wave_max(int v) { for (int off = 32; off > 0; off >>= 1) { const int t = __shfl_xor(v, off); v = t > v ? t : v; } return v; }
__device__ __forceinline__ int wave_sum(int v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; }

// RemoveRedundantCandidates (src/Mapping.cpp:317-346), the wave on one list
__device__ void remove_redundant_wave(const AlnArgs &a, const CandList &l)
{
	const int n = l.n(), lane = threadIdx.x & 63;
	if (n <= 1) return;
	int s1 = 0, s2 = 0;
	for (int i = lane; i < n; i += 64) {
		const int s = a.c_score[l.at(i)];
		if (s > s2) {
			if (s >= s1) { s2 = s1; s1 = s; }
			else s2 = s;
		}
	}
	for (int off = 32; off > 0; off >>= 1) {                // the two largest of the union (a value twice: both)
		const int b1 = __shfl_xor(s1, off), b2 = __shfl_xor(s2, off);
		const int hi = s1 > b1 ? s1 : b1, lo = s1 > b1 ? b1 : s1, rest = s2 > b2 ? s2 : b2;
		s1 = hi; s2 = lo > rest ? lo : rest;
	}
	const int thr = (s1 == s2 || s1 - s2 > 20) ? s1 : s2;
	for (int i = lane; i < n; i += 64)
		if (a.c_score[l.at(i)] < thr) a.c_score[l.at(i)] = 0;
	wave_sync_mem();
}

// rescue_windows, lanes over the candidates of mate 2 (ordinals in list order, as the loop of :127-165 hands them out)
__device__ int rescue_windows_wave(const AlnArgs &a, int64_t r, const PairState &st, bool write, unsigned long long base, bool &host, int &why)
{
	const int lane = threadIdx.x & 63;
	int k_total = 0;
	for (int j0 = 0; j0 < st.n2; j0 += 64) {
		const int j = j0 + lane;
		bool valid = false, too_big = false;
		int64_t left = 0;
		int slen = 0;
		if (j < st.n2 && a.c_score[st.a1 + j] >= st.thr) {
			const int64_t pd = a.cands[st.a1 + j].posDiff;
			left = pd - st.est_r;
			const int64_t right = pd + st.rl2;
			const int it = end_lower_bound(a, right);
			if (it != a.n_ends) {
				const int chr = a.end_chr[it];
				const int64_t fs = a.chr_fwd_start[chr], rs = a.chr_rev_start[chr], cl = a.chr_len[chr];
				if (left < a.genome_size && left < (fs - cl)) left = fs - cl + 1;
				else if (right >= a.genome_size && left < (rs - cl)) left = rs - cl + 1;
				slen = (int)(right - left);
				valid = slen >= st.rl1;
				if (valid && left < 0) { left = 0; slen = (int)(right - left); valid = slen >= st.rl1; }
				if (valid && right > a.two_genome_size) valid = false;
				if (valid && slen > kRescueMaxWindow) { too_big = true; valid = false; }
			}
		}
		// (the reference's loop stops at the first window beyond the kernel's reach; whatever it had counted before, the pair is the host's)
		if (__ballot(too_big)) { host = true; why = WHY_RESCUE_WINDOW; return k_total; }
		const uint64_t mask = __ballot(valid);
		if (write && valid) {
			const int k = k_total + __popcll(mask & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
			RescueTask t;
			t.left = left; t.read = (int32_t)r; t.j = j; t.slen = slen; t.score1 = st.sc1; t.ordinal = k;
			a.tasks[base + k] = t;
			const int64_t slot = a.n_cands + (int64_t)(base + k);
			a.c_score[slot] = 0; a.c_mate[slot] = -1; a.c_read[slot] = (int32_t)r;
		}
		k_total += __popcll(mask);
	}
	return k_total;
}

// pair_front for the pair of reads (r, r + 1) of an all-paired batch, every lane of the wave in it; the results are the same in all lanes
__device__ void pair_front_wave(const AlnArgs &a, const int64_t r, int &out_ck, long long &out_lo, long long &out_hi, PairState &st)
{
	const int lane = threadIdx.x & 63;
	st.nt = 0;
	const int ck = chunk_of(a, r);
	const CandList l1 = cand_list(a, r), l2 = cand_list(a, r + 1);          // (no rescue slots yet: resc_n is zero)
	const int64_t a0 = l1.c0, a1 = l2.c0;
	const int n1 = l1.nd, n2 = l2.nd;
	for (int i = lane; i < n1; i += 64) { a.c_score[a0 + i] = a.cands[a0 + i].score; a.c_mate[a0 + i] = -1; a.c_read[a0 + i] = (int32_t)r; }
	for (int j = lane; j < n2; j += 64) { a.c_score[a1 + j] = a.cands[a1 + j].score; a.c_mate[a1 + j] = -1; a.c_read[a1 + j] = (int32_t)(r + 1); }
	if (lane == 0) {
		a.records[r].est_lo = -1; a.records[r].est_hi = 0x7fffffff; a.records[r].rescue = 0;
		a.records[r + 1].est_lo = -1; a.records[r + 1].est_hi = 0x7fffffff; a.records[r + 1].rescue = 0;
	}
	wave_sync_mem();
	if ((int64_t)n1 * n2 > kAlnPairProduct) { if (lane == 0) flag_host(a, r, WHY_PAIR_PRODUCT); return; }
	// CheckPairedAlignmentCandidates, src/Mapping.cpp:348-400
	if (n1 * n2 > 1000) { remove_redundant_wave(a, l1); remove_redundant_wave(a, l2); }
	bool pairing = false;
	long long lo = -1, hi = 0x7fffffffffffffffll;            // (lane-local until the loop is through)
	const long long est = a.est_distance;
	for (int i = 0; i < n1; ++i) {
		const int si = a.c_score[a0 + i];
		if (si == 0) continue;
		const int64_t pd1 = a.cands[a0 + i].posDiff;
		int m = 0, cnt = 0, arg = -1;                        // of this lane's candidates of mate 2: the best admissible score, how many reach it, the first that does
		for (int j = lane; j < n2; j += 64) {
			const int sj = a.c_score[a1 + j];
			const int64_t pd2 = a.cands[a1 + j].posDiff;
			if (sj == 0 || pd2 < pd1) continue;
			const long long dist = pd2 - pd1;
			if (dist < est) {
				if (dist > lo) lo = dist;
				if (sj > m) { m = sj; cnt = 1; arg = j; }
				else if (sj == m) cnt++;
			} else if (dist < hi) hi = dist;
		}
		const int s = wave_max(m);
		if (s <= 0) continue;
		const int mine = m == s ? cnt : 0;
		if (wave_sum(mine) != 1) continue;                   // two candidates of the best score: no mate for this one (:374-375)
		const int best = __shfl(arg, __ffsll((unsigned long long)__ballot(mine == 1)) - 1);
		const int mj = a.c_mate[a1 + best];
		if (mj == -1) {
			pairing = true;
			if (lane == 0) { a.c_mate[a0 + i] = best; a.c_mate[a1 + best] = i; }
		} else if (si > a.c_score[a0 + mj]) {
			if (lane == 0) { a.c_mate[a0 + mj] = -1; a.c_mate[a0 + i] = best; a.c_mate[a1 + best] = i; }
		}
		wave_sync_mem();
	}
	for (int off = 32; off > 0; off >>= 1) {
		const long long l2_ = __shfl_xor(lo, off), h2_ = __shfl_xor(hi, off);
		lo = l2_ > lo ? l2_ : lo;
		hi = h2_ < hi ? h2_ : hi;
	}
	out_ck = ck; out_lo = lo; out_hi = hi;
	if (lane == 0) {
		int32_t plo = (int32_t)lo, phi = hi > 0x7fffffffll ? 0x7fffffff : (int32_t)hi;
		a.records[r].est_lo = plo; a.records[r].est_hi = phi;
		a.records[r + 1].est_lo = plo; a.records[r + 1].est_hi = phi;
	}
	if (pairing) {
		// RemoveUnMatedAlignmentCandidates, src/Mapping.cpp:402-427 (the mates are a matching: no two candidates of mate 1 share one of mate 2)
		for (int i = lane; i < n1; i += 64) {
			const int j = a.c_mate[a0 + i];
			if (j == -1) a.c_score[a0 + i] = 0;
			else { const int sum = a.c_score[a0 + i] + a.c_score[a1 + j]; a.c_score[a0 + i] = sum; a.c_score[a1 + j] = sum; }
		}
		wave_sync_mem();
		for (int j = lane; j < n2; j += 64)
			if (a.c_mate[a1 + j] == -1) a.c_score[a1 + j] = 0;
		wave_sync_mem();
	} else {
		// RescueUnpairedAlignment is due (src/Mapping.cpp:559-560; src/AlignmentRescue.cpp:73-170)
		if (lane == 0) { a.chunk_stats[ck].rescue_wanted = 1; a.records[r].rescue = 1; a.records[r + 1].rescue = 1; }
		int sc1 = 0, sc2 = 0;
		for (int i = lane; i < n1; i += 64) sc1 = max(sc1, a.c_score[a0 + i]);
		for (int j = lane; j < n2; j += 64) sc2 = max(sc2, a.c_score[a1 + j]);
		sc1 = wave_max(sc1); sc2 = wave_max(sc2);
		const int rl1 = (int)(a.read_off[r + 1] - a.read_off[r]), rl2 = (int)(a.read_off[r + 2] - a.read_off[r + 1]);
		int strategy;
		if (sc1 == 0 && sc2 == 0) strategy = 0;
		else if (sc1 < (int)(rl1 * 0.1) && sc2 < (int)(rl2 * 0.1)) strategy = 4;
		else if (sc1 > sc2 && sc1 - sc2 > 50) strategy = 1;
		else if (sc2 > sc1 && sc2 - sc1 > 50) strategy = 2;
		else strategy = 3;
		const int est_r = a.est_distance > a.max_insert ? a.max_insert : a.est_distance;
		bool host = false;
		int why = 0;
		if (strategy == 1 || strategy == 3) {
			// mate 2 next to the candidates of mate 1 (:97-125): a window that would be scanned after all goes to the host (pair_front)
			int thr = sc1 - 30;
			if (thr < 50) thr = 50;
			bool found = false;
			for (int i = lane; i < n1; i += 64) {
				if (a.c_score[a0 + i] < thr) continue;
				int64_t left = a.cands[a0 + i].posDiff, right = left + est_r + rl2;
				const int it = end_lower_bound(a, left);
				if (it == a.n_ends) continue;
				const int chr = a.end_chr[it];
				if (right < a.genome_size && right > a.chr_fwd_start[chr]) right = a.chr_fwd_start[chr] - 1;
				else if (right >= a.genome_size && right > a.chr_rev_start[chr]) right = a.chr_rev_start[chr] - 1;
				const int slen = (int)(right - left);
				if (slen < rl2) continue;
				if (left < 0 || right > a.two_genome_size) continue;
				found = true;
			}
			if (__ballot(found)) { host = true; why = WHY_RESCUE_DIR1; }
		}
		int nt = 0;
		if (!host && (strategy == 2 || strategy == 3)) {
			st.a1 = a1; st.ck = ck; st.n1 = n1; st.n2 = n2; st.sc1 = sc1; st.rl1 = rl1; st.rl2 = rl2; st.est_r = est_r;
			st.thr = sc2 - 30;
			if (st.thr < 50) st.thr = 50;
			nt = rescue_windows_wave(a, r, st, false, 0, host, why);
			if (!host && nt > 200) { host = true; why = WHY_CAPACITY; }
		}
		if (host) { if (lane == 0) { a.resc_n[r] = 0; flag_host(a, r, why); } return; }
		if (nt > 0) { st.nt = nt; return; }
	}
	remove_redundant_wave(a, l1);
	remove_redundant_wave(a, l2);
}

__device__ void pair_back_wave(const AlnArgs &a, const int64_t r, const PairState &st, unsigned long long base)
{
	const int lane = threadIdx.x & 63;
	bool host = false;
	int why = 0;
	if (base + (unsigned long long)st.nt > (unsigned long long)a.task_capacity) { host = true; why = WHY_CAPACITY; }
	else {
		(void)rescue_windows_wave(a, r, st, true, base, host, why);
		if (lane == 0) { a.resc_off[r] = (int32_t)base; a.resc_n[r] = (uint8_t)st.nt; }
		if (st.rl1 > kRescueMaxRead || st.rl1 < 8) { host = true; why = WHY_RESCUE_READ; }
		const uint8_t *rd = a.enc + a.read_off[r];
		bool bad = false;
		for (int i = lane; i < st.rl1 && !host; i += 64) {
			const unsigned u = rd[i] & 0xDFu;
			bad = bad || !(u == 'A' || u == 'C' || u == 'G' || u == 'T');
		}
		if (__ballot(bad)) { host = true; why = WHY_RESCUE_READ; }
	}
	if (lane != 0) return;
	if (host) { a.resc_n[r] = 0; flag_host(a, r, why); return; }
	a.r_pending[r] = 1;
}

// One PAIR per lane (one read per lane where a chunk is not paired).  Rounds 2-3 ran one READ per lane and let the second mate's lane
// leave at once -- half of every wave idle -- and sent two same-address atomics per pair at the chunk's interval (2000 pairs per
// chunk: a wave's 64 lanes hit one address); now a wave whose pairs lie in one chunk sends one pair of atomics, the rescue
// windows of the wave's pairs take their task slots with one atomic, and the pairs with many candidates are the whole wave's.
__global__ __launch_bounds__(256) void aln_pair_kernel(AlnArgs a)
{
	// (a.slow_pairs: the pairs aln_trivial_kernel did not decide, ctl[35] of them; else every pair / read of the batch)
	const int64_t n_units = a.slow_pairs ? (int64_t)a.ctl[35] : a.all_paired ? a.n_reads >> 1 : a.n_reads;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	const int lane = threadIdx.x & 63;
	const bool heavy_on = a.all_paired && !a.dbg_no_heavy;
	for (int64_t u0 = (int64_t)blockIdx.x * blockDim.x; u0 < n_units; u0 += stride) {
		const int64_t u = u0 + threadIdx.x;
		int ck = -1;
		long long lo = -1, hi = 0x7fffffffffffffffll;
		PairState st;
		st.nt = 0;
		const int64_t r = u < n_units ? (a.slow_pairs ? (int64_t)a.slow_pairs[u] << 1 : a.all_paired ? u << 1 : u) : 0;
		bool heavy = false;
		if (u < n_units && heavy_on) {
			const int64_t c0 = a.cand_off[r], c1 = a.cand_off[r + 1], c2 = a.cand_off[r + 2];
			heavy = (c1 - c0) * (c2 - c1) > (int64_t)a.pair_heavy;
		}
		if (u < n_units && !heavy) pair_front(a, r, ck, lo, hi, st);
		uint64_t hm = __ballot(heavy);
		while (hm) {
			const int src = __ffsll((unsigned long long)hm) - 1;
			hm &= hm - 1;
			const int64_t rh = ((int64_t)__shfl((int)(uint32_t)((uint64_t)r >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)(uint64_t)r, src);
			int ck_h = -1;
			long long lo_h = -1, hi_h = 0x7fffffffffffffffll;
			PairState st_h;
			pair_front_wave(a, rh, ck_h, lo_h, hi_h, st_h);
			if (lane == src) { ck = ck_h; lo = lo_h; hi = hi_h; st = st_h; }
		}
		const unsigned long long base = wave_reserve(&a.ctl[4], (unsigned long long)st.nt);
		if (st.nt > 0 && !heavy) pair_back(a, r, st, base);
		hm = __ballot(heavy && st.nt > 0);
		while (hm) {
			const int src = __ffsll((unsigned long long)hm) - 1;
			hm &= hm - 1;
			const int64_t rh = ((int64_t)__shfl((int)(uint32_t)((uint64_t)r >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)(uint64_t)r, src);
			PairState st_h;
			st_h.a1 = ((int64_t)__shfl((int)(uint32_t)((uint64_t)st.a1 >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)(uint64_t)st.a1, src);
			st_h.ck = __shfl(st.ck, src); st_h.n1 = __shfl(st.n1, src); st_h.n2 = __shfl(st.n2, src); st_h.sc1 = __shfl(st.sc1, src);
			st_h.rl1 = __shfl(st.rl1, src); st_h.rl2 = __shfl(st.rl2, src); st_h.est_r = __shfl(st.est_r, src); st_h.thr = __shfl(st.thr, src); st_h.nt = __shfl(st.nt, src);
			const unsigned long long bh = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(base >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)base, src);
			pair_back_wave(a, rh, st_h, bh);
		}
		// ---- the chunk's interval: lo = max over the pairs, hi = min ----
		const uint64_t have = __ballot(ck >= 0);
		if (have == 0) continue;
		const int ck0 = __shfl(ck, __ffsll((unsigned long long)have) - 1);
		if (__ballot(ck >= 0 && ck != ck0) == 0) {
			for (int off = 32; off > 0; off >>= 1) {
				const long long l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
				lo = l2 > lo ? l2 : lo;
				hi = h2 < hi ? h2 : hi;
			}
			if ((threadIdx.x & 63) == 0) {
				if (lo > -1) atomicMax((long long *)&a.chunk_stats[ck0].lo, lo);
				if (hi != 0x7fffffffffffffffll) atomicMin((long long *)&a.chunk_stats[ck0].hi, hi);
			}
		} else if (ck >= 0) {
			if (lo > -1) atomicMax((long long *)&a.chunk_stats[ck].lo, lo);
			if (hi != 0x7fffffffffffffffll) atomicMin((long long *)&a.chunk_stats[ck].hi, hi);
		}
	}
}

// ---- mate rescue: one wave per window ----------------------------------------------------------------------------------------
// IdentifyCommonKmers + GenerateSimplePairsFromCommonKmers(10) over a window (src/KmerAnalysis.cpp:104-162) produce, per diagonal,
// the maximal runs of consecutive common 8-mers = the maximal exact matches of at least 10 bases between the read and the window
// along that diagonal, sorted by (diagonal, read position).  The kernel finds those runs directly: every lane takes a block of
// consecutive diagonals and XORs 2-bit packed read words against the window shifted to that diagonal.
// IdnetifyRescueCandidate (src/AlignmentRescue.cpp:24-69) then groups consecutive runs whose diagonals lie within MaxGaps of
// the group's first one and keeps the first group of the largest total length.
__global__ __launch_bounds__(64) void aln_rescue_kernel(AlnArgs a)
{
	__shared__ uint64_t rd2[kRescueMaxRead / 32 + 2];          // read, 2 bits per base, base t in bits 2*(t&31) of word t>>5
	__shared__ uint64_t win2[kRescueMaxWindow / 32 + 4];       // window likewise
	__shared__ int raw_key[kRescueMaxRuns], raw_len[kRescueMaxRuns];      // runs as the lanes find them: (diagonal index << 8 | read position), length
	__shared__ int run_d[kRescueMaxRuns], run_t[kRescueMaxRuns], run_l[kRescueMaxRuns];   // ... sorted by (diagonal, read position)
	__shared__ int n_raw;
	__shared__ int kh_head[512], kh_next[kRescueMaxRead];           // the read's 10-mers: hash slot -> chain of read positions
	__shared__ uint32_t kh_key[kRescueMaxRead];
	const int lane = threadIdx.x;
	unsigned long long n_tasks = a.ctl[4];
	if (n_tasks > (unsigned long long)a.task_capacity) n_tasks = (unsigned long long)a.task_capacity;
	for (unsigned long long ti = blockIdx.x; ti < n_tasks; ti += gridDim.x) {
		const RescueTask t = a.tasks[ti];
		const int64_t slot = a.n_cands + (int64_t)ti;
		if (a.r_host[t.read]) continue;                            // (uniform per block)
		const int rlen = (int)(a.read_off[t.read + 1] - a.read_off[t.read]);
		const uint8_t *rd = a.enc + a.read_off[t.read];
		const int slen = t.slen;
		const int rwords = (rlen + 31) >> 5, wwords = (slen + 31) >> 5;
		__syncthreads();
		// the read as 2-bit codes, 64 bases per step: every lane converts one character (A 00, C 01, G 11, T 10 in either case
		// is one Gray step from the codes 0..3), two ballots collect the bit planes, which are then interleaved
		for (int w2 = 0; (w2 << 6) < rlen + 32; ++w2) {
			int p = (w2 << 6) + lane;
			unsigned g = 0;
			if (p < rlen) { unsigned ch = rd[p]; g = (ch >> 1) & 3; g ^= g >> 1; }
			uint64_t m0 = __ballot(g & 1), m1 = __ballot(g & 2);
			if (lane < 2) {
				uint32_t lo0 = (uint32_t)(m0 >> (lane << 5)), lo1 = (uint32_t)(m1 >> (lane << 5));
				auto spread = [](uint32_t v) {
					uint64_t x = v;
					x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
					x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
					x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
					x = (x | (x << 2)) & 0x3333333333333333ull;
					x = (x | (x << 1)) & 0x5555555555555555ull;
					return x;
				};
				int w = (w2 << 1) + lane;
				if (w < rwords + 1) rd2[w] = spread(lo0) | (spread(lo1) << 1);
			}
		}
		// the window straight from the 2-bit text (same packing: base i in bits 2 (i & 3) of byte i >> 2): one unaligned 64-bit
		// load + the next byte per word
		for (int w = lane; w < wwords + 2; w += 64) {
			int64_t g0 = t.left + ((int64_t)w << 5);
			uint64_t x = 0;
			if ((w << 5) < slen) {
				const uint8_t *tp = a.ix.text + ((uint64_t)g0 >> 2);
				uint64_t lo_w = 0;
				for (int k = 0; k < 8; ++k) lo_w |= (uint64_t)tp[k] << (k << 3);
				uint64_t hi_b = tp[8];
				int sh = ((int)g0 & 3) << 1;
				x = sh ? (lo_w >> sh) | (hi_b << (64 - sh)) : lo_w;
				int valid = slen - (w << 5);
				if (valid < 32) x &= (1ull << (valid << 1)) - 1;
			}
			win2[w] = x;
		}
		if (lane == 0) n_raw = 0;
		__syncthreads();
		// diagonals d = gpos - rpos of k-mer pairs: -(rlen - 8) .. slen - 8
		const int d_lo = -(rlen - 8), nd = slen + rlen - 15;
		if (!a.dbg_rescue_scan) {
			// Round 4: the runs through the read's 10-mers.  Every maximal exact match of >= 10 bases starts with a common 10-mer whose
			// predecessor pair differs (or does not exist): the read's <= 247 10-mers go into a 512-slot LDS hash, every lane looks the
			// window's 10-mers up (26 positions per lane for a 1650-base window), and a hit that starts a run is extended 32 bases per
			// step.  The scan below walked all ~1800 diagonals of the window, 8 words each (87 G VALU wave-instructions per 80 M
			// reads, 54 % of the kernel's cycles waiting on its own issue, profiles/r03w); the same runs come out (rank-sorted afterwards).
			auto bits_at = [](const uint64_t *v, int pos) -> uint64_t {        // 32 bases from base `pos` (2 bits each)
				const int w = pos >> 5, sh = (pos & 31) << 1;
				return sh ? (v[w] >> sh) | (v[w + 1] << (64 - sh)) : v[w];
			};
			for (int i = lane; i < 512; i += 64) kh_head[i] = -1;
			__syncthreads();
			for (int q = lane; q + 10 <= rlen; q += 64) {
				const uint32_t key = (uint32_t)(bits_at(rd2, q) & 0xFFFFFull);
				kh_key[q] = key;
				kh_next[q] = atomicExch(&kh_head[(key * 0x9E3779B1u) >> 23], q);
			}
			__syncthreads();
			for (int w = lane; w + 10 <= slen; w += 64) {
				const uint32_t key = (uint32_t)(bits_at(win2, w) & 0xFFFFFull);
				for (int q = kh_head[(key * 0x9E3779B1u) >> 23]; q >= 0; q = kh_next[q]) {
					if (kh_key[q] != key) continue;
					if (q > 0 && w > 0 && ((((rd2[(q - 1) >> 5] >> (((q - 1) & 31) << 1)) ^ (win2[(w - 1) >> 5] >> (((w - 1) & 31) << 1))) & 3) == 0)) continue;   // not where the run starts
					const int room = rlen - q < slen - w ? rlen - q : slen - w;
					int e = 0;
					while (e < room) {
						const uint64_t diff = bits_at(rd2, q + e) ^ bits_at(win2, w + e);
						const uint64_t ne = (diff | (diff >> 1)) & 0x5555555555555555ull;
						if (ne) { e += (__ffsll((unsigned long long)ne) - 1) >> 1; break; }
						e += 32;
					}
					if (e > room) e = room;
					const int at_ = atomicAdd(&n_raw, 1);
					if (at_ < kRescueMaxRuns) { raw_key[at_] = ((w - q - d_lo) << 8) | q; raw_len[at_] = e; }
				}
			}
		} else {
		const int per = (nd + 63) >> 6;
		// per diagonal: the equality bit of every read position (one bit per base, up to 256), then the positions where ten
		// consecutive bits are set by shift-and doubling; almost every diagonal ends there with nothing set
		for (int q = 0; q < per; ++q) {
			int di = lane * per + q;
			if (di >= nd) break;
			int d = d_lo + di;
			int t_lo = d < 0 ? -d : 0;
			int t_hi = rlen < slen - d ? rlen : slen - d;               // read positions [t_lo, t_hi) face window positions t + d
			uint64_t E[4] = {0, 0, 0, 0};
#pragma unroll
			for (int w = 0; w < kRescueMaxRead / 32; ++w) {                // (fixed trip count: E[] stays in registers)
				int base = w << 5;
				if (base >= t_hi || base + 32 <= t_lo) continue;
				int wp = base + d;                                      // window position facing read position `base` (negative: masked below)
				int idx = wp >> 5;                                      // floor division (arithmetic shift)
				int sh = (wp & 31) << 1;
				uint64_t lo_w = idx >= 0 ? win2[idx] : 0, hi_w = idx + 1 >= 0 ? win2[idx + 1] : 0;
				uint64_t ww = sh ? (lo_w >> sh) | (hi_w << (64 - sh)) : lo_w;
				uint64_t diff = rd2[w] ^ ww;
				uint64_t eq = ~(diff | (diff >> 1)) & 0x5555555555555555ull;   // bit 2b set: base b equal
				int b0 = t_lo > base ? t_lo - base : 0, b1 = t_hi - base < 32 ? t_hi - base : 32;
				eq &= (b1 >= 32 ? ~0ull : ((1ull << (b1 << 1)) - 1)) & ~((1ull << (b0 << 1)) - 1);
				// 2 bits per base -> 1 bit per base
				uint64_t x = eq;
				x = (x | (x >> 1)) & 0x3333333333333333ull;
				x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
				x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
				x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
				x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
				E[w >> 1] |= x << ((w & 1) << 5);
			}
			// R[p]: positions p .. p+9 all equal
			auto shr = [](const uint64_t *v, int k, uint64_t *o) {     // o = v >> k over 256 bits, 0 < k < 64
				o[0] = (v[0] >> k) | (v[1] << (64 - k)); o[1] = (v[1] >> k) | (v[2] << (64 - k)); o[2] = (v[2] >> k) | (v[3] << (64 - k)); o[3] = v[3] >> k;
			};
			uint64_t T[4], R2[4], R[4];
			shr(E, 1, T);
			for (int i = 0; i < 4; ++i) R2[i] = E[i] & T[i];            // >= 2
			shr(R2, 2, T);
			for (int i = 0; i < 4; ++i) R[i] = R2[i] & T[i];            // >= 4
			shr(R, 4, T);
			for (int i = 0; i < 4; ++i) R[i] &= T[i];                   // >= 8
			shr(R2, 8, T);
			for (int i = 0; i < 4; ++i) R[i] &= T[i];                   // >= 10
			if ((R[0] | R[1] | R[2] | R[3]) == 0) continue;
			// the maximal runs of at least 10: each starts at the lowest remaining bit of R
			for (;;) {
				int p = -1;
				for (int i = 0; i < 4; ++i)
					if (R[i]) { p = (i << 6) + __ffsll((unsigned long long)R[i]) - 1; break; }
				if (p < 0) break;
				int e = p + 10;                                         // extend while the bases stay equal
				while (e < 256 && ((E[e >> 6] >> (e & 63)) & 1)) e++;
				int at_ = atomicAdd(&n_raw, 1);
				if (at_ < kRescueMaxRuns) { raw_key[at_] = (di << 8) | p; raw_len[at_] = e - p; }
				for (int c = p; c < e; ++c) R[c >> 6] &= ~(1ull << (c & 63));   // (positions of this run cannot start another)
			}
		}
		}
		// the runs in (diagonal, read position) order -- the order IdentifyCommonKmers' sort leaves the k-mer hits in: rank sort
		__syncthreads();
		const int total = n_raw;
		if (total > kRescueMaxRuns) {
			if (lane == 0) flag_host(a, t.read, WHY_RESCUE_RUNS);
			continue;
		}
		for (int i = lane; i < total; i += 64) {
			int key = raw_key[i], rank = 0;
			for (int j = 0; j < total; ++j) rank += raw_key[j] < key ? 1 : 0;
			run_d[rank] = d_lo + (key >> 8); run_t[rank] = key & 255; run_l[rank] = raw_len[i];
		}
		__syncthreads();
		if (lane == 0) {
			// IdnetifyRescueCandidate
			int best_s = 0, best_i = 0, best_j = 0;
			for (int i = 0; i < total;) {
				int s = run_l[i], j;
				for (j = i + 1; j < total; ++j) {
					if (run_d[j] - run_d[i] < a.max_gaps) s += run_l[j];
					else break;
				}
				if (s > best_s) { best_s = s; best_i = i; best_j = j; }
				i = j;
			}
			int cnt = best_j - best_i;
			if (best_s > t.score1) {
				if (cnt > kAlnMaxSeeds) flag_host(a, t.read, WHY_RESCUE_SEEDS);
				else {
					// the group's pairs by (gPos, rPos) (:61); text coordinates
					kg_seed *out = a.resc_seeds + (int64_t)ti * kAlnMaxSeeds;
					for (int k = 0; k < cnt; ++k) {
						kg_seed sd;
						sd.rPos = run_t[best_i + k]; sd.len = run_l[best_i + k]; sd.gPos = t.left + run_t[best_i + k] + run_d[best_i + k];
						int p = k;
						while (p > 0 && (out[p - 1].gPos > sd.gPos || (out[p - 1].gPos == sd.gPos && out[p - 1].rPos > sd.rPos))) { out[p] = out[p - 1]; --p; }
						out[p] = sd;
					}
					a.resc_count[ti] = cnt;
					a.resc_posdiff[ti] = (int64_t)run_d[best_i] + t.left;
					// the new candidate of mate 1 is mated with candidate j of mate 2 (:158-164)
					const CandList l1 = cand_list(a, t.read);
					a.c_score[slot] = best_s;
					a.c_mate[slot] = t.j;
					a.c_mate[a.cand_off[t.read + 1] + t.j] = l1.nd + t.ordinal;
				}
			}
		}
	}
}

// what follows RescueUnpairedAlignment for the pairs that had windows (src/Mapping.cpp:561-563)
__global__ __launch_bounds__(256) void aln_post_rescue_kernel(AlnArgs a)
{
	int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	const int64_t n = a.slow_pairs ? (int64_t)a.ctl[35] : a.n_reads;          // (only the first mate of a pair is ever pending)
	for (; x < n; x += stride) {
		const int64_t r = a.slow_pairs ? (int64_t)a.slow_pairs[x] << 1 : x;
		if (!a.r_pending[r] || a.r_host[r]) continue;
		const CandList l1 = cand_list(a, r), l2 = cand_list(a, r + 1);
		bool mated = false;
		for (int i = l1.nd; i < l1.n(); ++i) mated = mated || a.c_score[l1.at(i)] > 0;
		if (mated) remove_unmated(a, l1, l2);
		remove_redundant(a, l1);
		remove_redundant(a, l2);
	}
}

// ---- normal pairs ------------------------------------------------------------------------------------------------------------
namespace {

struct Pairs {                      // vector<SeedPair_t> of one candidate, in the lane's private memory
	int64_t gPos[kAlnMaxPairs];
	int32_t rPos[kAlnMaxPairs];
	int32_t rLen[kAlnMaxPairs], gLen[kAlnMaxPairs];
	uint8_t simple[kAlnMaxPairs];
	int num;
};

// the same vector in a group's LDS block (aln_plan_group_kernel): the algorithms below are templates over either
struct PairsRef {
	int64_t *gPos;
	int32_t *rPos, *rLen, *gLen;
	uint8_t *simple;
	int num;
};

template <class P>
__device__ __forceinline__ void erase_empty(P &v)
{
	int w = 0;
	for (int i = 0; i < v.num; ++i)
		if (v.rLen[i] != 0) {
			if (w != i) { v.gPos[w] = v.gPos[i]; v.rPos[w] = v.rPos[i]; v.rLen[w] = v.rLen[i]; v.gLen[w] = v.gLen[i]; v.simple[w] = v.simple[i]; }
			w++;
		}
	v.num = w;
}

// RemoveTandemRepeatSeeds, src/AlignmentCandidates.cpp:235-260: every read position hit by more than one seed goes
template <class P>
__device__ void remove_tandem_repeats(P &v)
{
	if (v.num < 2) return;
	bool any = false;
	uint32_t drop = 0;
	for (int i = 0; i < v.num; ++i)
		for (int j = i + 1; j < v.num; ++j)
			if (v.rPos[i] == v.rPos[j]) { drop |= (1u << i) | (1u << j); any = true; }
	if (!any) return;
	for (int i = 0; i < v.num; ++i)
		if ((drop >> i) & 1) v.rLen[i] = v.gLen[i] = 0;
	erase_empty(v);
}

// RemoveTranslocatedSeeds, src/AlignmentCandidates.cpp:262-321.  ord[k] = index (in genome order) of the seed with the k-th
// smallest read position (read positions are distinct once the tandem repeats are gone)
template <class P>
__device__ void remove_translocated(P &v)
{
	const int num = v.num;
	if (num < 2) return;
	int ord[kAlnMaxSeeds];
	for (int i = 0; i < num; ++i) {
		int x = i, p = i;
		while (p > 0 && v.rPos[ord[p - 1]] > v.rPos[x]) { ord[p] = ord[p - 1]; --p; }
		ord[p] = x;
	}
	bool any = false;
	for (int i = 0; i < num; ++i) {
		if (ord[i] == i) continue;
		any = true;
		int hi = ord[i];
		for (int j = i + 1; j <= hi; ++j)
			if (ord[j] > hi) hi = ord[j];
		int s1 = 0, s2 = 0;
		for (int k = i; k <= hi; ++k) {
			if (k < ord[k]) s1 += v.rLen[ord[k]];
			else s2 += v.rLen[ord[k]];
		}
		for (int k = i; k <= hi; ++k) {
			bool drop = s1 > s2 ? k > ord[k] : k < ord[k];
			if (drop) v.rLen[ord[k]] = v.gLen[ord[k]] = 0;
		}
		i = hi;
	}
	if (any) erase_empty(v);
}

// CheckSeedOverlapping, src/AlignmentCandidates.cpp:323-373
template <class P>
__device__ bool resolve_overlap(P &v, int i, int j)
{
	bool master = true;
	int ov;
	if ((ov = v.rPos[i] + v.rLen[i] - v.rPos[j]) > 0) {
		if (v.rLen[i] < v.rLen[j]) {
			master = false;
			if (v.rLen[i] > ov) v.gLen[i] = (v.rLen[i] -= ov);
			else v.rLen[i] = v.gLen[i] = 0;
		} else if (v.rLen[j] > ov) {
			v.rPos[j] += ov; v.gPos[j] += ov; v.gLen[j] = (v.rLen[j] -= ov);
		} else v.rLen[j] = v.gLen[j] = 0;
	}
	if (v.rLen[i] > 0 && v.rLen[j] > 0 && (ov = (int)(v.gPos[i] + v.gLen[i] - v.gPos[j])) > 0) {
		if (v.gLen[i] < v.gLen[j]) {
			master = false;
			if (v.rLen[i] > ov) v.gLen[i] = (v.rLen[i] -= ov);
			else v.rLen[i] = v.gLen[i] = 0;
		} else if (v.rLen[j] > ov) {
			v.rPos[j] += ov; v.gPos[j] += ov; v.gLen[j] = (v.rLen[j] -= ov);
		} else v.rLen[j] = v.gLen[j] = 0;
	}
	return master;
}

// CheckOverlappingSeeds, src/AlignmentCandidates.cpp:375-418
template <class P>
__device__ void check_overlaps(P &v)
{
	const int num = v.num;
	if (num < 2) return;
	bool any = false;
	for (int i = 0; i < num;) {
		if (v.rLen[i] > 0) {
			int r_end = v.rPos[i] + v.rLen[i] - 1;
			int64_t g_end = v.gPos[i] + v.gLen[i] - 1;
			for (int j = i + 1; j < num; ++j) {
				if (v.rLen[j] == 0) continue;
				if (r_end < v.rPos[j] && g_end < v.gPos[j]) break;
				if (!resolve_overlap(v, i, j)) break;
			}
			if (v.rLen[i] == 0) {
				any = true;
				int q = i - 1;
				while (q > 0 && v.rLen[q] == 0) q--;
				i = q < 0 ? 0 : q;
			} else i++;
		} else {
			any = true;
			i++;
		}
	}
	if (any) erase_empty(v);
}

__device__ __forceinline__ bool by_gpos_less(int64_t g1, int r1, int64_t g2, int r2)   // CompByGenomePos, :17-21
{
	return g1 == g2 ? r1 < r2 : g1 < g2;
}

// IdentifyNormalPairs(rlen, glen, v), src/AlignmentCandidates.cpp:420-490 (glen = -1 for a read against the genome, the
// fragment's genome length inside GenerateNormalPairAlignment).  false: more gap pairs than the envelope holds
template <class P>
__device__ bool identify_normal_pairs(int rlen, int glen, P &v)
{
	if (v.num > 1) {
		remove_tandem_repeats(v);
		remove_translocated(v);
		check_overlaps(v);
		const int num = v.num;
		int added = 0;
		for (int i = 0, j = 1; j < num; ++i, ++j) {
			int r_gap = v.rPos[j] - (v.rPos[i] + v.rLen[i]);
			if (r_gap < 0) r_gap = 0;
			int g_gap = (int)(v.gPos[j] - (v.gPos[i] + v.gLen[i]));
			if (g_gap < 0) g_gap = 0;
			if (r_gap > 0 || g_gap > 0) {
				if (added == kAlnMaxGaps) return false;
				int t = num + added++;
				v.simple[t] = 0;
				v.rPos[t] = v.rPos[i] + v.rLen[i];
				v.gPos[t] = v.gPos[i] + v.gLen[i];
				v.rLen[t] = r_gap; v.gLen[t] = g_gap;
			}
		}
		// the appended gap pairs go between the seeds in (gPos, rPos) order: insertion, as the host does for up to 8 of them
		for (int t = num; t < num + added; ++t) {
			int64_t xg = v.gPos[t];
			int xr = v.rPos[t], xrl = v.rLen[t], xgl = v.gLen[t];
			uint8_t xs = v.simple[t];
			int p = t;
			while (p > 0 && by_gpos_less(xg, xr, v.gPos[p - 1], v.rPos[p - 1])) {
				v.gPos[p] = v.gPos[p - 1]; v.rPos[p] = v.rPos[p - 1]; v.rLen[p] = v.rLen[p - 1]; v.gLen[p] = v.gLen[p - 1]; v.simple[p] = v.simple[p - 1];
				--p;
			}
			v.gPos[p] = xg; v.rPos[p] = xr; v.rLen[p] = xrl; v.gLen[p] = xgl; v.simple[p] = xs;
		}
		v.num = num + added;
	}
	if (v.num > 0) {
		int r_gap = v.rPos[0] > 0 ? v.rPos[0] : 0;
		int g_gap = glen > 0 ? (int)v.gPos[0] : r_gap;            // glen = -1: the genome gap is the read gap (:458)
		if (r_gap > 0 || g_gap > 0) {
			for (int p = v.num; p > 0; --p) {
				v.gPos[p] = v.gPos[p - 1]; v.rPos[p] = v.rPos[p - 1]; v.rLen[p] = v.rLen[p - 1]; v.gLen[p] = v.gLen[p - 1]; v.simple[p] = v.simple[p - 1];
			}
			int64_t g = v.gPos[1] - g_gap;
			v.gPos[0] = g < 0 ? 0 : g;                             // (the reference's follow-up "gGaps += gPos" adds zero, :464)
			v.rPos[0] = 0; v.rLen[0] = r_gap; v.gLen[0] = g_gap; v.simple[0] = 0;
			v.num++;
		}
		int last = v.num - 1;
		r_gap = rlen - (v.rPos[last] + v.rLen[last]);
		g_gap = glen > 0 ? (int)(glen - (v.gPos[last] + v.gLen[last])) : r_gap;
		if (r_gap > 0 || g_gap > 0) {
			int t = v.num++;
			v.simple[t] = 0;
			v.rPos[t] = v.rPos[last] + v.rLen[last];
			v.gPos[t] = v.gPos[last] + v.gLen[last];
			v.rLen[t] = r_gap; v.gLen[t] = g_gap;
		}
	}
	return true;
}

// CheckCoordinateValidity, src/AlignmentCandidates.cpp:582-610
template <class P>
__device__ bool coordinates_valid(const AlnArgs &a, const P &v)
{
	int64_t g1 = 0, g2 = a.two_genome_size;
	for (int i = 0; i < v.num; ++i)
		if (v.gLen[i] > 0) { g1 = v.gPos[i]; break; }
	for (int i = v.num; i-- > 0;)
		if (v.gLen[i] > 0) { g2 = v.gPos[i] + v.gLen[i] - 1; break; }
	const int64_t L = a.genome_size;
	if ((g1 < L && g2 >= L) || (g1 >= L && g2 < L)) return false;
	int i1 = end_lower_bound(a, g1), i2 = end_lower_bound(a, g2);
	if (i1 == a.n_ends || i2 == a.n_ends || a.end_chr[i1] != a.end_chr[i2]) return false;
	return true;
}

// ---- CIGAR building --------------------------------------------------------------------------------------------------------
struct Cigar {                      // vector<pair<int,char>> cigar_vec
	int32_t len[kAlnMaxCigar];
	char op[kAlnMaxCigar];
	int n;
	bool overflow;
	__device__ __forceinline__ void push(int l, char o)
	{
		if (n < kAlnMaxCigar) { len[n] = l; op[n] = o; n++; }
		else overflow = true;
	}
};

// One column of the aligned strings nw_alignment leaves behind (src/nw_alignment.cpp:59-72), rebuilt from the op string.
// AddNewCigarElements and CheckLocalAlignmentQuality look at the CHARACTERS (a literal '-' in a read counts as a gap there),
// so the columns are characters here too.
struct Columns {
	const uint8_t *ops;
	int len;
	const uint8_t *rd;              // read characters of the fragment
	int64_t g;                      // text coordinate of the fragment
};

// The columns of an op string one after the other: the op, the read's character and the text's of each ('-' for a gap).  Ops and read characters come
// eight per load, text bases 32 per load (all three buffers have slack behind their last byte) -- a byte load each per column, each waited for, was most
// of what aln_finish_kernel did with its time.
struct __attribute__((packed, aligned(1))) ColU64u { uint64_t v; };
struct ColCursor {
	int t, ri, gi;                  // the next column; read / text characters consumed before it
	int op_at, rd_at, tx_at;        // the first index each loaded word holds
	uint64_t opw, rdw, txw;
	__device__ __forceinline__ ColCursor(int t0, int ri0, int gi0) : t(t0), ri(ri0), gi(gi0), op_at(t0 - 8), rd_at(ri0 - 8), tx_at(gi0 - 32), opw(0), rdw(0), txw(0) {}
	__device__ __forceinline__ void next(const AlnArgs &a, const Columns &c, char &c1, char &c2)
	{
		if (t - op_at >= 8) { op_at = t; opw = reinterpret_cast<const ColU64u *>(c.ops + t)->v; }
		const uint8_t op = (uint8_t)(opw >> (8 * (t - op_at)));
		if (op == KG_OP_GAP1) c1 = '-';
		else {
			if (ri - rd_at >= 8) { rd_at = ri; rdw = reinterpret_cast<const ColU64u *>(c.rd + ri)->v; }
			c1 = (char)(uint8_t)(rdw >> (8 * (ri - rd_at)));
			ri++;
		}
		if (op == KG_OP_GAP2) c2 = '-';
		else {
			if (gi - tx_at >= 32) { tx_at = gi; txw = text_word32(a, c.g + gi); }
			const int code = (int)((txw >> (2 * (gi - tx_at))) & 3);
			c2 = code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T';
			gi++;
		}
		t++;
	}
};

// AddNewCigarElements over columns [from, to), src/tools.cpp:49-104; (ri, gi) = characters consumed before `from`
__device__ int add_cigar_columns(const AlnArgs &a, const Columns &c, int from, int to, int ri, int gi, Cigar &cig)
{
	char state = '*';
	int cnt = 0, score = 0;
	ColCursor k(from, ri, gi);
	while (k.t < to) {
		char c1, c2;
		k.next(a, c, c1, c2);
		char st;
		if (c1 == '-') st = 'D';
		else if (c2 == '-') st = 'I';
		else { st = 'M'; if (c1 == c2) score++; }
		if (st == state) cnt++;
		else {
			if (cnt > 0) cig.push(cnt, state);
			cnt = 1;
			state = st;
		}
	}
	if (cnt > 0) cig.push(cnt, state);
	return score;
}

// CheckLocalAlignmentQuality, src/tools.cpp:255-290
__device__ bool local_quality_ok(const AlnArgs &a, const Columns &c)
{
	int type = -1, n = 0, mis = 0, runs = 0;
	ColCursor k(0, 0, 0);
	while (k.t < c.len) {
		char c1, c2;
		k.next(a, c, c1, c2);
		int ty;
		if (c1 == '-') ty = 0;
		else if (c2 == '-') ty = 1;
		else { ty = 2; n++; if (c1 != c2) mis++; }
		if (ty != type) { type = ty; runs++; }
	}
	return !(runs >= 4 || (mis >= 3 && mis >= (int)(n * 0.3)));
}

// ProcessHeadSequencePair after the alignment, src/tools.cpp:314-339: leading gaps of either string are trimmed
__device__ int finish_head(const AlnArgs &a, const Columns &c, int64_t &gPos, int &gLen, int &rPos, int &rLen, Cigar &cig)
{
	if (!local_quality_ok(a, c)) { cig.push(rLen, 'S'); return 0; }
	ColCursor k(0, 0, 0);
	// leading '-' of the read side: genome characters without a partner
	int p = 0;
	while (k.t < c.len) {
		ColCursor k2 = k;
		char c1, c2;
		k2.next(a, c, c1, c2);
		if (c1 != '-') break;
		k = k2; p++;
	}
	if (p > 0) { gPos += p; gLen -= p; }
	p = 0;
	while (k.t < c.len) {
		ColCursor k2 = k;
		char c1, c2;
		k2.next(a, c, c1, c2);
		if (c2 != '-') break;
		k = k2; p++;
	}
	if (p > 0) { rPos += p; rLen -= p; cig.push(p, 'S'); }
	return add_cigar_columns(a, c, k.t, c.len, k.ri, k.gi, cig);
}

// ProcessTailSequencePair after the alignment, src/tools.cpp:366-394
__device__ int finish_tail(const AlnArgs &a, const Columns &c, int &gLen, int &rLen, Cigar &cig)
{
	if (!local_quality_ok(a, c)) { cig.push(rLen, 'S'); return 0; }
	// the characters of the last column are known after a walk over all columns before it (the tail is short: one walk per trimmed column)
	int end = c.len;
	auto last_chars = [&](int upto, char &c1, char &c2) {
		ColCursor k(0, 0, 0);
		c1 = 0; c2 = 0;
		while (k.t < upto) k.next(a, c, c1, c2);
	};
	// trailing '-' of the read side
	int cnt = 0;
	for (;;) {
		if (end <= 0) break;
		char c1, c2;
		last_chars(end, c1, c2);
		if (c1 != '-') break;
		end--; cnt++;
	}
	if (cnt > 0) gLen -= cnt;
	int cnt2 = 0;
	for (;;) {
		if (end <= 0) break;
		char c1, c2;
		last_chars(end, c1, c2);
		if (c2 != '-') break;
		end--; cnt2++;
	}
	if (cnt2 > 0) rLen -= cnt2;
	int score = add_cigar_columns(a, c, 0, end, 0, 0, cig);
	if (cnt2 > 0) cig.push(cnt2, 'S');
	return score;
}

// what pass 1 decided for a pair
enum : uint8_t { W_NONE = 0, W_SIMPLE = 1, W_IMMEDIATE = 2, W_JOB = 3, W_PLAN = 4, W_PENDING = 5, W_INLINE = 6 };
constexpr int kInlineJobs = 3;          // gap fragments of at most 8 x 8 a candidate may align in its own lane (aln_plan_kernel)
struct Work {
	uint8_t kind[kAlnMaxPairs];
	uint8_t op[kAlnMaxPairs];
	int32_t op_len[kAlnMaxPairs];
	int32_t val[kAlnMaxPairs];      // IMMEDIATE: score (-1 = the > 3000 soft clip); JOB: job index
};

// GenMappingReport's pair loop and tail for one candidate (src/AlignmentCandidates.cpp:657-722): CIGAR, AlnScore, coordinates.
// `first`: the read is the first of its pair (or single).  Returns false when the result does not fit the record.
__device__ bool finish_candidate(const AlnArgs &a, int64_t cand, bool first, const uint8_t *rd, Pairs &v, const Work &w)
{
	const int num = v.num;
	Cigar cig;
	cig.n = 0; cig.overflow = false;
	int score = 0;
	for (int j = 0; j < num; ++j) {
		if (w.kind[j] == W_NONE) continue;
		if (w.kind[j] == W_SIMPLE) {
			cig.push(v.rLen[j], 'M');
			score += v.rLen[j];
			continue;
		}
		const bool head = j == 0, tail = j == num - 1 && !head;
		int s;
		if (w.kind[j] == W_IMMEDIATE) {
			if (w.op[j] != 0) cig.push(w.op_len[j], (char)w.op[j]);
			s = w.val[j];
		} else {
			Columns c;
			if (w.kind[j] == W_INLINE) {
				// aligned in the planning lane itself (nw8_inline): op string number val & 255 -- parked in the candidate's own CIGAR slot, which is
				// written only after the last pair has been read (a lane-private array here sends the compiler's SimplifyCFG pass into a fault: ops
				// would point into two address spaces) --, val >> 8 columns
				c.ops = reinterpret_cast<const uint8_t *>(a.rep_cigar + cand * KG_ALN_CIGAR_MAX) + 16 * (w.val[j] & 255);
				c.len = w.val[j] >> 8;
			} else if (w.kind[j] == W_JOB) {
				const NwJobDesc jd = a.jobs[w.val[j]];
				c.ops = a.nw_ops + jd.ops;
				c.len = a.nw_len[w.val[j]];
			} else {
				// the partitioned fragment: literal runs and the sub-fragments' op strings, one after the other (src/tools.cpp:165-208)
				const AlnPlan pl = a.plans[w.val[j]];
				// (the bytes leave eight at a time -- an accumulator, literal runs as a fill pattern, job op strings through 8-byte loads -- as in
				//  frag_stitch_kernel: one byte load and store per column was most of this branch)
				uint8_t *out = a.nw_ops + pl.ops;
				int at = 0, na = 0;
				uint64_t acc = 0;
				auto put = [&](uint64_t b) {
					acc |= b << (8 * na);
					if (++na == 8) { reinterpret_cast<ColU64u *>(out + at)->v = acc; at += 8; acc = 0; na = 0; }
				};
				for (int k = 0; k < pl.count; ++k) {
					const AlnPiece pc = a.pieces[pl.first + k];
					if (pc.kind <= KG_OP_GAP2) {
						int n = pc.v;
						while (na != 0 && n > 0) { put((uint64_t)pc.kind); --n; }
						const uint64_t pat = (uint64_t)pc.kind * 0x0101010101010101ull;
						for (; n >= 8; n -= 8) { reinterpret_cast<ColU64u *>(out + at)->v = pat; at += 8; }
						for (; n > 0; --n) put((uint64_t)pc.kind);
					} else {
						const uint8_t *src = a.nw_ops + a.jobs[pc.v].ops;
						const int L = a.nw_len[pc.v];
						int t = 0;
						while (na != 0 && t < L) put((uint64_t)src[t++]);
						for (; t + 8 <= L; t += 8) { reinterpret_cast<ColU64u *>(out + at)->v = reinterpret_cast<const ColU64u *>(src + t)->v; at += 8; }
						for (; t < L; ++t) put((uint64_t)src[t]);
					}
				}
				for (int k = 0; k < na; ++k) out[at + k] = (uint8_t)(acc >> (8 * k));
				c.ops = out;
				c.len = at + na;
			}
			c.rd = rd + v.rPos[j];
			c.g = v.gPos[j];
			if (head) s = finish_head(a, c, v.gPos[j], v.gLen[j], v.rPos[j], v.rLen[j], cig);
			else if (tail) s = finish_tail(a, c, v.gLen[j], v.rLen[j], cig);
			else s = add_cigar_columns(a, c, 0, c.len, 0, 0, cig);
		}
		if (head) {
			if (s > 0) score += s;
			if (s <= 0) { v.gPos[0] = v.gPos[1]; v.gLen[0] = 0; }         // :674-686
		} else if (tail) {
			if (s > 0) score += s;
			if (s <= 0) { v.gPos[j] = v.gPos[j - 1] + v.gLen[j - 1]; v.gLen[j] = 0; }
		} else score += s;
	}
	if (cig.overflow) return false;
	a.rep_chr[cand] = 0;
	a.rep_pos[cand] = 0;
	a.rep_fwd[cand] = 1;
	a.rep_cigar_len[cand] = 0;
	if (cig.n > 1) {                                                     // GapPenalty, :612-622, :701-706
		int gp = 0;
		for (int i = 0; i < cig.n; ++i)
			if (cig.op[i] == 'I' || cig.op[i] == 'D') gp += cig.len[i];
		score -= gp;
		if (score <= 0) { a.rep_score[cand] = 0; a.c_score[cand] = -1; return true; }     // (c_score -1: "continue" before the best/second-best step)
	}
	if (cig.n == 0) score = 0;
	else {
		// GenCoordinateInfo, :515-562
		const int64_t gPos = v.gPos[0], end_gPos = v.gPos[num - 1] + v.gLen[num - 1] - 1;
		bool fwd;
		int chr;
		int64_t pos;
		bool rev = false;
		if (gPos < a.genome_size) {
			fwd = first;
			if (a.n_chr == 1) { chr = 0; pos = gPos + 1; }
			else {
				int it = end_lower_bound(a, gPos);
				chr = a.end_chr[it];
				pos = gPos + 1 - a.chr_fwd_start[chr];
			}
		} else {
			fwd = !first;
			rev = true;
			if (a.n_chr == 1) { chr = 0; pos = a.two_genome_size - end_gPos; }
			else {
				int it = end_lower_bound(a, gPos);
				if (it == a.n_ends) it = a.n_ends - 1;
				pos = a.contig_end[it] - end_gPos + 1;
				chr = a.end_chr[it];
			}
		}
		// GenerateCIGAR, :492-513 (the reverse strand shows the elements in reverse order)
		char *out = a.rep_cigar + cand * KG_ALN_CIGAR_MAX;
		int at = 0;
		char state = 0;
		int cnt = 0;
		bool fits = true;
		auto emit = [&](int nn, char st) {
			char buf[12];
			int k = 0;
			do { buf[k++] = (char)('0' + nn % 10); nn /= 10; } while (nn);
			if (at + k + 1 > KG_ALN_CIGAR_MAX - 1) { fits = false; return; }
			while (k) out[at++] = buf[--k];
			out[at++] = st;
		};
		for (int q = 0; q < cig.n; ++q) {
			int i = rev ? cig.n - 1 - q : q;
			if (cig.op[i] != state) {
				if (cnt > 0) emit(cnt, state);
				cnt = cig.len[i];
				state = cig.op[i];
			} else cnt += cig.len[i];
		}
		if (cnt > 0) emit(cnt, state);
		if (!fits) return false;
		a.rep_cigar_len[cand] = (uint8_t)at;
		a.rep_chr[cand] = chr;
		a.rep_pos[cand] = pos;
		a.rep_fwd[cand] = fwd ? 1 : 0;
		if (pos <= 0) score = 0;
	}
	a.rep_score[cand] = score;
	return true;
}

}  // namespace


// The runs plan_partition's scalar loop finds, bit-parallel: the read fragment (<= 256 characters) and the text around the
// genome fragment as 2 bits per base in registers; per diagonal one XOR per 32 bases, the equality bits compressed to one per
// base (256-bit vector), the positions where 8 consecutive bits are set by shift-and doubling, the maximal runs read off in
// increasing read position.  Same runs, same order as the scalar loop (which stays for MaxGaps > 32 and for fragments at
// the very start of the text).  -1: a non-ACGT character in the read fragment, or more runs than the envelope takes.
__device__ int partition_runs_packed(const AlnArgs &a, const uint8_t *f1, int64_t g, int rL, int gL, int mg, Pairs &v)
{
	uint64_t RD[8], TW[10];
	bool ok = true;
	const uint64_t k7f = 0x7F7F7F7F7F7F7F7Full;
#pragma unroll
	for (int w = 0; w < 8; ++w) {
		uint64_t acc = 0;
#pragma unroll
		for (int h = 0; h < 4; ++h) {
			const int t0 = 32 * w + 8 * h;
			if (t0 < rL) {
				uint64_t x = reinterpret_cast<const AlnU64u *>(f1 + t0)->v;                 // 8 characters (the buffer has slack behind the last read)
				const int nv = rL - t0;
				const uint64_t keep = nv >= 8 ? ~0ull : (1ull << (8 * nv)) - 1;
				uint64_t u = x & 0xDFDFDFDFDFDFDFDFull;
				// 0x80 in every byte that equals the constant (exact per byte, no borrow between bytes)
				uint64_t ya = u ^ 0x4141414141414141ull, yc = u ^ 0x4343434343434343ull, yg = u ^ 0x4747474747474747ull, yt = u ^ 0x5454545454545454ull;
				uint64_t good = ~((((ya & k7f) + k7f) | ya) & (((yc & k7f) + k7f) | yc) & (((yg & k7f) + k7f) | yg) & (((yt & k7f) + k7f) | yt)) & 0x8080808080808080ull;
				if ((good & keep) != (0x8080808080808080ull & keep)) ok = false;
				uint64_t c2 = (x >> 1) & 0x0303030303030303ull;                              // A 0, C 1, G 3, T 2 ...
				c2 ^= (c2 >> 1) & 0x0101010101010101ull;                                     // ... one Gray step from the codes 0..3
				c2 &= keep;
				c2 = (c2 | (c2 >> 6)) & 0x000F000F000F000Full;
				c2 = (c2 | (c2 >> 12)) & 0x000000FF000000FFull;
				c2 = (c2 | (c2 >> 24)) & 0xFFFFull;
				acc |= c2 << (16 * h);
			}
		}
		RD[w] = acc;
	}
	if (!ok) return -1;
	const int64_t g0 = g - (int64_t)(mg - 1);
	const int words = (rL + 31) >> 5;
#pragma unroll
	for (int k = 0; k < 10; ++k) TW[k] = k <= words + 1 ? text_word32(a, g0 + 32 * k) : 0;
	for (int d = -(mg - 1); d <= mg - 1; ++d) {
		const int t_lo = d < 0 ? -d : 0;
		const int t_hi = rL < gL - d ? rL : gL - d;
		if (t_hi - t_lo < 8) continue;
		const int s = d + mg - 1;
		const bool far = (s >> 5) != 0;
		const int sb = (s & 31) << 1;
		uint64_t E[4] = {0, 0, 0, 0};
#pragma unroll
		for (int w = 0; w < 8; ++w) {
			const int base = w << 5;
			if (base >= t_hi || base + 32 <= t_lo) continue;
			uint64_t lo = far ? TW[w + 1] : TW[w], hi = far ? TW[w + 2] : TW[w + 1];
			uint64_t tw = sb ? (lo >> sb) | (hi << (64 - sb)) : lo;
			uint64_t diff = RD[w] ^ tw;
			uint64_t eq = ~(diff | (diff >> 1)) & 0x5555555555555555ull;             // bit 2b set: base b equal
			int b0 = t_lo > base ? t_lo - base : 0, b1 = t_hi - base < 32 ? t_hi - base : 32;
			eq &= (b1 >= 32 ? ~0ull : ((1ull << (b1 << 1)) - 1)) & ~((1ull << (b0 << 1)) - 1);
			uint64_t x = eq;
			x = (x | (x >> 1)) & 0x3333333333333333ull;
			x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
			x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
			x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
			x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
			E[w >> 1] |= x << ((w & 1) << 5);
		}
		auto shr = [](const uint64_t *q, int k, uint64_t *o) {     // o = q >> k over 256 bits, 0 < k < 64
			o[0] = (q[0] >> k) | (q[1] << (64 - k)); o[1] = (q[1] >> k) | (q[2] << (64 - k)); o[2] = (q[2] >> k) | (q[3] << (64 - k)); o[3] = q[3] >> k;
		};
		uint64_t T[4], R[4];
		shr(E, 1, T);
		for (int i = 0; i < 4; ++i) R[i] = E[i] & T[i];             // >= 2
		shr(R, 2, T);
		for (int i = 0; i < 4; ++i) R[i] &= T[i];                   // >= 4
		shr(R, 4, T);
		for (int i = 0; i < 4; ++i) R[i] &= T[i];                   // >= 8
		while ((R[0] | R[1] | R[2] | R[3]) != 0) {
			int pos = R[0] ? __ffsll((unsigned long long)R[0]) - 1 : R[1] ? 64 + __ffsll((unsigned long long)R[1]) - 1
			        : R[2] ? 128 + __ffsll((unsigned long long)R[2]) - 1 : 192 + __ffsll((unsigned long long)R[3]) - 1;
			int e = pos + 8;                                        // extend while the bases stay equal
			while (e < 256 && ((E[e >> 6] >> (e & 63)) & 1)) e++;
			if (v.num == kAlnMaxSeeds) return -1;
			int k = v.num++;
			v.rPos[k] = pos; v.gPos[k] = pos + d; v.rLen[k] = v.gLen[k] = e - pos; v.simple[k] = 1;
			for (int c = pos; c < e; ++c) R[c >> 6] &= ~(1ull << (c & 63));   // (positions of this run cannot start another)
		}
	}
	return v.num;
}

// GenerateNormalPairAlignment for a fragment pair with both sides > 30 (src/tools.cpp:146-212; non-PacBio: MaxShift = MaxGaps):
// GenerateSimplePairsFromFragmentPair -- the common 8-mers of the two fragments whose positions differ by less than MaxShift,
// merged into exact matches of at least 8 bases (src/KmerAnalysis.cpp:104-179) -- then IdentifyNormalPairs(rLen, gLen, ...) on
// them, and per resulting piece either a literal stretch or a sub-fragment alignment.
// In two halves around the reservation of its list entries (one wave_reserve per list for the whole wave, aln_partition_kernel):
// partition_compute returns 1: the pair has a plan of n_pieces pieces, n_jobs of them sub-fragment alignments, ops_need op bytes in all;
// 0: the partition is empty (the caller aligns the whole fragment); -1: outside the envelope (host).
__device__ int partition_compute(const AlnArgs &a, const uint8_t *f1, int64_t g, int rL, int gL, Pairs &v, int &n_pieces, int &n_jobs, int &ops_need)
{
	v.num = 0;
	n_pieces = n_jobs = ops_need = 0;
	const int mg = a.max_gaps;
	if (mg >= 1 && mg <= 32 && g >= (int64_t)(mg - 1) && rL <= 256) {
		int rc = partition_runs_packed(a, f1, g, rL, gL, mg, v);
		if (rc < 0) return -1;
	} else {
	// the 8-mer code maps characters through nst_nt4_table and skips 'N': plain A/C/G/T (either case) is what the comparison
	// of 2-bit codes below reproduces
	for (int i = 0; i < rL; ++i) {
		unsigned u = f1[i] & 0xDFu;
		if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T')) return -1;
	}
	// runs of at least 8 equal bases along the diagonals |gpos - rpos| < MaxShift, in (diagonal, read position) order
	for (int d = -(mg - 1); d <= mg - 1; ++d) {
		int t_lo = d < 0 ? -d : 0;
		int t_hi = rL < gL - d ? rL : gL - d;
		int run = 0;
		for (int t = t_lo; t <= t_hi; ++t) {
			bool eq = false;
			if (t < t_hi) {
				unsigned ch = f1[t];
				unsigned c1 = (ch >> 1) & 3;
				c1 ^= c1 >> 1;
				eq = (int)c1 == text_code(a, g + t + d);
			}
			if (eq) run++;
			else {
				if (run >= 8) {
					if (v.num == kAlnMaxSeeds) return -1;
					int k = v.num++;
					v.rPos[k] = t - run; v.gPos[k] = t - run + d; v.rLen[k] = v.gLen[k] = run; v.simple[k] = 1;
				}
				run = 0;
			}
		}
	}
	}
	if (v.num == 0) return 0;
	// sort(SimplePairVec, CompByGenomePos), src/KmerAnalysis.cpp:177
	for (int i = 1; i < v.num; ++i) {
		int64_t xg = v.gPos[i];
		int xr = v.rPos[i], xl = v.rLen[i];
		int p = i;
		while (p > 0 && by_gpos_less(xg, xr, v.gPos[p - 1], v.rPos[p - 1])) { v.gPos[p] = v.gPos[p - 1]; v.rPos[p] = v.rPos[p - 1]; v.rLen[p] = v.gLen[p] = v.rLen[p - 1]; --p; }
		v.gPos[p] = xg; v.rPos[p] = xr; v.rLen[p] = v.gLen[p] = xl;
	}
	if (!identify_normal_pairs(rL, gL, v)) return -1;
	if (v.num == 0) return 0;
	// the pieces; op strings: the assembled one (at most rL + gL columns) and one per sub-fragment
	ops_need = rL + gL;
	for (int i = 0; i < v.num; ++i) {
		if (v.rLen[i] <= 0 && v.gLen[i] <= 0) continue;
		n_pieces++;
		bool lit = v.gLen[i] == 0 || v.rLen[i] == 0 || (v.rLen[i] == 1 && v.gLen[i] == 1) || v.simple[i];
		if (!lit) { n_jobs++; ops_need += v.rLen[i] + v.gLen[i]; }
	}
	return 1;
}

// the plan of partition_compute into the entries reserved for it; false: a list is full (host) -- what was reserved INSIDE the job list is
// left as empty jobs then (the NW kernels walk every job below the counter: none may be an earlier batch's)
__device__ bool partition_write(const AlnArgs &a, int64_t enc_off, int64_t g, int rL, int gL, const Pairs &v, int n_pieces, int n_jobs, int ops_need,
                                unsigned long long plan_at, unsigned long long piece_at, unsigned long long job_at, unsigned long long ops_at, int32_t &plan_index)
{
	if (plan_at >= (unsigned long long)a.job_capacity || piece_at + n_pieces > 4ull * (unsigned long long)a.job_capacity ||
	    job_at + n_jobs > (unsigned long long)a.job_capacity || ops_at + ops_need > (unsigned long long)a.ops_capacity) {
		for (unsigned long long k = job_at; k < job_at + (unsigned long long)n_jobs && k < (unsigned long long)a.job_capacity; ++k) { NwJobDesc jd; jd.o1 = 0; jd.o2 = 0; jd.ops = 0; jd.m = 0; jd.n = 0; a.jobs[k] = jd; }
		return false;
	}
	AlnPlan pl;
	pl.ops = (int64_t)ops_at; pl.first = (int32_t)piece_at; pl.count = n_pieces;
	a.plans[plan_at] = pl;
	unsigned long long ops_next = ops_at + (unsigned long long)(rL + gL);
	int pk = 0, jk = 0;
	for (int i = 0; i < v.num; ++i) {
		const int prl = v.rLen[i], pgl = v.gLen[i];
		if (prl <= 0 && pgl <= 0) continue;
		AlnPiece pc;
		if (pgl == 0) { pc.kind = KG_OP_GAP2; pc.v = prl; }                     // read bases against '-' (:170-174)
		else if (prl == 0) { pc.kind = KG_OP_GAP1; pc.v = pgl; }                // '-' against genome bases (:176-180)
		else if ((prl == 1 && pgl == 1) || v.simple[i]) { pc.kind = KG_OP_DIAG; pc.v = prl; }   // copied as they are (:182-186, :192)
		else {
			NwJobDesc jd;
			jd.o1 = enc_off + v.rPos[i]; jd.o2 = g + v.gPos[i]; jd.ops = (int64_t)ops_next; jd.m = prl; jd.n = pgl;
			ops_next += (unsigned long long)(prl + pgl);
			a.jobs[job_at + jk] = jd;
			pc.kind = 3; pc.v = (int32_t)(job_at + jk);
			jk++;
		}
		a.pieces[piece_at + pk++] = pc;
	}
	plan_index = (int32_t)plan_at;
	return true;
}

// ---- pass 1a: the candidates whose report needs no alignment and no private arrays -------------------------------------------------
// Most candidates of 150 bp reads at 1 % error are a few seeds ON ONE DIAGONAL, in order, without overlap, separated by single
// substituted bases: IdentifyNormalPairs (src/AlignmentCandidates.cpp:420-490) then removes nothing and only inserts the gap pairs
// between them (equal read and genome length), every gap pair is decided without nw_alignment -- the <= 2-mismatch shortcut or the
// 1 x 1 case of Process{Head,Normal,Tail}SequencePair (src/tools.cpp:240, 301, 352) -- every CIGAR element is an M, and
// GenMappingReport's result is: AlnScore = seed bases + matching gap bases, CIGAR "<rlen>M", the coordinate of the first pair
// (of the second when the head pair scored nothing, :674-686; likewise the tail).  This kernel decides exactly those candidates in
// registers -- 87 % of aln_plan_kernel's wave cycles were waits on its per-lane arrays in scratch memory (profiles/r03w) -- and
// lists every other candidate, untouched, for aln_plan_kernel (dense: its lanes all walk the general path).  KG_ALN_NO_FAST: off.
constexpr int kFastSeeds = 6;        // seeds of a candidate this kernel takes
constexpr int kFastGap = 2048;       // longest gap it looks at (head / tail gaps of candidates at repeat copies run to most of the read)

// mismatches of the read characters rd[0 .. L) against the text at g (raw characters as CalFragPairMismatchBases compares them,
// src/tools.cpp:40-47), counted up to `stop` (the decisions below only ask "at most 2?"); dash: a literal '-' in the first character
// (the 1 x 1 case then goes to nw_alignment, src/tools.cpp:229-233)
__device__ __forceinline__ int fast_gap_mismatches(const AlnArgs &a, const uint8_t *rd, int64_t g, int L, int stop, bool &dash)
{
	int n = 0;
	dash = rd[0] == '-';
	for (int b0 = 0; b0 < L && n < stop; b0 += 32) {
		const uint64_t tw = text_word32(a, g + b0);
		const int lim = L - b0 < 32 ? L - b0 : 32;
		for (int i0 = 0; i0 < lim && n < stop; i0 += 8) {
			const uint64_t w = reinterpret_cast<const AlnU64u *>(rd + b0 + i0)->v;          // (the character array has 64 bytes of slack)
			const int m = lim - i0 < 8 ? lim - i0 : 8;
			for (int i = 0; i < m; ++i) {
				const int c = (int)((w >> (8 * i)) & 255);
				const int code = (int)((tw >> (2 * (i0 + i))) & 3);
				const int t = code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T';
				n += c != t ? 1 : 0;
			}
		}
	}
	return n;
}

// A gap pair of L bases on the diagonal (read and genome side alike), role 0 = head, 1 = between seeds, 2 = tail: what
// Process{Head,Normal,Tail}SequencePair decide WITHOUT an alignment (src/tools.cpp:225-397), as aln_plan_kernel's pair loop does:
//   >= 0        : an 'M' element of L bases scoring that many identical ones (the <= 2-mismatch shortcut :240 / :301 / :352, or 1 x 1)
//   kFastClip   : the whole gap soft-clipped, score 0 (a head beyond 50 bases :307-311, a tail beyond 100 :358-362)
//   kFastSlow   : nw_alignment / the 8-mer partition / the > 3000 clip are due: the general kernel's
constexpr int kFastClip = -1, kFastSlow = -2;
__device__ __forceinline__ int fast_gap_value(const AlnArgs &a, const uint8_t *rd, int64_t g, int L, int role)
{
	if (role != 1 && L > 3000) return kFastSlow;
	bool dash = false;
	const int n = fast_gap_mismatches(a, rd, g, L, 3, dash);
	if (n <= 2 && n <= (int)(L * 0.2)) return L - n;
	if ((role == 0 && L > 50) || (role == 2 && L > 100)) return kFastClip;
	if (L == 1 && !dash) return 0;                               // one base against one other base: 1M, nothing identical
	return kFastSlow;
}

// What aln_plan_fast_kernel decides for one candidate (see above), as a value: both that kernel and aln_trivial_kernel use it.
struct FastRep {
	int state;                       // 0: decided (the fields below hold GenMappingReport's result), 1: the general kernel's, 2: CheckCoordinateValidity failed
	int score, chr, cigar_len;
	int64_t pos;
	bool fwd;
	uint64_t t0, t1;                 // the CIGAR text, at most 16 characters
};
enum { FAST_DECIDED = 0, FAST_SLOW = 1, FAST_INVALID = 2 };

// lower_bound(g): index of the first ChrLocMap key >= g; end_at(i): key i (both read the block's copy in the LDS when it holds the keys)
template <class LowerBound, class EndAt>
__device__ __forceinline__ FastRep fast_report(const AlnArgs &a, int count, const kg_seed *seeds, int64_t rbase, int rlen, bool first, LowerBound lower_bound, EndAt end_at)
{
	FastRep o;
	o.state = FAST_SLOW; o.score = 0; o.chr = 0; o.cigar_len = 0; o.pos = 0; o.fwd = true; o.t0 = o.t1 = 0;
	bool slow = count < 1 || count > kFastSeeds || rlen > 4000;
	// ---- the seeds: one diagonal, in order, no overlap; gaps of at most a text word ----
	int64_t d = 0;
	int prev_end = 0, first_r = 0, seed_bases = 0;
	int gap_at[kFastSeeds + 1], gap_len[kFastSeeds + 1];      // (indexed by unrolled constants: registers)
#pragma unroll
	for (int i = 0; i < kFastSeeds; ++i) {
		gap_at[i] = 0; gap_len[i] = 0;
		if (!slow && i < count) {
			const kg_seed sd = seeds[i];
			const int64_t di = sd.gPos - (int64_t)sd.rPos;
			if (i == 0) { d = di; first_r = sd.rPos; if (di < 0 || sd.rPos > kFastGap) slow = true; }
			else {
				if (di != d || sd.rPos < prev_end || sd.rPos - prev_end > kFastGap) slow = true;
				gap_at[i] = prev_end; gap_len[i] = sd.rPos - prev_end;
			}
			prev_end = sd.rPos + sd.len;
			seed_bases += sd.len;
		}
	}
	const int tail_len = rlen - prev_end;
	if (tail_len < 0 || tail_len > kFastGap) slow = true;
	if (slow) return o;
	// ---- CheckCoordinateValidity (:582-610) on [d, d + rlen - 1]: one strand copy, one contig ----
	const int64_t g1 = d, g2 = d + rlen - 1, L = a.genome_size;
	const int i1 = lower_bound(g1);
	bool valid = !((g1 < L && g2 >= L) || (g1 >= L && g2 < L)) && i1 < a.n_ends;
	if (valid && g2 > end_at(i1)) {
		const int i2 = lower_bound(g2);
		valid = i2 < a.n_ends && a.end_chr[i1] == a.end_chr[i2];
		if (valid) return o;          // (two keys of one contig cannot lie in one strand copy: never taken; the general path decides)
	}
	if (!valid) { o.state = FAST_INVALID; return o; }          // no report, and no best/second-best step (:647)
	// ---- the gap pairs ----
	const uint8_t *rd = a.enc + rbase;
	int score = seed_bases;
	int head_val = 1, tail_val = 1;          // (> 0: scored; 0: an M element without identical bases; kFastClip: soft-clipped)
	if (first_r > 0) { head_val = fast_gap_value(a, rd, d, first_r, 0); slow = head_val == kFastSlow; score += head_val > 0 ? head_val : 0; }
#pragma unroll
	for (int i = 1; i < kFastSeeds; ++i)
		if (!slow && i < count && gap_len[i] > 0) {
			const int v = fast_gap_value(a, rd + gap_at[i], d + gap_at[i], gap_len[i], 1);
			slow = v == kFastSlow;
			score += v > 0 ? v : 0;
		}
	if (!slow && tail_len > 0) { tail_val = fast_gap_value(a, rd + prev_end, d + prev_end, tail_len, 2); slow = tail_val == kFastSlow; score += tail_val > 0 ? tail_val : 0; }
	if (slow) return o;
	// ---- GenMappingReport's tail: GenCoordinateInfo (:515-562), GenerateCIGAR (:492-513) ----
	const int64_t gPos = head_val > 0 ? d : d + first_r;                     // a head pair that scored nothing gives its place up (:674-686)
	const int64_t end_gPos = (tail_val > 0 ? d + rlen : d + prev_end) - 1;   // ... and so does the tail pair
	bool fwd;
	int chr;
	int64_t pos;
	const bool rev = gPos >= L;
	if (!rev) {
		fwd = first;
		if (a.n_chr == 1) { chr = 0; pos = gPos + 1; }
		else { chr = a.end_chr[i1]; pos = gPos + 1 - a.chr_fwd_start[chr]; }
	} else {
		fwd = !first;
		if (a.n_chr == 1) { chr = 0; pos = a.two_genome_size - end_gPos; }
		else { pos = end_at(i1) - end_gPos + 1; chr = a.end_chr[i1]; }
	}
	// the elements: [head S] M [tail S] -- every M element merges into one; the reverse strand shows them in reverse order
	const int clip_h = head_val == kFastClip ? first_r : 0, clip_t = tail_val == kFastClip ? tail_len : 0;
	const int e_len[3] = {rev ? clip_t : clip_h, rlen - clip_h - clip_t, rev ? clip_h : clip_t};
	char out[16];
	int at = 0;
#pragma unroll
	for (int q = 0; q < 3; ++q) {
		int nn = e_len[q];
		if (nn <= 0) continue;
		char buf[4];
		int k = 0;
		do { buf[k++] = (char)('0' + nn % 10); nn /= 10; } while (nn);
		while (k) out[at++] = buf[--k];
		out[at++] = q == 1 ? 'M' : 'S';
	}
	uint64_t t0 = 0, t1 = 0;
#pragma unroll
	for (int q = 0; q < 16; ++q) {
		const uint64_t ch = q < at ? (uint64_t)(uint8_t)out[q] : 0ull;
		if (q < 8) t0 |= ch << (8 * q); else t1 |= ch << (8 * (q - 8));
	}
	o.state = FAST_DECIDED;
	o.t0 = t0; o.t1 = t1; o.cigar_len = at;
	o.chr = chr; o.pos = pos; o.fwd = fwd;
	o.score = pos <= 0 ? 0 : score;
	return o;
}

// The candidates the planning kernels walk: every chained candidate and the slots of the rescue windows -- or, when aln_trivial_kernel
// has decided the trivial pairs, the candidates of the OTHER pairs (a.slow_cands, ctl[34] of them) and the rescue slots.
__device__ __forceinline__ int64_t plan_slots(const AlnArgs &a)
{
	unsigned long long n_tasks = a.ctl[4];
	if (n_tasks > (unsigned long long)a.task_capacity) n_tasks = (unsigned long long)a.task_capacity;
	return (a.slow_cands ? (int64_t)a.ctl[34] : a.n_cands) + (int64_t)n_tasks;
}
__device__ __forceinline__ int64_t slot_cand(const AlnArgs &a, int64_t slot)
{
	if (!a.slow_cands) return slot;
	const int64_t n = (int64_t)a.ctl[34];
	return slot < n ? (int64_t)a.slow_cands[slot] : a.n_cands + (slot - n);
}

__global__ __launch_bounds__(256) void aln_plan_fast_kernel(AlnArgs a)
{
	__shared__ int64_t s_end[128];
	const bool ends_in_lds = a.n_ends <= 128;
	if (ends_in_lds)
		for (int i = threadIdx.x; i < a.n_ends; i += blockDim.x) s_end[i] = a.contig_end[i];
	__syncthreads();
	auto end_at = [&](int i) { return ends_in_lds ? s_end[i] : a.contig_end[i]; };
	auto lower_bound = [&](int64_t g) {
		int lo = 0, hi = a.n_ends;
		while (lo < hi) {
			int mid = (lo + hi) >> 1;
			if (end_at(mid) < g) lo = mid + 1; else hi = mid;
		}
		return lo;
	};
	const int64_t n_all = plan_slots(a);
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t slot0 = (int64_t)blockIdx.x * blockDim.x; slot0 < n_all; slot0 += stride) {
		const int64_t slot = slot0 + threadIdx.x;
		bool slow = false;
		int64_t cand = 0;
		if (slot < n_all) {
			cand = a.plan_order ? (int64_t)a.plan_order[slot] : slot_cand(a, slot);
			a.rep_score[cand] = 0; a.rep_chr[cand] = 0; a.rep_pos[cand] = 0; a.rep_fwd[cand] = 1; a.rep_cigar_len[cand] = 0;
			const int64_t r = a.c_read[cand];
			if (!a.r_host[r] && a.c_score[cand] != 0) {
				const bool rescued = cand >= a.n_cands;
				int count;
				const kg_seed *seeds;
				if (!rescued) { const kg_candidate cd = a.cands[cand]; count = cd.count; seeds = a.cand_seeds + cd.first; }
				else { const int64_t t = cand - a.n_cands; count = a.resc_count[t]; seeds = a.resc_seeds + t * kAlnMaxSeeds; }
				const int64_t rbase = a.read_off[r];
				const int rlen = (int)(a.read_off[r + 1] - rbase);
				const int ck = chunk_of(a, r);
				const bool first = a.chunk_paired[ck] ? (((r - a.chunk_off[ck]) & 1) == 0) : true;
				const FastRep o = fast_report(a, count, seeds, rbase, rlen, first, lower_bound, end_at);
				slow = o.state == FAST_SLOW;
				if (o.state == FAST_INVALID) a.c_score[cand] = -1;
				else if (o.state == FAST_DECIDED) {
					uint64_t *dst = reinterpret_cast<uint64_t *>(a.rep_cigar + cand * KG_ALN_CIGAR_MAX);
					dst[0] = o.t0;
					if (o.cigar_len > 8) dst[1] = o.t1;
					a.rep_cigar_len[cand] = (uint8_t)o.cigar_len;
					a.rep_chr[cand] = o.chr;
					a.rep_pos[cand] = o.pos;
					a.rep_fwd[cand] = o.fwd ? 1 : 0;
					a.rep_score[cand] = o.score;
				}
			}
		}
		// ---- everything else: listed for the general kernel, densely ----
		const uint64_t mask = __ballot(slow);
		if (mask) {
			const int leader = __ffsll((unsigned long long)mask) - 1;
			unsigned long long at = 0;
			if ((int)(threadIdx.x & 63) == leader) at = atomicAdd(&a.ctl[32], (unsigned long long)__popcll(mask));
			at = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(at >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)at, leader);
			const uint64_t below = (threadIdx.x & 63) == 0 ? 0ull : (~0ull >> (64 - (threadIdx.x & 63)));
			if (slow) a.plan_slow[at + (unsigned long long)__popcll(mask & below)] = (int32_t)cand;
		}
	}
}

// ---- pass 0: the trivial pairs, start to finish, one pair per lane ---------------------------------------------------------------
// At 150 bp / 1 % error three pairs in four are trivial: each mate has ONE candidate, the two are each other's mate under
// CheckPairedAlignmentCandidates (src/Mapping.cpp:348-400: 0 <= PosDiff2 - PosDiff1 < EstDistance), and both candidates are the kind
// aln_plan_fast_kernel decides in registers.  For such a pair everything between chaining and the record is a function of ~200 bytes:
//   RemoveUnMatedAlignmentCandidates adds the two scores (:402-427), RemoveRedundantCandidates sees one candidate (:317-346),
//   GenMappingReport yields the two reports (fast_report), score > sub_score = 0 on both mates, the best candidates are mated so
//   CheckPairedFinalAlignments leaves at once (:429-438; with -m its loops change nothing for one candidate per mate),
//   SetPairedAlignmentFlag takes its first branch (:78-93), EvaluateMAPQ answers 60 (:160-175), OutputPairedAlignments prints the two
//   records with RNEXT / PNEXT / TLEN (:177-270) and counts the pair into iPaired / iDistance (:209-213).
// The per-candidate arrays (c_score, c_mate, rep_*) are never written or read for these pairs, none of the later kernels sees them:
// the pairs this kernel does NOT take are listed (a.slow_pairs, ctl[35]) with their candidates (a.slow_cands, ctl[34]), and
// aln_pair / aln_post_rescue / aln_bin / aln_plan_fast / aln_plan / aln_final walk those lists, densely.  The two 112-byte records
// of a lane are assembled in registers, staged through the LDS 32 pairs at a time and leave as whole 16-byte chunks of consecutive
// memory (aln_final_kernel writes a record field by field: 64 lanes, 64 lines per store).  KG_ALN_NO_TRIVIAL: off.
static_assert(sizeof(kg_aln_record) == 112 && offsetof(kg_aln_record, kind) == 16 && offsetof(kg_aln_record, est_lo) == 44 && offsetof(kg_aln_record, has_mate) == 52 &&
              offsetof(kg_aln_record, cigar) == 56 && offsetof(kg_aln_record, next) == 104 && offsetof(kg_aln_record, primary) == 108, "aln_trivial_kernel lays the record out by hand");

__device__ __forceinline__ void stage_record(uint32_t *w, int64_t pos, int64_t mate_pos, int flag, int chr, int tlen, int score, int est_lo, bool flip, const FastRep &o)
{
	w[0] = (uint32_t)(uint64_t)pos; w[1] = (uint32_t)((uint64_t)pos >> 32);
	w[2] = (uint32_t)(uint64_t)mate_pos; w[3] = (uint32_t)((uint64_t)mate_pos >> 32);
	w[4] = KG_ALN_MAPPED; w[5] = (uint32_t)flag; w[6] = (uint32_t)chr; w[7] = 60; w[8] = (uint32_t)tlen;
	w[9] = (uint32_t)score; w[10] = 0;                           // score, sub_score
	w[11] = (uint32_t)est_lo; w[12] = 0x7fffffffu;               // the pair's own EstDistance interval (est_lo, est_hi]
	w[13] = 1u | ((flip ? 1u : 0u) << 8) | ((uint32_t)o.cigar_len << 16);      // has_mate, flip, cigar_len, rescue = 0
	w[14] = (uint32_t)o.t0; w[15] = (uint32_t)(o.t0 >> 32); w[16] = (uint32_t)o.t1; w[17] = (uint32_t)(o.t1 >> 32);
#pragma unroll
	for (int k = 18; k < 26; ++k) w[k] = 0;
	w[26] = 0xffffffffu;                                         // next = -1
	w[27] = 1;                                                   // primary, pad
}

// One PAIR per lane.  (A form with one READ per lane -- the mates in neighbouring lanes, exchanging position, strand and length by a shuffle, half
// the chain of dependent loads per lane -- was built and measured slower, 62 against 39 ms per 100 M-read step: the staging, the flush, the
// ballots and the statistics are per wave, and a wave then covers 32 pairs instead of 64; profiles/r06g_*.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void aln_trivial_kernel(AlnArgs a)
{
	__shared__ int64_t s_end[128];
	__shared__ __attribute__((aligned(16))) uint32_t s_rec[4][32 * 2 * 28];          // per wave: the records of 32 pairs (7168 bytes)
	const bool ends_in_lds = a.n_ends <= 128;
	if (ends_in_lds)
		for (int i = threadIdx.x; i < a.n_ends; i += blockDim.x) s_end[i] = a.contig_end[i];
	__syncthreads();
	auto end_at = [&](int i) { return ends_in_lds ? s_end[i] : a.contig_end[i]; };
	auto lower_bound = [&](int64_t g) {
		int lo = 0, hi = a.n_ends;
		while (lo < hi) {
			int mid = (lo + hi) >> 1;
			if (end_at(mid) < g) lo = mid + 1; else hi = mid;
		}
		return lo;
	};
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t *const stage = s_rec[wave];
	const int64_t n_pairs = a.n_reads >> 1;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	const long long est = a.est_distance;
	for (int64_t u0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); u0 < n_pairs; u0 += stride) {      // u0: the wave's first pair
		const int64_t u = u0 + lane;
		const bool live = u < n_pairs;
		const int64_t r = u << 1;
		bool trivial = false;
		int n1 = 0, n2 = 0;
		int64_t c1 = 0;
		FastRep o1, o2;
		o1.state = o2.state = FAST_SLOW;
		long long dist = 0;
		int rl1 = 0, rl2 = 0;
		if (live) {
			c1 = a.cand_off[r];
			const int64_t c2 = a.cand_off[r + 1], c3 = a.cand_off[r + 2];
			n1 = (int)(c2 - c1); n2 = (int)(c3 - c2);
			if (n1 == 1 && n2 == 1) {
				const kg_candidate k1 = a.cands[c1], k2 = a.cands[c2];
				dist = k2.posDiff - k1.posDiff;
				// each is the other's only and best mate (score > 0; a tie or a better rival needs a second candidate), :362-391
				if (k1.score > 0 && k2.score > 0 && dist >= 0 && dist < est) {
					const int64_t b1 = a.read_off[r], b2 = a.read_off[r + 1], b3 = a.read_off[r + 2];
					rl1 = (int)(b2 - b1); rl2 = (int)(b3 - b2);
					o1 = fast_report(a, k1.count, a.cand_seeds + k1.first, b1, rl1, true, lower_bound, end_at);
					if (o1.state == FAST_DECIDED && o1.score > 0)
						o2 = fast_report(a, k2.count, a.cand_seeds + k2.first, b2, rl2, false, lower_bound, end_at);
					trivial = o1.state == FAST_DECIDED && o2.state == FAST_DECIDED && o1.score > 0 && o2.score > 0 && o1.score <= kAlnMaxScore && o2.score <= kAlnMaxScore;
				}
			}
		}
		// ---- the pairs left to the general kernels, and their candidates, densely ----
		{
			const bool slow = live && !trivial;
			const uint64_t mask = __ballot(slow);
			if (mask) {
				const int nc = slow ? n1 + n2 : 0;
				int pre = nc;                                   // inclusive prefix sum of the lanes' candidate counts
#pragma unroll
				for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(pre, off); if (lane >= off) pre += t; }
				const int total = __shfl(pre, 63);
				const int leader = __ffsll((unsigned long long)mask) - 1;
				unsigned long long at_p = 0, at_c = 0;
				if (lane == leader) {
					at_p = atomicAdd(&a.ctl[35], (unsigned long long)__popcll(mask));
					if (total) at_c = atomicAdd(&a.ctl[34], (unsigned long long)total);
				}
				at_p = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(at_p >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)at_p, leader);
				at_c = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(at_c >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)at_c, leader);
				if (slow) {
					const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
					a.slow_pairs[at_p + (unsigned long long)__popcll(mask & below)] = (int32_t)u;
					int32_t *dst = a.slow_cands + at_c + (unsigned long long)(pre - nc);
					for (int k = 0; k < nc; ++k) dst[k] = (int32_t)(c1 + k);
				}
			}
		}
		// ---- the trivial pairs' records ----
		const uint64_t tmask = __ballot(trivial);
		if (tmask == 0) continue;
		int tl = 0;
		long long ad = 0;
		if (trivial) {
			tl = (int)(o2.pos - o1.pos + (o1.fwd ? rl2 : 0 - rl1));      // :204-207
			ad = tl < 0 ? -(long long)tl : (long long)tl;
			if (ad >= 10000) ad = 0;                                        // :211
		}
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			const bool mine = trivial && (lane >> 5) == half;
			if (mine) {
				uint32_t *w = stage + (lane & 31) * 56;
				stage_record(w, o1.pos, o2.pos, 0x43 | (o1.fwd ? 0x20 : 0x10), o1.chr, tl, o1.score, (int)dist, !o1.fwd, o1);
				stage_record(w + 28, o2.pos, o1.pos, 0x83 | (o2.fwd ? 0x20 : 0x10), o2.chr, 0 - tl, o2.score, (int)dist, o2.fwd, o2);
			}
			// (a wave's LDS traffic is in program order; the fences keep the compiler from moving the reads above the writes of OTHER lanes)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const uint32_t hm = (uint32_t)(tmask >> (32 * half));
			if (hm == 0) continue;
			uint4 *const out = reinterpret_cast<uint4 *>(a.records + ((u0 + 32 * half) << 1));
			const uint4 *const in = reinterpret_cast<const uint4 *>(stage);
#pragma unroll
			for (int it = 0; it < 7; ++it) {
				const int q = it * 64 + lane;                   // 16-byte chunk of the half's 7168 bytes; 14 chunks per pair
				if ((hm >> (q / 14)) & 1u) out[q] = in[q];
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
		// ---- what the pairs add to their chunks: iPaired / iDistance, the reads of MAPQ 60, the chunk's EstDistance interval ----
		int ck = -1;
		if (trivial) ck = chunk_of(a, r);
		const int ck0 = __shfl(ck, __ffsll((unsigned long long)tmask) - 1);
		if (__ballot(trivial && ck != ck0) == 0) {
			long long lo = trivial ? dist : -1, sum = ad;
			for (int off = 32; off > 0; off >>= 1) {
				const long long l2 = __shfl_xor(lo, off);
				lo = l2 > lo ? l2 : lo;
				sum += __shfl_xor(sum, off);
			}
			if (lane == 0) {
				kg_chunk_stats &cs = a.chunk_stats[ck0];
				const int np = __popcll(tmask);
				atomicAdd((unsigned long long *)&cs.paired, 2ull * (unsigned long long)np);
				if (sum) atomicAdd((unsigned long long *)&cs.distance, (unsigned long long)sum);
				atomicAdd(&cs.unique, 2 * np);
				atomicMax((long long *)&cs.lo, lo);
			}
		} else if (trivial) {
			kg_chunk_stats &cs = a.chunk_stats[ck];
			atomicAdd((unsigned long long *)&cs.paired, 2ull);
			if (ad) atomicAdd((unsigned long long *)&cs.distance, (unsigned long long)ad);
			atomicAdd(&cs.unique, 2);
			atomicMax((long long *)&cs.lo, dist);
		}
		if (lane == 0) atomicAdd(&a.ctl[36], (unsigned long long)__popcll(tmask));      // (pairs decided here, this batch)
	}
}

// nw_alignment for a fragment pair of at most 8 x 8 (97 % of the gap fragments of 150 bp reads), in the planning lane's own registers:
// the recurrences, the boundary values and the traceback's tie order of nw_small8_kernel (nw_kernels.hip; reference src/nw_alignment.cpp:18-80 on
// doubled scores) -- so that a candidate whose only alignments are such fragments is finished where it is planned instead of being parked
// (spill slot out and in, job descriptors, a pass of the NW kernels, aln_finish_kernel's pass over it).  ops[0 .. len) = the columns left to right.
__device__ __forceinline__ int nw8_inline(const AlnArgs &a, const uint8_t *f1, int64_t g, int m, int n, uint8_t *ops)
{
	constexpr int kNeg = -(1 << 20);
	const uint64_t w1 = reinterpret_cast<const AlnU64u *>(f1)->v;            // (the character array has 64 bytes of slack)
	const uint32_t tw = (uint32_t)text_word32(a, g);
	int c2[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) c2[j] = j < n ? (int)((tw >> (2 * j)) & 3u) : 8 + j;
	int S[9], T[9];
	S[0] = 0; T[0] = 0;
#pragma unroll
	for (int j = 1; j <= 8; ++j) { S[j] = -2 - j; T[j] = kNeg; }
	uint64_t fr = 0, ft = 0;          // bit 8 (i - 1) + (j - 1): s == r / s == t at cell (i, j)
#pragma unroll
	for (int i = 1; i <= 8; ++i) {
		if (i <= m) {
			const unsigned u = (unsigned)((w1 >> (8 * (i - 1))) & 0xDFu);
			const int c1 = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;          // nst_nt4_table
			int diag = S[0];
			S[0] = -2 - i;
			int left_s = S[0], left_r = kNeg;
#pragma unroll
			for (int j = 1; j <= 8; ++j) {
				const int up_s = S[j], up_t = T[j];
				const int r = max(left_r - 1, left_s - 3);
				const int tt = max(up_t - 1, up_s - 3);
				const int d = diag + (c1 == c2[j - 1] ? 3 : -3);
				const int sc = max(d, max(r, tt));
				fr |= (uint64_t)(sc == r) << (8 * (i - 1) + (j - 1));
				ft |= (uint64_t)(sc == tt) << (8 * (i - 1) + (j - 1));
				diag = up_s; S[j] = sc; T[j] = tt; left_s = sc; left_r = r;
			}
		}
	}
	// the traceback yields the columns right to left (:59-72: s == r first, then s == t, else the diagonal)
	int i = m, j = n, len = 0;
	uint64_t lo = 0, hi = 0;
	while (i > 0 || j > 0) {
		const int bit = 8 * (i - 1) + (j - 1);
		const bool g1 = i == 0 || (j > 0 && ((fr >> bit) & 1));
		const bool g2 = !g1 && (j == 0 || ((ft >> bit) & 1));
		const uint64_t op = g1 ? KG_OP_GAP1 : g2 ? KG_OP_GAP2 : KG_OP_DIAG;
		const int at = 15 - len;
		if (at >= 8) hi |= op << (8 * (at - 8)); else lo |= op << (8 * at);
		len++;
		if (g1) j--; else if (g2) i--; else { i--; j--; }
	}
	const int sh = 16 - len;                                                 // bytes to shift down
	if (sh >= 8) { lo = sh == 8 ? hi : sh == 16 ? 0 : hi >> (8 * (sh - 8)); hi = 0; }
	else if (sh > 0) { lo = (lo >> (8 * sh)) | (hi << (8 * (8 - sh))); hi >>= 8 * sh; }
	reinterpret_cast<AlnU64u *>(ops)->v = lo;
	reinterpret_cast<AlnU64u *>(ops + 8)->v = hi;
	return len;
}

// ---- pass 1: one candidate per lane -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void aln_plan_kernel(AlnArgs a)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	int64_t n_all = plan_slots(a);                                               // chained candidates (all, or those of the pairs aln_trivial_kernel left), then the slots of the rescue windows
	if (a.plan_slow) n_all = (int64_t)a.ctl[32];                                 // ... or what aln_plan_fast_kernel left
	// (the wave's lanes stay together through the loop: what the parked candidates need of the lists is reserved for all of them by one atomic per list)
	for (int64_t slot0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); slot0 < n_all; slot0 += stride) {
		const int64_t slot = slot0 + (threadIdx.x & 63);
		// what phase 1 leaves for phase 2 (a parked candidate)
		Pairs v;
		Work w;
		int64_t cand = 0, r = 0, rbase = 0;
		int num = 0, n_new_jobs = 0, new_ops = 0, n_pending = 0, n_small = 0, n_inline = 0;
		bool pending = false, inline_all = false, first_mate = true;
		// ---- phase 1: up to the point where the candidate is finished, handed to the host, or has to be parked ----
		bool park = false;
		do {
		if (slot >= n_all) break;
		// (binned: lanes of a wave then hold candidates with the same number of seeds -- the loops below run equally long)
		cand = a.plan_slow ? (int64_t)a.plan_slow[slot] : a.plan_order ? (int64_t)a.plan_order[slot] : slot_cand(a, slot);
		a.rep_score[cand] = 0; a.rep_chr[cand] = 0; a.rep_pos[cand] = 0; a.rep_fwd[cand] = 1; a.rep_cigar_len[cand] = 0;
		r = a.c_read[cand];
		if (a.r_host[r]) break;
		if (a.c_score[cand] == 0) break;                                  // GenMappingReport skips it, :643
		const bool rescued = cand >= a.n_cands;
		kg_candidate cd;
		const kg_seed *seeds;
		if (!rescued) { cd = a.cands[cand]; seeds = a.cand_seeds + cd.first; }
		else {
			int64_t t = cand - a.n_cands;
			cd.count = a.resc_count[t]; cd.first = 0; cd.posDiff = a.resc_posdiff[t]; cd.score = 0;
			seeds = a.resc_seeds + t * kAlnMaxSeeds;
		}
		if (cd.count > kAlnMaxSeeds) { flag_host(a, r, WHY_SEEDS); break; }
		rbase = a.read_off[r];
		const int rlen = (int)(a.read_off[r + 1] - rbase);
		const uint8_t *rd = a.enc + rbase;
		const int ck = chunk_of(a, r);
		const bool first = a.chunk_paired[ck] ? (((r - a.chunk_off[ck]) & 1) == 0) : true;
		first_mate = first;
		v.num = cd.count;
		for (int i = 0; i < cd.count; ++i) {
			kg_seed s = seeds[i];
			v.gPos[i] = s.gPos; v.rPos[i] = s.rPos; v.rLen[i] = v.gLen[i] = s.len; v.simple[i] = 1;
		}
		if (rlen > 4000) { flag_host(a, r, WHY_READ_LEN); break; }
		if (!identify_normal_pairs(rlen, -1, v)) { flag_host(a, r, WHY_GAPS); break; }
		if (!coordinates_valid(a, v)) { a.c_score[cand] = -1; break; }      // no report, and no best/second-best step (:647)
		num = v.num;
		bool host = false, jobs = false;
		int why = WHY_PARTITION;
		for (int j = 0; j < num && !host; ++j) {
			w.kind[j] = W_NONE; w.op[j] = 0; w.op_len[j] = 0; w.val[j] = 0;
			const int rL = v.rLen[j], gL = v.gLen[j];
			if (rL == 0 && gL == 0) continue;
			if (v.simple[j]) { w.kind[j] = W_SIMPLE; continue; }
			const int role = j == 0 ? 0 : j == num - 1 ? 2 : 1;
			if (role != 1 && rL > 3000) {                                      // :671-676, :690-695
				w.kind[j] = W_IMMEDIATE; w.op[j] = 'S'; w.op_len[j] = rL; w.val[j] = -1;
				continue;
			}
			if (role == 1 && (rL == 0 || gL == 0)) {                           // ProcessNormalSequencePair, src/tools.cpp:229-233
				w.kind[j] = W_IMMEDIATE;
				if (rL > 0) { w.op[j] = 'I'; w.op_len[j] = rL; }
				else if (gL > 0) { w.op[j] = 'D'; w.op_len[j] = gL; }
				continue;
			}
			const uint8_t *f1 = rd + v.rPos[j];
			if (rL == gL) {                                                     // the <= 2-mismatch shortcut, :240, :301, :352
				bool dash_ = false;
				int n = fast_gap_mismatches(a, f1, v.gPos[j], rL, 3, dash_);      // (eight characters per load, stops at the third mismatch: only <= 2 matter here)
				if (n <= 2 && n <= (int)(rL * 0.2)) {
					w.kind[j] = W_IMMEDIATE; w.op[j] = 'M'; w.op_len[j] = rL; w.val[j] = rL - n;
					continue;
				}
			}
			if ((role == 0 && rL > 50) || (role == 2 && rL > 100)) {           // :307-311, :358-362
				w.kind[j] = W_IMMEDIATE; w.op[j] = 'S'; w.op_len[j] = rL; w.val[j] = 0;
				continue;
			}
			if (rL == 1 && gL == 1 && f1[0] != '-') {
				// one base against one base: nw_alignment can only answer with the diagonal, the quality check passes a single
				// column, nothing is trimmed, AddNewCigarElements books 1M with one identical base iff the characters are equal
				w.kind[j] = W_IMMEDIATE; w.op[j] = 'M'; w.op_len[j] = 1; w.val[j] = (char)f1[0] == text_char(a, v.gPos[j]) ? 1 : 0;
				continue;
			}
			if (rL > kAlnMaxFrag || gL > kAlnMaxFrag || rL <= 0 || gL <= 0) { host = true; break; }
			if (rL > 30 && gL > 30) {
				if (a.dbg_no_partition) { host = true; break; }
				// GenerateNormalPairAlignment's 8-mer partition, src/tools.cpp:146-212: about one candidate in thirteen has such a
				// pair, so almost every wave would walk the long path for a few lanes -- the pair is handed to the dense
				// aln_partition_kernel instead (one task per lane), which writes its outcome into the parked candidate
				w.kind[j] = W_PENDING;
				jobs = true; pending = true; n_pending++;
				continue;
			}
			// nw_alignment(rL, frag1, gL, frag2): a job for the NW kernels (its slot and op bytes are reserved below, with everything else the candidate needs)
			w.kind[j] = W_JOB; w.val[j] = -1;
			n_new_jobs++; new_ops += rL + gL;
			n_small += (rL <= 8 && gL <= 8) ? 1 : 0;
			jobs = true;
		}
		if (host) { flag_host(a, r, why); break; }
		if (!jobs) {
			if (!finish_candidate(a, cand, first, rd, v, w)) flag_host(a, r, WHY_CIGAR);
			break;
		}
		inline_all = !pending && n_small == n_new_jobs && n_new_jobs <= kInlineJobs && !a.dbg_no_inline;
		park = !inline_all;
		} while (false);
		if (inline_all) {
			// every alignment the candidate needs is at most 8 x 8: they are made here, in this lane's registers, and the candidate is finished at
			// once -- nothing of it goes through the spill list, the job list, the NW kernels or aln_finish_kernel
			uint8_t *const inl = reinterpret_cast<uint8_t *>(a.rep_cigar + cand * KG_ALN_CIGAR_MAX);      // (3 x 16 bytes: the candidate's CIGAR slot, free until its text is written)
			int k = 0;
			for (int j = 0; j < num; ++j) {
				if (w.kind[j] != W_JOB) continue;
				const int len = nw8_inline(a, a.enc + rbase + v.rPos[j], v.gPos[j], v.rLen[j], v.gLen[j], inl + 16 * k);
				w.kind[j] = W_INLINE; w.val[j] = k | (len << 8);
				k++;
			}
			if (!finish_candidate(a, cand, first_mate, a.enc + rbase, v, w)) flag_host(a, r, WHY_CIGAR);
			n_inline = k;
		}
		// park the candidate until its alignments exist.  What the wave's candidates need of the lists -- a spill slot each, their NW jobs and op
		// bytes, their partition tasks -- is reserved with ONE atomic per list for the whole wave (wave_reserve)
		const unsigned long long sp = wave_reserve(&a.ctl[0], park ? 1ull : 0ull);
		unsigned long long job_at = wave_reserve(&a.ctl[1], park ? (unsigned long long)n_new_jobs : 0ull);
		unsigned long long ops_at = wave_reserve(&a.ctl[2], park ? (unsigned long long)new_ops : 0ull);
		unsigned long long task_at = wave_reserve(&a.ctl[3], park ? (unsigned long long)n_pending : 0ull);
		(void)wave_reserve(&a.ctl[37], (unsigned long long)n_inline);          // (tally: alignments made in the planning lanes, this batch)
		if (!park) continue;
		const bool sp_ok = sp < (unsigned long long)a.spill_capacity;
		const bool jobs_ok = job_at + (unsigned long long)n_new_jobs <= (unsigned long long)a.job_capacity && ops_at + (unsigned long long)new_ops <= (unsigned long long)a.ops_capacity;
		if (!sp_ok || !jobs_ok) {
			// A list is full: the read is the host's.  Everything this lane took INSIDE the lists is still written, whichever list overflowed --
			// aln_finish_kernel walks every spill slot below ctl[0], the NW kernels every job below ctl[1], aln_partition_kernel every task below
			// ctl[3], and none may find an earlier batch's entry there: the spill slot names this candidate with no pairs (its read is flagged,
			// aln_finish skips it), the job slots become empty jobs, the tasks name the flagged read
			if (sp_ok) { a.spill[sp].cand = (int32_t)cand; a.spill[sp].num = 0; }
			for (unsigned long long k = job_at; k < job_at + (unsigned long long)n_new_jobs && k < (unsigned long long)a.job_capacity; ++k) { NwJobDesc jd; jd.o1 = 0; jd.o2 = 0; jd.ops = 0; jd.m = 0; jd.n = 0; a.jobs[k] = jd; }
			flag_host(a, r, WHY_CAPACITY);
			for (unsigned long long k = task_at; k < task_at + (unsigned long long)n_pending && k < (unsigned long long)a.job_capacity; ++k) {
				PartTask pt;
				pt.enc_off = rbase; pt.g = 0; pt.spill = 0; pt.j = 0; pt.read = (int32_t)r; pt.rL = 0; pt.gL = 0;
				a.part_tasks[k] = pt;
			}
			continue;
		}
		for (int j = 0; j < num; ++j) {
			if (w.kind[j] != W_JOB) continue;
			const int rL = v.rLen[j], gL = v.gLen[j];
			NwJobDesc jd;
			jd.o1 = rbase + v.rPos[j]; jd.o2 = v.gPos[j]; jd.ops = (int64_t)ops_at; jd.m = rL; jd.n = gL;
			a.jobs[job_at] = jd;
			w.val[j] = (int32_t)job_at;
			job_at++; ops_at += (unsigned long long)(rL + gL);
		}
		AlnSpill &o = a.spill[sp];
		o.cand = (int32_t)cand;
		o.num = num;
		for (int j = 0; j < num; ++j) {
			AlnSpillPair q;
			q.gPos = v.gPos[j]; q.rPos = v.rPos[j]; q.rLen = (int16_t)v.rLen[j]; q.gLen = (int16_t)v.gLen[j];
			q.val = w.val[j]; q.kind = w.kind[j]; q.op = w.op[j]; q.op_len = (int16_t)w.op_len[j];
			o.p[j] = q;
		}
		if (pending) {
			// (about one parked candidate in thirteen: a second round for those)
			if (task_at + (unsigned long long)n_pending > (unsigned long long)a.job_capacity) flag_host(a, r, WHY_CAPACITY);
			for (int j = 0; j < num; ++j) {
				if (w.kind[j] != W_PENDING) continue;
				const unsigned long long t = task_at++;
				if (t >= (unsigned long long)a.job_capacity) break;          // (the read is the host's, flagged above; every slot inside the list is written)
				PartTask pt;
				pt.enc_off = rbase + v.rPos[j]; pt.g = v.gPos[j]; pt.spill = (int32_t)sp; pt.j = j; pt.read = (int32_t)r;
				pt.rL = (int16_t)v.rLen[j]; pt.gL = (int16_t)v.gLen[j];
				a.part_tasks[t] = pt;
			}
		}
	}
}

// ---- pass 1b: the 8-mer partitions, one task per lane (dense) -----------------------------------------------------------------
__global__ __launch_bounds__(256) void aln_partition_kernel(AlnArgs a)
{
	unsigned long long n = a.ctl[3];
	if (n > (unsigned long long)a.job_capacity) n = (unsigned long long)a.job_capacity;
	const int lane = threadIdx.x & 63;
	// (the wave's lanes stay together: the list entries of all of them are reserved by one atomic per list)
	for (unsigned long long t0 = (unsigned long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); t0 < n; t0 += (unsigned long long)gridDim.x * blockDim.x) {
		const unsigned long long t = t0 + lane;
		PartTask pt;
		pt.enc_off = 0; pt.g = 0; pt.spill = 0; pt.j = 0; pt.read = 0; pt.rL = 0; pt.gL = 0;
		bool go = false;
		if (t < n) { pt = a.part_tasks[t]; go = !a.r_host[pt.read]; }
		Pairs v;
		int n_pieces = 0, n_jobs = 0, ops_need = 0, pr = -3;          // (-3: no task)
		if (go) {
			pr = partition_compute(a, a.enc + pt.enc_off, pt.g, pt.rL, pt.gL, v, n_pieces, n_jobs, ops_need);
			if (pr < 0) flag_host(a, pt.read, WHY_PARTITION);
		}
		// planned: a plan, its pieces, its jobs, its op bytes; no common 8-mer survived: the whole fragment is one alignment (src/tools.cpp:214-221)
		const bool planned = pr == 1, whole = pr == 0;
		const unsigned long long plan_at = wave_reserve(&a.ctl[5], planned ? 1ull : 0ull);
		const unsigned long long piece_at = wave_reserve(&a.ctl[6], planned ? (unsigned long long)n_pieces : 0ull);
		const unsigned long long job_at = wave_reserve(&a.ctl[1], planned ? (unsigned long long)n_jobs : whole ? 1ull : 0ull);
		const unsigned long long ops_at = wave_reserve(&a.ctl[2], planned ? (unsigned long long)ops_need : whole ? (unsigned long long)(pt.rL + pt.gL) : 0ull);
		if (planned) {
			AlnSpillPair &q = a.spill[pt.spill].p[pt.j];
			int32_t plan_index = 0;
			if (partition_write(a, pt.enc_off, pt.g, pt.rL, pt.gL, v, n_pieces, n_jobs, ops_need, plan_at, piece_at, job_at, ops_at, plan_index)) { q.kind = W_PLAN; q.val = plan_index; }
			else flag_host(a, pt.read, WHY_CAPACITY);
		} else if (whole) {
			AlnSpillPair &q = a.spill[pt.spill].p[pt.j];
			if (job_at >= (unsigned long long)a.job_capacity || ops_at + (unsigned long long)(pt.rL + pt.gL) > (unsigned long long)a.ops_capacity) {
				if (job_at < (unsigned long long)a.job_capacity) { NwJobDesc jd; jd.o1 = 0; jd.o2 = 0; jd.ops = 0; jd.m = 0; jd.n = 0; a.jobs[job_at] = jd; }      // (inside the list: an empty job, not an earlier batch's)
				flag_host(a, pt.read, WHY_CAPACITY);
			} else {
				NwJobDesc jd;
				jd.o1 = pt.enc_off; jd.o2 = pt.g; jd.ops = (int64_t)ops_at; jd.m = pt.rL; jd.n = pt.gL;
				a.jobs[job_at] = jd;
				q.kind = W_JOB; q.val = (int32_t)job_at;
			}
		}
	}
}

// ---- pass 2: the parked candidates ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aln_finish_kernel(AlnArgs a)
{
	unsigned long long n = a.ctl[0];
	if (n > (unsigned long long)a.spill_capacity) n = (unsigned long long)a.spill_capacity;
	unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (; t < n; t += stride) {
		const AlnSpill &sp = a.spill[t];
		const int64_t cand = sp.cand;
		const int64_t r = a.c_read[cand];
		if (a.r_host[r]) continue;
		const int ck = chunk_of(a, r);
		const bool first = a.chunk_paired[ck] ? (((r - a.chunk_off[ck]) & 1) == 0) : true;
		Pairs v;
		Work w;
		v.num = sp.num;
		for (int j = 0; j < sp.num; ++j) {
			AlnSpillPair q = sp.p[j];
			v.gPos[j] = q.gPos; v.rPos[j] = q.rPos; v.rLen[j] = q.rLen; v.gLen[j] = q.gLen; v.simple[j] = q.kind == W_SIMPLE;
			w.kind[j] = q.kind; w.op[j] = q.op; w.op_len[j] = q.op_len; w.val[j] = q.val;
		}
		if (!finish_candidate(a, cand, first, a.enc + a.read_off[r], v, w)) flag_host(a, r, WHY_CIGAR);
	}
}

// ---- pass 2 by the wave ----------------------------------------------------------------------------------------------------------
// aln_finish_kernel above gives a candidate to a lane: the lane walks the columns of each of its alignments one after the other, two to four
// times (CheckLocalAlignmentQuality, the trimming of the head / tail, AddNewCigarElements), 64 candidates of different shapes per wave in lock
// step -- a launch lasted as long as its slowest wave, 0.8 ms for 150 k candidates (profiles/r06h).  Here the WAVE takes a candidate and its
// lanes take the columns: lane c of tile k looks at column 64 k + c -- the op, the read's character and the text's, found through prefix counts
// of the ops in front of it -- and three ballots turn the tile into bit sets: the read side shows '-', the text side shows '-', both show the
// same character.  Everything the reference's loops derive from the columns is bit arithmetic on those sets (uniform, a few instructions per
// run instead of tens per column); the normal pairs of the candidate live one per lane, the CIGAR elements one per lane.  Same records.
namespace {

__device__ __forceinline__ int rl32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int64_t rl64(int64_t v, int lane)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, lane);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), lane);
	return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
	return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint64_t bits_below(int n) { return n >= 64 ? ~0ull : n <= 0 ? 0ull : ((1ull << n) - 1); }

// the columns of one alignment as bit sets; tile k (columns 64 k .. 64 k + 63) is held by lane k
struct WaveCols {
	uint64_t m1, m2, eq;
	int len;
	__device__ __forceinline__ uint64_t valid(int k) const { return bits_below(len - 64 * k); }
	__device__ __forceinline__ uint64_t M1(int k) const { return (uint64_t)rl64((int64_t)m1, k) & valid(k); }       // the read side shows '-' (a gap, or a literal '-' of the read)
	__device__ __forceinline__ uint64_t M2(int k) const { return (uint64_t)rl64((int64_t)m2, k) & valid(k); }       // the text side shows '-'
	__device__ __forceinline__ uint64_t EQ(int k) const { return (uint64_t)rl64((int64_t)eq, k) & valid(k); }       // both sides show the same character
};

__device__ __forceinline__ void wave_columns(const AlnArgs &a, const uint8_t *ops, int len, const uint8_t *rd, int64_t g, int lane, WaveCols &wc)
{
	wc.m1 = 0; wc.m2 = 0; wc.eq = 0; wc.len = len;
	int ri = 0, gi = 0;
	const uint64_t lt = bits_below(lane);
	for (int k = 0; k * 64 < len; ++k) {
		const int col = k * 64 + lane;
		const bool valid = col < len;
		const uint8_t op = valid ? ops[col] : (uint8_t)KG_OP_DIAG;
		const bool n1 = valid && op != KG_OP_GAP1, n2 = valid && op != KG_OP_GAP2;          // the column consumes a read / a text character
		const uint64_t b1 = __ballot(n1), b2 = __ballot(n2);
		char c1 = '-', c2 = '-';
		if (n1) c1 = (char)rd[ri + __popcll(b1 & lt)];
		if (n2) c2 = text_char(a, g + gi + __popcll(b2 & lt));
		const uint64_t M1 = __ballot(valid && c1 == '-'), M2 = __ballot(valid && c2 == '-'), EQ = __ballot(valid && c1 == c2);
		if (lane == k) { wc.m1 = M1; wc.m2 = M2; wc.eq = EQ; }
		ri += __popcll(b1); gi += __popcll(b2);
	}
}

struct WaveCigar {                  // element i in lane i
	int my_len, my_op;
	int n;
	bool overflow;
	__device__ __forceinline__ void push(int l, char o, int lane)
	{
		if (n < kAlnMaxCigar) { if (lane == n) { my_len = l; my_op = (int)o; } n++; }
		else overflow = true;
	}
};

// AddNewCigarElements over columns [from, to), src/tools.cpp:49-104: a run per class -- 'D' where the read side shows '-', else 'I' where the text side
// does, else 'M' -- and the number of identical characters among the 'M' columns
__device__ int wave_add_cigar(const WaveCols &wc, int from, int to, int lane, WaveCigar &cig)
{
	int score = 0;
	for (int k = from >> 6; k * 64 < to; ++k) {
		const uint64_t range = ~bits_below(from - 64 * k) & bits_below(to - 64 * k);
		score += __popcll(wc.EQ(k) & ~wc.M1(k) & ~wc.M2(k) & range);
	}
	char state = '*';
	int cnt = 0, col = from;
	while (col < to) {
		int k = col >> 6;
		const int bit = col & 63;
		uint64_t m1 = wc.M1(k), b = ~m1 & wc.M2(k);
		const bool cur1 = ((m1 >> bit) & 1) != 0, curb = ((b >> bit) & 1) != 0;
		const char st = cur1 ? 'D' : curb ? 'I' : 'M';
		const uint64_t f1 = cur1 ? ~0ull : 0ull, fb = curb ? ~0ull : 0ull;
		uint64_t x = ((m1 ^ f1) | (b ^ fb)) & ~bits_below(bit + 1);            // columns behind col, in its tile, of another class
		int end = to;
		for (;;) {
			if (x) { end = k * 64 + (int)__builtin_ctzll(x); break; }
			++k;
			if (k * 64 >= to) break;
			m1 = wc.M1(k); b = ~m1 & wc.M2(k);
			x = (m1 ^ f1) | (b ^ fb);
		}
		if (end > to) end = to;
		const int run = end - col;
		if (st == state) cnt += run;
		else {
			if (cnt > 0) cig.push(cnt, state, lane);
			cnt = run;
			state = st;
		}
		col = end;
	}
	if (cnt > 0) cig.push(cnt, state, lane);
	return score;
}

// CheckLocalAlignmentQuality, src/tools.cpp:255-290
__device__ bool wave_quality_ok(const WaveCols &wc)
{
	int n = 0, mis = 0, runs = 0;
	uint64_t c1 = 0, cb = 0;
	for (int k = 0; k * 64 < wc.len; ++k) {
		const uint64_t v = wc.valid(k), m1 = wc.M1(k), m2 = wc.M2(k), b = ~m1 & m2, t2 = v & ~m1 & ~m2;
		n += __popcll(t2);
		mis += __popcll(t2 & ~wc.EQ(k));
		uint64_t chg = ((m1 ^ ((m1 << 1) | c1)) | (b ^ ((b << 1) | cb))) & v;
		if (k == 0) chg |= 1;                                            // (the first column always opens a run)
		runs += __popcll(chg);
		c1 = m1 >> 63; cb = b >> 63;
	}
	return !(runs >= 4 || (mis >= 3 && mis >= (int)(n * 0.3)));
}

// columns from `start` on whose bit is set in the set whose tile this lane holds in `mine`
__device__ int wave_lead_run(const WaveCols &wc, uint64_t mine, int start)
{
	int c = start;
	while (c < wc.len) {
		const int k = c >> 6, bit = c & 63;
		const uint64_t x = ~((uint64_t)rl64((int64_t)mine, k) & wc.valid(k)) & ~bits_below(bit);
		if (x) { c = k * 64 + (int)__builtin_ctzll(x); break; }
		c = (k + 1) * 64;
	}
	if (c > wc.len) c = wc.len;
	return c - start;
}
// ... and the columns in front of `end`, backwards
__device__ int wave_trail_run(const WaveCols &wc, uint64_t mine, int end)
{
	int c = end;
	while (c > 0) {
		const int k = (c - 1) >> 6, bit = (c - 1) & 63;
		const uint64_t x = ~((uint64_t)rl64((int64_t)mine, k) & wc.valid(k)) & bits_below(bit + 1);
		if (x) { c = k * 64 + 64 - (int)__builtin_clzll(x); break; }
		c = k * 64;
	}
	return end - c;
}

// finish_candidate for the candidate of spill slot t, by the wave (every lane of it is here)
__device__ void finish_candidate_wave(const AlnArgs &a, unsigned long long t, int lane)
{
	const AlnSpill &sp = a.spill[t];
	const int64_t cand = (int64_t)uni(sp.cand);
	const int64_t r = (int64_t)uni((int)a.c_read[cand]);
	if (uni((int)a.r_host[r])) return;
	const int ck = uni(chunk_of(a, r));
	const bool first = uni((int)a.chunk_paired[ck]) ? (((r - a.chunk_off[ck]) & 1) == 0) : true;
	const int num = uni(sp.num);
	const uint8_t *rd = a.enc + a.read_off[r];
	// pair j in lane j
	int64_t gPos = 0;
	int rPos = 0, rLen = 0, gLen = 0, val = 0, op_len = 0, kind = W_NONE, op = 0;
	if (lane < num) {
		const AlnSpillPair q = sp.p[lane];
		gPos = q.gPos; rPos = q.rPos; rLen = q.rLen; gLen = q.gLen; val = q.val; op_len = q.op_len; kind = q.kind; op = q.op;
	}
	WaveCigar cig;
	cig.my_len = 0; cig.my_op = 0; cig.n = 0; cig.overflow = false;
	int score = 0;
	for (int j = 0; j < num; ++j) {
		const int kj = rl32(kind, j);
		if (kj == W_NONE) continue;
		const int rLj = rl32(rLen, j);
		if (kj == W_SIMPLE) {
			cig.push(rLj, 'M', lane);
			score += rLj;
			continue;
		}
		const bool head = j == 0, tail = j == num - 1 && !head;
		int s;
		if (kj == W_IMMEDIATE) {
			const int oj = rl32(op, j);
			if (oj != 0) cig.push(rl32(op_len, j), (char)oj, lane);
			s = rl32(val, j);
		} else {
			const int vj = rl32(val, j);
			const uint8_t *ops;
			int len;
			if (kj == W_INLINE) {
				ops = reinterpret_cast<const uint8_t *>(a.rep_cigar + cand * KG_ALN_CIGAR_MAX) + 16 * (vj & 255);
				len = vj >> 8;
			} else if (kj == W_JOB) {
				ops = a.nw_ops + uni64(a.jobs[vj].ops);
				len = uni(a.nw_len[vj]);
			} else {
				// the partitioned fragment: literal runs and the sub-fragments' op strings laid one behind the other (src/tools.cpp:165-208), 64 bytes per step
				const AlnPlan pl = a.plans[vj];
				uint8_t *out = a.nw_ops + uni64(pl.ops);
				const int first_piece = uni(pl.first), n_pieces = uni(pl.count);
				int at = 0;
				for (int k = 0; k < n_pieces; ++k) {
					const AlnPiece pc = a.pieces[first_piece + k];
					const int pk = uni((int)pc.kind), pv = uni(pc.v);
					if (pk <= KG_OP_GAP2) {
						for (int i = lane; i < pv; i += 64) out[at + i] = (uint8_t)pk;
						at += pv > 0 ? pv : 0;
					} else {
						const uint8_t *src = a.nw_ops + uni64(a.jobs[pv].ops);
						const int L = uni(a.nw_len[pv]);
						for (int i = lane; i < L; i += 64) out[at + i] = src[i];
						at += L > 0 ? L : 0;
					}
				}
				wave_sync_mem();
				ops = out;
				len = at;
			}
			WaveCols wc;
			wave_columns(a, ops, len, rd + rl32(rPos, j), rl64(gPos, j), lane, wc);
			if (head) {
				// ProcessHeadSequencePair after the alignment, src/tools.cpp:314-339
				if (!wave_quality_ok(wc)) { cig.push(rLj, 'S', lane); s = 0; }
				else {
					const int p = wave_lead_run(wc, wc.m1, 0);
					const int p2 = wave_lead_run(wc, wc.m2, p);
					if (lane == j) {
						if (p > 0) { gPos += p; gLen -= p; }
						if (p2 > 0) { rPos += p2; rLen -= p2; }
					}
					if (p2 > 0) cig.push(p2, 'S', lane);
					s = wave_add_cigar(wc, p + p2, len, lane, cig);
				}
			} else if (tail) {
				// ProcessTailSequencePair after the alignment, src/tools.cpp:366-394
				if (!wave_quality_ok(wc)) { cig.push(rLj, 'S', lane); s = 0; }
				else {
					const int cnt = wave_trail_run(wc, wc.m1, len);
					const int cnt2 = wave_trail_run(wc, wc.m2, len - cnt);
					if (lane == j) {
						if (cnt > 0) gLen -= cnt;
						if (cnt2 > 0) rLen -= cnt2;
					}
					s = wave_add_cigar(wc, 0, len - cnt - cnt2, lane, cig);
					if (cnt2 > 0) cig.push(cnt2, 'S', lane);
				}
			} else s = wave_add_cigar(wc, 0, len, lane, cig);
		}
		if (head) {
			if (s > 0) score += s;
			if (s <= 0) { const int64_t g1 = rl64(gPos, 1); if (lane == 0) { gPos = g1; gLen = 0; } }         // :674-686
		} else if (tail) {
			if (s > 0) score += s;
			if (s <= 0) { const int64_t gp = rl64(gPos, j - 1) + rl32(gLen, j - 1); if (lane == j) { gPos = gp; gLen = 0; } }
		} else score += s;
	}
	if (cig.overflow) { if (lane == 0) flag_host(a, r, WHY_CIGAR); return; }
	int rep_chr = 0, rep_fwd = 1, rep_len = 0;
	int64_t rep_pos = 0;
	bool scored = true;                                                     // false: "continue" before the best / second-best step (c_score -1)
	bool fits = true;
	if (cig.n > 1) {                                                       // GapPenalty, :612-622, :701-706
		const int gp = wave_sum(lane < cig.n && (cig.my_op == 'I' || cig.my_op == 'D') ? cig.my_len : 0);
		score -= gp;
		if (score <= 0) { score = 0; scored = false; }
	}
	if (scored) {
		if (cig.n == 0) score = 0;
		else {
			// GenCoordinateInfo, :515-562
			const int64_t gPos0 = rl64(gPos, 0), end_gPos = rl64(gPos, num - 1) + rl32(gLen, num - 1) - 1;
			bool fwd, rev = false;
			int chr;
			int64_t pos;
			if (gPos0 < a.genome_size) {
				fwd = first;
				if (a.n_chr == 1) { chr = 0; pos = gPos0 + 1; }
				else {
					const int it = uni(end_lower_bound(a, gPos0));
					chr = uni(a.end_chr[it]);
					pos = gPos0 + 1 - a.chr_fwd_start[chr];
				}
			} else {
				fwd = !first;
				rev = true;
				if (a.n_chr == 1) { chr = 0; pos = a.two_genome_size - end_gPos; }
				else {
					int it = uni(end_lower_bound(a, gPos0));
					if (it == a.n_ends) it = a.n_ends - 1;
					pos = a.contig_end[it] - end_gPos + 1;
					chr = uni(a.end_chr[it]);
				}
			}
			// GenerateCIGAR, :492-513 (the reverse strand shows the elements in reverse order)
			char *out = a.rep_cigar + cand * KG_ALN_CIGAR_MAX;
			int at = 0, cnt = 0, state = 0;
			auto emit = [&](int nn, int st) {
				int nd = 1;
				for (int x = nn; x >= 10; x /= 10) nd++;
				if (at + nd + 1 > KG_ALN_CIGAR_MAX - 1) { fits = false; return; }
				for (int d = nd - 1; d >= 0; --d) { if (lane == 0) out[at + d] = (char)('0' + nn % 10); nn /= 10; }
				if (lane == 0) out[at + nd] = (char)st;
				at += nd + 1;
			};
			for (int q = 0; q < cig.n; ++q) {
				const int i = rev ? cig.n - 1 - q : q;
				const int l = rl32(cig.my_len, i), o = rl32(cig.my_op, i);
				if (o != state) {
					if (cnt > 0) emit(cnt, state);
					cnt = l;
					state = o;
				} else cnt += l;
			}
			if (cnt > 0) emit(cnt, state);
			if (fits) {
				rep_len = at; rep_chr = chr; rep_pos = pos; rep_fwd = fwd ? 1 : 0;
				if (pos <= 0) score = 0;
			}
		}
	}
	if (lane == 0) {
		if (!fits) {
			// (the lane form leaves the candidate's report fields zeroed and hands the read to the host)
			a.rep_chr[cand] = 0; a.rep_pos[cand] = 0; a.rep_fwd[cand] = 1; a.rep_cigar_len[cand] = 0;
			flag_host(a, r, WHY_CIGAR);
		} else {
			a.rep_chr[cand] = rep_chr;
			a.rep_pos[cand] = rep_pos;
			a.rep_fwd[cand] = (uint8_t)rep_fwd;
			a.rep_cigar_len[cand] = (uint8_t)rep_len;
			a.rep_score[cand] = score;
			if (!scored) a.c_score[cand] = -1;
		}
	}
}

}  // namespace

__global__ __launch_bounds__(256) void aln_finish_wave_kernel(AlnArgs a)
{
	unsigned long long n = a.ctl[0];
	if (n > (unsigned long long)a.spill_capacity) n = (unsigned long long)a.spill_capacity;
	const int lane = threadIdx.x & 63;
	const unsigned long long n_waves = (unsigned long long)gridDim.x * (blockDim.x >> 6);
	for (unsigned long long t = (unsigned long long)uni((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))); t < n; t += n_waves) finish_candidate_wave(a, t, lane);
}

// ---- pass 2 by groups of eight (or sixteen) lanes ---------------------------------------------------------------------------------------------
// A candidate per wave (above) spends a whole wave's issue slot on every instruction of what is mostly bit arithmetic common to the wave:
// ~1700 instructions per candidate, 150 k candidates per launch, 0.48 ms -- bound by instruction issue.  Here kFinG lanes take a candidate,
// 64 / kFinG candidates share a wave's instructions: the same scheme on tiles of kFinG columns, with the candidate's normal pairs, CIGAR elements
// and column sets in the LDS (1 KB per group).  The groups of a wave run apart where their candidates differ (different numbers of pairs,
// different kinds of pairs); every ballot and shuffle stays inside one group, whose lanes always run together.
namespace {

// (kFinG lanes per candidate, a template parameter: eight is the product's -- 26.8 ms per step against 30.5 with sixteen, profiles/r06q --, sixteen an A/B form: KG_ALN_FINISH_G16)
template <int kFinG>
struct FinShared {
	static constexpr int kFinTiles = (2 * kAlnMaxFrag + kFinG - 1) / kFinG;
	AlnSpillPair p[kAlnMaxPairs];
	int32_t cig_len[kAlnMaxCigar];
	uint8_t cig_op[kAlnMaxCigar + 4];
	uint16_t m1[kFinTiles], m2[kFinTiles], eq[kFinTiles];
};

template <int kFinG>
struct Fin {
	FinShared<kFinG> *sh;
	int gl, gb;            // lane in the group, the group's first lane in the wave
	int len;               // columns of the alignment at hand
	int cig_n;
	bool cig_overflow;
	__device__ __forceinline__ uint32_t ballot(bool p) const { return (uint32_t)((__ballot(p) >> gb) & ((1ull << kFinG) - 1)); }
	static __device__ __forceinline__ uint32_t below(int n) { return n >= kFinG ? ((1u << kFinG) - 1) : n <= 0 ? 0u : ((1u << n) - 1); }
	__device__ __forceinline__ uint32_t valid(int k) const { return below(len - kFinG * k); }
	__device__ __forceinline__ uint32_t M1(int k) const { return (uint32_t)sh->m1[k]; }
	__device__ __forceinline__ uint32_t M2(int k) const { return (uint32_t)sh->m2[k]; }
	__device__ __forceinline__ uint32_t EQ(int k) const { return (uint32_t)sh->eq[k]; }
	__device__ __forceinline__ void push(int l, char o)
	{
		if (cig_n < kAlnMaxCigar) { if (gl == 0) { sh->cig_len[cig_n] = l; sh->cig_op[cig_n] = (uint8_t)o; } cig_n++; }
		else cig_overflow = true;
	}
	// the columns of the alignment as bit sets, a tile of kFinG columns per entry (wave_columns)
	__device__ __forceinline__ void columns(const AlnArgs &a, const uint8_t *ops, int n_cols, const uint8_t *rd, int64_t g)
	{
		len = n_cols;
		int ri = 0, gi = 0;
		const uint32_t lt = below(gl);
		for (int k = 0; k * kFinG < n_cols; ++k) {
			const int col = k * kFinG + gl;
			const bool ok = col < n_cols;
			const uint8_t op = ok ? ops[col] : (uint8_t)KG_OP_DIAG;
			const bool n1 = ok && op != KG_OP_GAP1, n2 = ok && op != KG_OP_GAP2;
			const uint32_t b1 = ballot(n1), b2 = ballot(n2);
			char c1 = '-', c2 = '-';
			if (n1) c1 = (char)rd[ri + __popc(b1 & lt)];
			if (n2) c2 = text_char(a, g + gi + __popc(b2 & lt));
			const uint32_t m1 = ballot(ok && c1 == '-'), m2 = ballot(ok && c2 == '-'), eq = ballot(ok && c1 == c2);
			if (gl == 0) { sh->m1[k] = (uint16_t)m1; sh->m2[k] = (uint16_t)m2; sh->eq[k] = (uint16_t)eq; }
			ri += __popc(b1); gi += __popc(b2);
		}
	}
	// AddNewCigarElements over columns [from, to), src/tools.cpp:49-104 (wave_add_cigar)
	__device__ int add_cigar(int from, int to)
	{
		constexpr uint32_t all = (1u << kFinG) - 1;
		int score = 0;
		for (int k = from / kFinG; k * kFinG < to; ++k) {
			const uint32_t range = ~below(from - kFinG * k) & below(to - kFinG * k);
			score += __popc(EQ(k) & ~M1(k) & ~M2(k) & range);
		}
		char state = '*';
		int cnt = 0, col = from;
		while (col < to) {
			int k = col / kFinG;
			const int bit = col - k * kFinG;
			uint32_t m1 = M1(k), b = ~m1 & M2(k);
			const bool cur1 = ((m1 >> bit) & 1) != 0, curb = ((b >> bit) & 1) != 0;
			const char st = cur1 ? 'D' : curb ? 'I' : 'M';
			const uint32_t f1 = cur1 ? all : 0u, fb = curb ? all : 0u;
			uint32_t x = ((m1 ^ f1) | (b ^ fb)) & ~below(bit + 1) & all;
			int end = to;
			for (;;) {
				if (x) { end = k * kFinG + (__ffs((int)x) - 1); break; }
				++k;
				if (k * kFinG >= to) break;
				m1 = M1(k); b = ~m1 & M2(k);
				x = ((m1 ^ f1) | (b ^ fb)) & all;
			}
			if (end > to) end = to;
			const int run = end - col;
			if (st == state) cnt += run;
			else {
				if (cnt > 0) push(cnt, state);
				cnt = run;
				state = st;
			}
			col = end;
		}
		if (cnt > 0) push(cnt, state);
		return score;
	}
	// CheckLocalAlignmentQuality, src/tools.cpp:255-290
	__device__ bool quality_ok() const
	{
		int n = 0, mis = 0, runs = 0;
		uint32_t c1 = 0, cb = 0;
		for (int k = 0; k * kFinG < len; ++k) {
			const uint32_t v = valid(k), m1 = M1(k), m2 = M2(k), b = ~m1 & m2, t2 = v & ~m1 & ~m2;
			n += __popc(t2);
			mis += __popc(t2 & ~EQ(k));
			uint32_t chg = ((m1 ^ ((m1 << 1) | c1)) | (b ^ ((b << 1) | cb))) & v;
			if (k == 0) chg |= 1;
			runs += __popc(chg);
			c1 = (m1 >> (kFinG - 1)) & 1; cb = (b >> (kFinG - 1)) & 1;
		}
		return !(runs >= 4 || (mis >= 3 && mis >= (int)(n * 0.3)));
	}
	__device__ int lead_run(const uint16_t *set, int start) const
	{
		constexpr uint32_t all = (1u << kFinG) - 1;
		int c = start;
		while (c < len) {
			const int k = c / kFinG, bit = c - k * kFinG;
			const uint32_t x = ~((uint32_t)set[k] & valid(k)) & ~below(bit) & all;
			if (x) { c = k * kFinG + (__ffs((int)x) - 1); break; }
			c = (k + 1) * kFinG;
		}
		if (c > len) c = len;
		return c - start;
	}
	__device__ int trail_run(const uint16_t *set, int end) const
	{
		int c = end;
		while (c > 0) {
			const int k = (c - 1) / kFinG, bit = (c - 1) - k * kFinG;
			const uint32_t x = ~((uint32_t)set[k] & valid(k)) & below(bit + 1);
			if (x) { c = k * kFinG + 32 - __clz((int)x); break; }
			c = k * kFinG;
		}
		return end - c;
	}
};

// GapPenalty, GenCoordinateInfo, GenerateCIGAR and the candidate's report fields (src/AlignmentCandidates.cpp:612-622, 492-562, 701-722) by a group of
// lanes that all hold the same values (lane 0 of it stores): cn CIGAR elements, the first pair's gPos, the last pair's last text coordinate
__device__ void report_by_group(const AlnArgs &a, int64_t cand, int64_t r, bool first, int gl, int score, int cn, const int32_t *cig_len, const uint8_t *cig_op,
                                int64_t gPos0, int64_t end_gPos)
{
	int rep_chr = 0, rep_fwd = 1, rep_len = 0;
	int64_t rep_pos = 0;
	bool scored = true, fits = true;
	if (cn > 1) {                                                          // GapPenalty, :612-622, :701-706
		int gp = 0;
		for (int i = 0; i < cn; ++i) { const int o = cig_op[i]; if (o == 'I' || o == 'D') gp += cig_len[i]; }
		score -= gp;
		if (score <= 0) { score = 0; scored = false; }
	}
	if (scored) {
		if (cn == 0) score = 0;
		else {
			// GenCoordinateInfo, :515-562
			bool fwd, rev = false;
			int chr;
			int64_t pos;
			if (gPos0 < a.genome_size) {
				fwd = first;
				if (a.n_chr == 1) { chr = 0; pos = gPos0 + 1; }
				else {
					const int it = end_lower_bound(a, gPos0);
					chr = a.end_chr[it];
					pos = gPos0 + 1 - a.chr_fwd_start[chr];
				}
			} else {
				fwd = !first;
				rev = true;
				if (a.n_chr == 1) { chr = 0; pos = a.two_genome_size - end_gPos; }
				else {
					int it = end_lower_bound(a, gPos0);
					if (it == a.n_ends) it = a.n_ends - 1;
					pos = a.contig_end[it] - end_gPos + 1;
					chr = a.end_chr[it];
				}
			}
			// GenerateCIGAR, :492-513 (the reverse strand shows the elements in reverse order)
			char *out = a.rep_cigar + cand * KG_ALN_CIGAR_MAX;
			int at = 0, cnt = 0, state = 0;
			auto emit = [&](int nn, int st) {
				int nd = 1;
				for (int x = nn; x >= 10; x /= 10) nd++;
				if (at + nd + 1 > KG_ALN_CIGAR_MAX - 1) { fits = false; return; }
				for (int d = nd - 1; d >= 0; --d) { if (gl == 0) out[at + d] = (char)('0' + nn % 10); nn /= 10; }
				if (gl == 0) out[at + nd] = (char)st;
				at += nd + 1;
			};
			for (int q = 0; q < cn; ++q) {
				const int i = rev ? cn - 1 - q : q;
				const int l = cig_len[i], o = cig_op[i];
				if (o != state) {
					if (cnt > 0) emit(cnt, state);
					cnt = l;
					state = o;
				} else cnt += l;
			}
			if (cnt > 0) emit(cnt, state);
			if (fits) {
				rep_len = at; rep_chr = chr; rep_pos = pos; rep_fwd = fwd ? 1 : 0;
				if (pos <= 0) score = 0;
			}
		}
	}
	if (gl == 0) {
		if (!fits) {
			a.rep_chr[cand] = 0; a.rep_pos[cand] = 0; a.rep_fwd[cand] = 1; a.rep_cigar_len[cand] = 0;
			flag_host(a, r, WHY_CIGAR);
		} else {
			a.rep_chr[cand] = rep_chr;
			a.rep_pos[cand] = rep_pos;
			a.rep_fwd[cand] = (uint8_t)rep_fwd;
			a.rep_cigar_len[cand] = (uint8_t)rep_len;
			a.rep_score[cand] = score;
			if (!scored) a.c_score[cand] = -1;
		}
	}
}

// finish_candidate for the candidate of spill slot t, by a group of kFinG lanes (all of them here)
template <int kFinG>
__device__ void finish_candidate_group(const AlnArgs &a, unsigned long long t, Fin<kFinG> &fi)
{
	const AlnSpill &sp = a.spill[t];
	const int64_t cand = sp.cand;
	const int64_t r = a.c_read[cand];
	if (a.r_host[r]) return;
	const int ck = chunk_of(a, r);
	const bool first = a.chunk_paired[ck] ? (((r - a.chunk_off[ck]) & 1) == 0) : true;
	const int num = sp.num;
	const uint8_t *rd = a.enc + a.read_off[r];
	FinShared<kFinG> *sh = fi.sh;
	const int gl = fi.gl;
	{
		const uint32_t *src = reinterpret_cast<const uint32_t *>(sp.p);
		uint32_t *dst = reinterpret_cast<uint32_t *>(sh->p);
		const int words = num * (int)(sizeof(AlnSpillPair) / 4);
		for (int i = gl; i < words; i += kFinG) dst[i] = src[i];
	}
	fi.cig_n = 0; fi.cig_overflow = false;
	int score = 0;
	for (int j = 0; j < num; ++j) {
		const int kj = sh->p[j].kind;
		if (kj == W_NONE) continue;
		const int rLj = sh->p[j].rLen;
		if (kj == W_SIMPLE) {
			fi.push(rLj, 'M');
			score += rLj;
			continue;
		}
		const bool head = j == 0, tail = j == num - 1 && !head;
		int s;
		if (kj == W_IMMEDIATE) {
			const int oj = sh->p[j].op;
			if (oj != 0) fi.push(sh->p[j].op_len, (char)oj);
			s = sh->p[j].val;
		} else {
			const int vj = sh->p[j].val;
			const uint8_t *ops;
			int len;
			if (kj == W_INLINE) {
				ops = reinterpret_cast<const uint8_t *>(a.rep_cigar + cand * KG_ALN_CIGAR_MAX) + 16 * (vj & 255);
				len = vj >> 8;
			} else if (kj == W_JOB) {
				ops = a.nw_ops + a.jobs[vj].ops;
				len = a.nw_len[vj];
			} else {
				// the partitioned fragment: literal runs and the sub-fragments' op strings laid one behind the other (src/tools.cpp:165-208)
				const AlnPlan pl = a.plans[vj];
				uint8_t *out = a.nw_ops + pl.ops;
				int at = 0;
				for (int k = 0; k < pl.count; ++k) {
					const AlnPiece pc = a.pieces[pl.first + k];
					if (pc.kind <= KG_OP_GAP2) {
						for (int i = gl; i < pc.v; i += kFinG) out[at + i] = pc.kind;
						at += pc.v > 0 ? pc.v : 0;
					} else {
						const uint8_t *src = a.nw_ops + a.jobs[pc.v].ops;
						const int L = a.nw_len[pc.v];
						for (int i = gl; i < L; i += kFinG) out[at + i] = src[i];
						at += L > 0 ? L : 0;
					}
				}
				wave_sync_mem();
				ops = out;
				len = at;
			}
			fi.columns(a, ops, len, rd + sh->p[j].rPos, sh->p[j].gPos);
			if (head) {
				// ProcessHeadSequencePair after the alignment, src/tools.cpp:314-339
				if (!fi.quality_ok()) { fi.push(rLj, 'S'); s = 0; }
				else {
					const int p = fi.lead_run(sh->m1, 0);
					const int p2 = fi.lead_run(sh->m2, p);
					if (gl == 0) {
						if (p > 0) { sh->p[j].gPos += p; sh->p[j].gLen = (int16_t)(sh->p[j].gLen - p); }
						if (p2 > 0) { sh->p[j].rPos += p2; sh->p[j].rLen = (int16_t)(sh->p[j].rLen - p2); }
					}
					if (p2 > 0) fi.push(p2, 'S');
					s = fi.add_cigar(p + p2, len);
				}
			} else if (tail) {
				// ProcessTailSequencePair after the alignment, src/tools.cpp:366-394
				if (!fi.quality_ok()) { fi.push(rLj, 'S'); s = 0; }
				else {
					const int cnt = fi.trail_run(sh->m1, len);
					const int cnt2 = fi.trail_run(sh->m2, len - cnt);
					if (gl == 0) {
						if (cnt > 0) sh->p[j].gLen = (int16_t)(sh->p[j].gLen - cnt);
						if (cnt2 > 0) sh->p[j].rLen = (int16_t)(sh->p[j].rLen - cnt2);
					}
					s = fi.add_cigar(0, len - cnt - cnt2);
					if (cnt2 > 0) fi.push(cnt2, 'S');
				}
			} else s = fi.add_cigar(0, len);
		}
		if (head) {
			if (s > 0) score += s;
			if (s <= 0) { const int64_t g1 = sh->p[1].gPos; if (gl == 0) { sh->p[0].gPos = g1; sh->p[0].gLen = 0; } }         // :674-686
		} else if (tail) {
			if (s > 0) score += s;
			if (s <= 0) { const int64_t gp = sh->p[j - 1].gPos + sh->p[j - 1].gLen; if (gl == 0) { sh->p[j].gPos = gp; sh->p[j].gLen = 0; } }
		} else score += s;
	}
	if (fi.cig_overflow) { if (gl == 0) flag_host(a, r, WHY_CIGAR); return; }
	report_by_group(a, cand, r, first, gl, score, fi.cig_n, sh->cig_len, sh->cig_op, sh->p[0].gPos, num > 0 ? sh->p[num - 1].gPos + sh->p[num - 1].gLen - 1 : 0);
}

}  // namespace

template <int kFinG>
__global__ __launch_bounds__(256) void aln_finish_group_kernel(AlnArgs a)
{
	constexpr int kFinGroups = 256 / kFinG;
	__shared__ FinShared<kFinG> s_fin[kFinGroups];
	unsigned long long n = a.ctl[0];
	if (n > (unsigned long long)a.spill_capacity) n = (unsigned long long)a.spill_capacity;
	Fin<kFinG> fi;
	fi.gl = threadIdx.x & (kFinG - 1);
	fi.gb = (threadIdx.x & 63) & ~(kFinG - 1);
	fi.sh = &s_fin[threadIdx.x / kFinG];
	fi.len = 0; fi.cig_n = 0; fi.cig_overflow = false;
	const unsigned long long n_groups = (unsigned long long)gridDim.x * kFinGroups;
	for (unsigned long long t = (unsigned long long)blockIdx.x * kFinGroups + threadIdx.x / kFinG; t < n; t += n_groups) finish_candidate_group(a, t, fi);
}

// ---- pass 1 by groups of eight lanes -----------------------------------------------------------------------------------------------
// aln_plan_kernel gives a candidate to a lane, and the lane keeps the candidate's normal pairs and what it decides about them in 1344 B of
// private (scratch) memory (profiles/r06h: waiting 0.78 of its cycles).  The experiment: eight lanes take a candidate: its pairs live in the LDS (1 KB per group) --
// IdentifyNormalPairs and CheckCoordinateValidity are the same code (templates over the storage), run by the eight lanes alike --, the pairs
// are classified eight at a time (the mismatch count of each is a chain of its own), and a candidate that needs no alignment is reported by
// the group (report_by_group).  What is parked is written exactly as aln_plan_kernel writes it: the same spill slots, jobs and tasks.
// (The alignments of aln_plan_kernel's own lanes, KG_ALN_INLINE, stay that kernel's.)
// MEASURED: correct (CHECK_ALIGN 0 of 5.2 M, profiles/r06s) and slower -- 86 ms per step against aln_plan_kernel's 31: IdentifyNormalPairs is serial per
// candidate, and a wave that runs it for 8 candidates instead of 64 issues eight times the instructions.  An A/B form (KG_ALN_PLAN_GROUP), off.
namespace {

constexpr int kPlanG = 8;
struct PlanShared {
	int64_t gPos[kAlnMaxPairs];
	int32_t rPos[kAlnMaxPairs], rLen[kAlnMaxPairs], gLen[kAlnMaxPairs];
	int32_t op_len[kAlnMaxPairs], val[kAlnMaxPairs];
	int32_t cig_len[kAlnMaxCigar];
	uint8_t simple[kAlnMaxPairs + 2], kind[kAlnMaxPairs + 2], op[kAlnMaxPairs + 2];
	uint8_t cig_op[kAlnMaxCigar + 4];
};

__device__ __forceinline__ int group_sum(int v) { for (int off = kPlanG / 2; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; }

}  // namespace

__global__ __launch_bounds__(256) void aln_plan_group_kernel(AlnArgs a)
{
	constexpr int kGroups = 256 / kPlanG;
	__shared__ PlanShared s_plan[kGroups];
	PlanShared *const sh = &s_plan[threadIdx.x / kPlanG];
	const int gl = threadIdx.x & (kPlanG - 1);
	const int lead = (threadIdx.x & 63) & ~(kPlanG - 1);                       // the group's first lane in the wave
	const int64_t n_groups = (int64_t)gridDim.x * kGroups;
	int64_t n_all = plan_slots(a);
	if (a.plan_slow) n_all = (int64_t)a.ctl[32];
	// (the wave's groups stay together through the loop: the list entries of all of them are reserved by one atomic per list)
	for (int64_t slot0 = ((int64_t)blockIdx.x * kGroups + (threadIdx.x >> 6) * (64 / kPlanG)); slot0 < n_all; slot0 += n_groups) {
		const int64_t slot = slot0 + ((threadIdx.x & 63) / kPlanG);
		int64_t cand = 0, r = 0, rbase = 0;
		int num = 0, n_new_jobs = 0, new_ops = 0, n_pending = 0;
		bool park = false;
		do {
		if (slot >= n_all) break;
		cand = a.plan_slow ? (int64_t)a.plan_slow[slot] : a.plan_order ? (int64_t)a.plan_order[slot] : slot_cand(a, slot);
		if (gl == 0) { a.rep_score[cand] = 0; a.rep_chr[cand] = 0; a.rep_pos[cand] = 0; a.rep_fwd[cand] = 1; a.rep_cigar_len[cand] = 0; }
		r = a.c_read[cand];
		if (a.r_host[r]) break;
		if (a.c_score[cand] == 0) break;                                  // GenMappingReport skips it, :643
		const bool rescued = cand >= a.n_cands;
		int count;
		const kg_seed *seeds;
		if (!rescued) { const kg_candidate cd = a.cands[cand]; count = cd.count; seeds = a.cand_seeds + cd.first; }
		else {
			const int64_t t = cand - a.n_cands;
			count = a.resc_count[t];
			seeds = a.resc_seeds + t * kAlnMaxSeeds;
		}
		if (count > kAlnMaxSeeds) { if (gl == 0) flag_host(a, r, WHY_SEEDS); break; }
		rbase = a.read_off[r];
		const int rlen = (int)(a.read_off[r + 1] - rbase);
		const uint8_t *rd = a.enc + rbase;
		const int ck = chunk_of(a, r);
		const bool first = a.chunk_paired[ck] ? (((r - a.chunk_off[ck]) & 1) == 0) : true;
		for (int i = gl; i < count; i += kPlanG) {
			const kg_seed sd = seeds[i];
			sh->gPos[i] = sd.gPos; sh->rPos[i] = sd.rPos; sh->rLen[i] = sd.len; sh->gLen[i] = sd.len; sh->simple[i] = 1;
		}
		if (rlen > 4000) { if (gl == 0) flag_host(a, r, WHY_READ_LEN); break; }
		PairsRef v;
		v.gPos = sh->gPos; v.rPos = sh->rPos; v.rLen = sh->rLen; v.gLen = sh->gLen; v.simple = sh->simple; v.num = count;
		if (!identify_normal_pairs(rlen, -1, v)) { if (gl == 0) flag_host(a, r, WHY_GAPS); break; }
		if (!coordinates_valid(a, v)) { if (gl == 0) a.c_score[cand] = -1; break; }      // no report, and no best/second-best step (:647)
		num = v.num;
		// ---- the pairs, kPlanG at a time: what aln_plan_kernel's loop decides for pair j ----
		int host_l = 0, jobs_l = 0;
		for (int j = gl; j < num; j += kPlanG) {
			int kind = W_NONE, op = 0, op_len = 0, val = 0;
			const int rL = sh->rLen[j], gL = sh->gLen[j];
			do {
				if (rL == 0 && gL == 0) break;
				if (sh->simple[j]) { kind = W_SIMPLE; break; }
				const int role = j == 0 ? 0 : j == num - 1 ? 2 : 1;
				if (role != 1 && rL > 3000) { kind = W_IMMEDIATE; op = 'S'; op_len = rL; val = -1; break; }               // :671-676, :690-695
				if (role == 1 && (rL == 0 || gL == 0)) {                                                                    // ProcessNormalSequencePair, src/tools.cpp:229-233
					kind = W_IMMEDIATE;
					if (rL > 0) { op = 'I'; op_len = rL; }
					else if (gL > 0) { op = 'D'; op_len = gL; }
					break;
				}
				const uint8_t *f1 = rd + sh->rPos[j];
				const int64_t g = sh->gPos[j];
				if (rL == gL) {                                                     // the <= 2-mismatch shortcut, :240, :301, :352
					bool dash_ = false;
					const int n = fast_gap_mismatches(a, f1, g, rL, 3, dash_);
					if (n <= 2 && n <= (int)(rL * 0.2)) { kind = W_IMMEDIATE; op = 'M'; op_len = rL; val = rL - n; break; }
				}
				if ((role == 0 && rL > 50) || (role == 2 && rL > 100)) { kind = W_IMMEDIATE; op = 'S'; op_len = rL; val = 0; break; }   // :307-311, :358-362
				if (rL == 1 && gL == 1 && f1[0] != '-') { kind = W_IMMEDIATE; op = 'M'; op_len = 1; val = (char)f1[0] == text_char(a, g) ? 1 : 0; break; }
				if (rL > kAlnMaxFrag || gL > kAlnMaxFrag || rL <= 0 || gL <= 0) { host_l = 1; break; }
				if (rL > 30 && gL > 30) {
					if (a.dbg_no_partition) { host_l = 1; break; }
					kind = W_PENDING; jobs_l = 1; n_pending++;
					break;
				}
				kind = W_JOB; val = -1;
				n_new_jobs++; new_ops += rL + gL;
				jobs_l = 1;
			} while (false);
			sh->kind[j] = (uint8_t)kind; sh->op[j] = (uint8_t)op; sh->op_len[j] = op_len; sh->val[j] = val;
		}
		const bool host = group_sum(host_l) != 0, jobs = group_sum(jobs_l) != 0;
		n_new_jobs = group_sum(n_new_jobs); new_ops = group_sum(new_ops); n_pending = group_sum(n_pending);
		if (host) { if (gl == 0) flag_host(a, r, WHY_PARTITION); break; }
		if (!jobs) {
			// no alignment needed: GenMappingReport's pair loop over simple pairs and elements known at once (finish_candidate), then the report
			int cn = 0, score = 0;
			bool overflow = false;
			auto push = [&](int l, int o) {
				if (cn < kAlnMaxCigar) { if (gl == 0) { sh->cig_len[cn] = l; sh->cig_op[cn] = (uint8_t)o; } cn++; }
				else overflow = true;
			};
			for (int j = 0; j < num; ++j) {
				const int kj = sh->kind[j];
				if (kj == W_NONE) continue;
				if (kj == W_SIMPLE) { push(sh->rLen[j], 'M'); score += sh->rLen[j]; continue; }
				const bool head = j == 0, tail = j == num - 1 && !head;
				if (sh->op[j] != 0) push(sh->op_len[j], sh->op[j]);
				const int sj = sh->val[j];
				if (head) {
					if (sj > 0) score += sj;
					if (sj <= 0) { const int64_t g1 = sh->gPos[1]; if (gl == 0) { sh->gPos[0] = g1; sh->gLen[0] = 0; } }         // :674-686
				} else if (tail) {
					if (sj > 0) score += sj;
					if (sj <= 0) { const int64_t gp = sh->gPos[j - 1] + sh->gLen[j - 1]; if (gl == 0) { sh->gPos[j] = gp; sh->gLen[j] = 0; } }
				} else score += sj;
			}
			if (overflow) { if (gl == 0) flag_host(a, r, WHY_CIGAR); break; }
			report_by_group(a, cand, r, first, gl, score, cn, sh->cig_len, sh->cig_op, sh->gPos[0], num > 0 ? sh->gPos[num - 1] + sh->gLen[num - 1] - 1 : 0);
			break;
		}
		park = true;
		} while (false);
		// what the wave's parked candidates need of the lists, reserved with ONE atomic per list (the first lane of a group asks for its candidate)
		const bool ask = park && gl == 0;
		unsigned long long sp = wave_reserve(&a.ctl[0], ask ? 1ull : 0ull);
		unsigned long long job_at = wave_reserve(&a.ctl[1], ask ? (unsigned long long)n_new_jobs : 0ull);
		unsigned long long ops_at = wave_reserve(&a.ctl[2], ask ? (unsigned long long)new_ops : 0ull);
		unsigned long long task_at = wave_reserve(&a.ctl[3], ask ? (unsigned long long)n_pending : 0ull);
		auto from_lead = [&](unsigned long long x) { return ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(x >> 32), lead) << 32) | (uint32_t)__shfl((int)(uint32_t)x, lead); };
		sp = from_lead(sp); job_at = from_lead(job_at); ops_at = from_lead(ops_at); task_at = from_lead(task_at);
		if (!park) continue;
		const bool sp_ok = sp < (unsigned long long)a.spill_capacity;
		const bool jobs_ok = job_at + (unsigned long long)n_new_jobs <= (unsigned long long)a.job_capacity && ops_at + (unsigned long long)new_ops <= (unsigned long long)a.ops_capacity;
		if (!sp_ok || !jobs_ok) {
			// a list is full: the read is the host's; everything taken INSIDE the lists is still written (see aln_plan_kernel)
			if (gl == 0) {
				if (sp_ok) { a.spill[sp].cand = (int32_t)cand; a.spill[sp].num = 0; }
				for (unsigned long long k = job_at; k < job_at + (unsigned long long)n_new_jobs && k < (unsigned long long)a.job_capacity; ++k) { NwJobDesc jd; jd.o1 = 0; jd.o2 = 0; jd.ops = 0; jd.m = 0; jd.n = 0; a.jobs[k] = jd; }
				flag_host(a, r, WHY_CAPACITY);
				for (unsigned long long k = task_at; k < task_at + (unsigned long long)n_pending && k < (unsigned long long)a.job_capacity; ++k) {
					PartTask pt;
					pt.enc_off = rbase; pt.g = 0; pt.spill = 0; pt.j = 0; pt.read = (int32_t)r; pt.rL = 0; pt.gL = 0;
					a.part_tasks[k] = pt;
				}
			}
			continue;
		}
		// the jobs in pair order (their places in the list follow from the pairs before them), the tasks likewise
		if (gl == 0) {
			for (int j = 0; j < num; ++j) {
				if (sh->kind[j] != W_JOB) continue;
				const int rL = sh->rLen[j], gL = sh->gLen[j];
				NwJobDesc jd;
				jd.o1 = rbase + sh->rPos[j]; jd.o2 = sh->gPos[j]; jd.ops = (int64_t)ops_at; jd.m = rL; jd.n = gL;
				a.jobs[job_at] = jd;
				sh->val[j] = (int32_t)job_at;
				job_at++; ops_at += (unsigned long long)(rL + gL);
			}
			if (n_pending > 0) {
				if (task_at + (unsigned long long)n_pending > (unsigned long long)a.job_capacity) flag_host(a, r, WHY_CAPACITY);
				for (int j = 0; j < num; ++j) {
					if (sh->kind[j] != W_PENDING) continue;
					const unsigned long long t = task_at++;
					if (t >= (unsigned long long)a.job_capacity) break;
					PartTask pt;
					pt.enc_off = rbase + sh->rPos[j]; pt.g = sh->gPos[j]; pt.spill = (int32_t)sp; pt.j = j; pt.read = (int32_t)r;
					pt.rL = (int16_t)sh->rLen[j]; pt.gL = (int16_t)sh->gLen[j];
					a.part_tasks[t] = pt;
				}
			}
		}
		AlnSpill &o = a.spill[sp];
		if (gl == 0) { o.cand = (int32_t)cand; o.num = num; }
		for (int j = gl; j < num; j += kPlanG) {
			AlnSpillPair q;
			q.gPos = sh->gPos[j]; q.rPos = sh->rPos[j]; q.rLen = (int16_t)sh->rLen[j]; q.gLen = (int16_t)sh->gLen[j];
			q.val = sh->val[j]; q.kind = sh->kind[j]; q.op = sh->op[j]; q.op_len = (int16_t)sh->op_len[j];
			o.p[j] = q;
		}
	}
}

// ---- per read: best / second best, final pair check, flags, MAPQ, records ----------------------------------------------------
namespace {

struct ReadSum {                    // the ReadItem_t fields the output depends on
	int score, sub_score, best, can_num, mapq, rlen;
	CandList l;
};

// the tail of GenMappingReport's loop, src/AlignmentCandidates.cpp:724-740 (bMultiHit false)
__device__ void summarise(const AlnArgs &a, int64_t r, ReadSum &s)
{
	s.l = cand_list(a, r);
	s.can_num = s.l.n();
	s.rlen = (int)(a.read_off[r + 1] - a.read_off[r]);
	s.score = s.sub_score = s.best = 0;
	s.mapq = 0;
	for (int i = 0; i < s.can_num; ++i) {
		int cs = a.c_score[s.l.at(i)];
		if (cs == 0 || cs == -1) continue;                          // skipped before the comparison (Score == 0, invalid coordinates, gap penalty)
		int sc = a.rep_score[s.l.at(i)];
		if (sc > s.score) { s.best = i; s.sub_score = s.score; s.score = sc; }
		else if (sc == s.score) {
			s.sub_score = s.score;
			if (!a.multi_hit && a.chr_len[a.rep_chr[s.l.at(i)]] > a.chr_len[a.rep_chr[s.l.at(s.best)]]) s.best = i;
		}
	}
}

__device__ __forceinline__ int eval_mapq(const AlnArgs &a, const ReadSum &s)   // EvaluateMAPQ, src/Mapping.cpp:160-175
{
	if (s.score == 0 || s.score == s.sub_score) return 0;
	int q;
	const int d = s.score - s.sub_score;
	if (s.sub_score == 0 || d > 5) q = 60;
	else if (d > 0) q = a.mapq_tab[s.score * 6 + d];
	// score < sub_score happens (CheckPairedFinalAlignments can settle on a mated candidate below the read's second best): the
	// expression then exceeds 60 for every score >= 8 (30 ln 8 = 62.4); the few smaller scores are tabulated as well
	else if (s.score >= 8) q = 60;
	else q = a.mapq_tab[(kAlnMaxScore + 1) * 6 + s.score * (kAlnMaxScore + 1) + (-d)];
	return q > 60 ? 60 : q;
}

// rep[i] of a read; a read without candidates holds one empty report (score 0, mate -1, forward; :627-634)
__device__ __forceinline__ int rep_score_at(const AlnArgs &a, const ReadSum &s, int i) { return (s.can_num == 0) ? 0 : a.rep_score[s.l.at(i)]; }
__device__ __forceinline__ int rep_mate_at(const AlnArgs &a, const ReadSum &s, int i) { return (s.can_num == 0) ? -1 : a.c_mate[s.l.at(i)]; }
__device__ __forceinline__ bool rep_fwd_at(const AlnArgs &a, const ReadSum &s, int i) { return (s.can_num == 0) ? true : a.rep_fwd[s.l.at(i)] != 0; }

// the per-mate halves of SetPairedAlignmentFlag, src/Mapping.cpp:96-156: the flag of the record that can be printed for `me`
// (its best candidate), or of the unmapped record
__device__ int one_mate_flag(const AlnArgs &a, const ReadSum &me, const ReadSum &other, int base)
{
	if (me.score > 0) {                                            // (score > sub_score and score == sub_score > 0 set the best candidate's flag alike)
		if (me.score <= me.sub_score && rep_score_at(a, me, me.best) <= 0) return 0;   // not assigned; such a record is never printed
		int f = base | (rep_fwd_at(a, me, me.best) ? 0x20 : 0x10);
		int j = rep_mate_at(a, me, me.best);
		if (j != -1 && rep_score_at(a, other, j) > 0) f |= 0x2;
		else f |= 0x8;
		return f;
	}
	int f = base | 0x4;
	if (other.score == 0) f |= 0x8;
	else f |= (rep_fwd_at(a, other, other.best) ? 0x10 : 0x20);
	return f;
}

// record slot `at` (the read's own slot, or one of the extra slots of -m) for candidate `cand_i` of the read
__device__ void write_record_at(const AlnArgs &a, int64_t at, const ReadSum &s, int cand_i, int kind, int flag, bool has_mate, int64_t mate_pos, int tlen, bool flip)
{
	kg_aln_record &o = a.records[at];
	o.kind = kind; o.flag = flag; o.mapq = s.mapq; o.score = s.score; o.sub_score = s.sub_score;
	o.has_mate = has_mate ? 1 : 0; o.mate_pos = mate_pos; o.tlen = tlen; o.flip = flip ? 1 : 0;      // (est_lo / est_hi / rescue: aln_pair_kernel)
	o.chr = -1; o.pos = 0; o.cigar_len = 0;
	o.next = -1; o.primary = cand_i == s.best ? 1 : 0; o.pad[0] = o.pad[1] = o.pad[2] = 0;
	if (kind == KG_ALN_MAPPED) {
		int64_t c = s.l.at(cand_i);
		o.chr = a.rep_chr[c]; o.pos = a.rep_pos[c];
		int n = a.rep_cigar_len[c];
		o.cigar_len = (uint8_t)n;
		// (eight characters per load and store: both sides are 8-byte aligned and KG_ALN_CIGAR_MAX long; what lies behind cigar_len is nobody's)
		const uint64_t *src = reinterpret_cast<const uint64_t *>(a.rep_cigar + c * KG_ALN_CIGAR_MAX);
		uint64_t *dst = reinterpret_cast<uint64_t *>(o.cigar);
		for (int i = 0; 8 * i < n; ++i) dst[i] = src[i];
	}
}

__device__ __forceinline__ void write_record(const AlnArgs &a, int64_t r, const ReadSum &s, int kind, int flag, bool has_mate, int64_t mate_pos, int tlen, bool flip)
{
	write_record_at(a, r, s, s.best, kind, flag, has_mate, mate_pos, tlen, flip);
}

// -m: the next record of read r goes into its own slot when that is still free, else into an extra slot chained behind `last`
// (the slot written before).  false: the extra slots are used up.
__device__ bool next_slot(const AlnArgs &a, int64_t r, int64_t &last, int64_t &at)
{
	if (last < 0) { at = r; last = r; return true; }
	unsigned long long k = atomicAdd(&a.ctl[7], 1ull);
	if (k >= (unsigned long long)a.extra_capacity) return false;
	at = a.n_reads + (int64_t)k;
	a.records[last].next = (int32_t)at;
	last = at;
	return true;
}

// SetPairedAlignmentFlag for candidate i of `me` when the run prints more than the best candidate (-m): assigned exactly where the
// reference assigns it (src/Mapping.cpp:78-93, 96-156), the unset value elsewhere
__device__ int multi_flag(const AlnArgs &a, const ReadSum &me, const ReadSum &other, int i, int base, bool both_unique, int best_flag)
{
	if (both_unique || me.score > me.sub_score) return i == me.best ? best_flag : a.unset_flag;
	// me.score == me.sub_score > 0: every candidate with a positive score is assigned
	int f = base | (rep_fwd_at(a, me, i) ? 0x20 : 0x10);
	int j = rep_mate_at(a, me, i);
	if (j != -1 && rep_score_at(a, other, j) > 0) f |= 0x2; else f |= 0x8;
	return f;
}

}  // namespace

__global__ __launch_bounds__(256) void aln_final_kernel(AlnArgs a)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	const int64_t n_units = a.slow_pairs ? (int64_t)a.ctl[35] : a.n_reads;          // (the pairs aln_trivial_kernel left, or every read)
	// (the wave's lanes stay together: what they add to their chunk's statistics is summed across the wave -- 64 lanes, mostly one chunk, one address)
	for (int64_t x0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); x0 < n_units; x0 += stride) {
		const int64_t x = x0 + (threadIdx.x & 63);
		int ck = -1;
		long long add_paired = 0, add_dist = 0;
		int add_unmapped = 0, add_unique = 0, add_host = 0;
		do {
		if (x >= n_units) break;
		const int64_t r = a.slow_pairs ? (int64_t)a.slow_pairs[x] << 1 : x;
		const int ck_r = chunk_of(a, r);
		const bool paired = a.chunk_paired[ck_r] != 0;
		if (paired && ((r - a.chunk_off[ck_r]) & 1)) break;
		ck = ck_r;
		if (a.r_host[r]) {
			a.records[r].kind = KG_ALN_HOST;
			if (paired) a.records[r + 1].kind = KG_ALN_HOST;
			add_host = paired ? 2 : 1;
			break;
		}
		ReadSum s1;
		summarise(a, r, s1);
		if (s1.score > kAlnMaxScore || s1.sub_score > kAlnMaxScore) {             // beyond the MAPQ table (reads longer than 2047 bases)
			atomicAdd(&a.ctl[8 + WHY_SCORE], 1ull);
			a.records[r].kind = KG_ALN_HOST;
			if (paired) a.records[r + 1].kind = KG_ALN_HOST;
			break;
		}
		if (!paired) {
			// SetSingleAlignmentFlag + EvaluateMAPQ + OutputSingledAlignments, src/Mapping.cpp:49-71, 160-175, 272-315
			s1.mapq = eval_mapq(a, s1);
			if (s1.score == 0) {
				add_unmapped = 1;
				write_record(a, r, s1, KG_ALN_UNMAPPED, 0x4, false, 0, 0, false);
			} else {
				// the candidates from `best` on whose score is the read's: the first one, or with -m all of them (:291-304); every
				// one of them carries an assigned flag (a second candidate of the read's score makes score == sub_score, :58-66)
				int64_t last = -1, at = r;
				bool full = false;
				for (int i = s1.best; i < s1.can_num && !full; ++i) {
					if (a.rep_score[s1.l.at(i)] != s1.score) continue;
					if (!next_slot(a, r, last, at)) { full = true; break; }
					bool fwd = a.rep_fwd[s1.l.at(i)] != 0;
					write_record_at(a, at, s1, i, KG_ALN_MAPPED, fwd ? 0 : 0x10, false, 0, 0, !fwd);
					if (!a.multi_hit) break;
				}
				if (full) { atomicAdd(&a.ctl[8 + WHY_CAPACITY], 1ull); a.records[r].kind = KG_ALN_HOST; break; }
				if (last < 0) write_record(a, r, s1, KG_ALN_NONE, 0, false, 0, 0, false);
				if (s1.mapq == 60) add_unique = 1;
			}
			break;
		}
		ReadSum s2;
		summarise(a, r + 1, s2);
		if (s2.score > kAlnMaxScore || s2.sub_score > kAlnMaxScore) {
			atomicAdd(&a.ctl[8 + WHY_SCORE], 1ull);
			a.records[r].kind = KG_ALN_HOST; a.records[r + 1].kind = KG_ALN_HOST;
			break;
		}
		// CheckPairedFinalAlignments, src/Mapping.cpp:429-480 (bMultiHit false)
		{
			bool mated = false;
			if (s1.can_num > 0 && s2.can_num > 0) mated = a.c_mate[s1.l.at(s1.best)] == s2.best;
			else if (s1.can_num == 0 && s2.can_num > 0) mated = -1 == s2.best;       // (a report of an empty read: mate -1)
			else if (s1.can_num > 0 && s2.can_num == 0) mated = a.c_mate[s1.l.at(s1.best)] == 0;
			else mated = false;                                                      // -1 == 0
			if (!mated || a.multi_hit) {                                             // (!bMultiHit && bMated returns, :438)
				if (!mated && s1.score > 0 && s2.score > 0) {
					int s = 0;
					for (int i = 0; i < s1.can_num; ++i) {
						int j;
						if (a.rep_score[s1.l.at(i)] > 0 && (j = a.c_mate[s1.l.at(i)]) != -1 && a.rep_score[s2.l.at(j)] > 0) {
							mated = true;
							int t = a.rep_score[s1.l.at(i)] + a.rep_score[s2.l.at(j)];
							if (s < t) {
								s = t;
								s1.best = i; s1.score = a.rep_score[s1.l.at(i)];
								s2.best = j; s2.score = a.rep_score[s2.l.at(j)];
							}
						}
					}
				}
				if (mated) {
					for (int i = 0; i < s1.can_num; ++i) {
						int j;
						if (a.rep_score[s1.l.at(i)] != s1.score || ((j = a.c_mate[s1.l.at(i)]) != -1 && a.rep_score[s2.l.at(j)] != s2.score)) {
							a.rep_score[s1.l.at(i)] = 0;
							a.c_mate[s1.l.at(i)] = -1;
						}
					}
				} else {
					for (int i = 0; i < s1.can_num; ++i) {
						a.c_mate[s1.l.at(i)] = -1;
						if (a.rep_score[s1.l.at(i)] > 0 && a.rep_score[s1.l.at(i)] != s1.score) a.rep_score[s1.l.at(i)] = 0;
					}
					for (int j = 0; j < s2.can_num; ++j) {
						a.c_mate[s2.l.at(j)] = -1;
						if (a.rep_score[s2.l.at(j)] > 0 && a.rep_score[s2.l.at(j)] != s2.score) a.rep_score[s2.l.at(j)] = 0;
					}
				}
			}
		}
		// SetPairedAlignmentFlag, src/Mapping.cpp:73-158
		int f1, f2;
		if (s1.score > s1.sub_score && s2.score > s2.sub_score) {
			f1 = 0x41; f2 = 0x81;
			if (s2.best == rep_mate_at(a, s1, s1.best)) { f1 |= 0x2; f2 |= 0x2; }
			f1 |= rep_fwd_at(a, s1, s1.best) ? 0x20 : 0x10;
			f2 |= rep_fwd_at(a, s2, s2.best) ? 0x20 : 0x10;
		} else {
			f1 = one_mate_flag(a, s1, s2, 0x41);
			f2 = one_mate_flag(a, s2, s1, 0x81);
		}
		s1.mapq = eval_mapq(a, s1);
		s2.mapq = eval_mapq(a, s2);
		// OutputPairedAlignments, src/Mapping.cpp:177-270: the best candidate, or with -m every candidate from the best one on that
		// still has a positive score
		const bool both_unique = s1.score > s1.sub_score && s2.score > s2.sub_score;
		bool full = false;
		if (s1.score == 0) {
			add_unmapped++;
			write_record(a, r, s1, KG_ALN_UNMAPPED, f1, false, 0, 0, false);
		} else {
			if (s1.mapq == 60) add_unique++;
			int64_t last = -1, at = r;
			for (int i = s1.best; i < s1.can_num && !full; ++i) {
				if (rep_score_at(a, s1, i) > 0) {
					if (!next_slot(a, r, last, at)) { full = true; break; }
					int fl = a.multi_hit ? multi_flag(a, s1, s2, i, 0x41, both_unique, f1) : f1;
					int j = rep_mate_at(a, s1, i);
					bool fwd = rep_fwd_at(a, s1, i);
					if (j != -1 && rep_score_at(a, s2, j) > 0) {
						int dist = (int)(a.rep_pos[s2.l.at(j)] - a.rep_pos[s1.l.at(i)] + (fwd ? s2.rlen : 0 - s1.rlen));
						if (i == s1.best) {
							add_paired = 2;
							int ad = dist < 0 ? -dist : dist;
							if (ad < 10000) add_dist = ad;
						}
						write_record_at(a, at, s1, i, KG_ALN_MAPPED, fl, true, a.rep_pos[s2.l.at(j)], dist, !fwd);
					} else write_record_at(a, at, s1, i, KG_ALN_MAPPED, fl, false, 0, 0, !fwd);
				}
				if (!a.multi_hit) break;
			}
			if (last < 0) write_record(a, r, s1, KG_ALN_NONE, 0, false, 0, 0, false);
		}
		if (s2.score == 0) {
			add_unmapped++;
			write_record(a, r + 1, s2, KG_ALN_UNMAPPED, f2, false, 0, 0, false);
		} else {
			if (s2.mapq == 60) add_unique++;
			int64_t last = -1, at = r + 1;
			for (int j = s2.best; j < s2.can_num && !full; ++j) {
				if (rep_score_at(a, s2, j) > 0) {
					if (!next_slot(a, r + 1, last, at)) { full = true; break; }
					int fl = a.multi_hit ? multi_flag(a, s2, s1, j, 0x81, both_unique, f2) : f2;
					int i = rep_mate_at(a, s2, j);
					bool fwd = rep_fwd_at(a, s2, j);
					if (i != -1 && rep_score_at(a, s1, i) > 0) {
						bool fwd1 = rep_fwd_at(a, s1, i);
						int dist = 0 - (int)(a.rep_pos[s2.l.at(j)] - a.rep_pos[s1.l.at(i)] + (fwd1 ? s2.rlen : 0 - s1.rlen));
						write_record_at(a, at, s2, j, KG_ALN_MAPPED, fl, true, a.rep_pos[s1.l.at(i)], dist, fwd);
					} else write_record_at(a, at, s2, j, KG_ALN_MAPPED, fl, false, 0, 0, fwd);
				}
				if (!a.multi_hit) break;
			}
			if (last < 0) write_record(a, r + 1, s2, KG_ALN_NONE, 0, false, 0, 0, false);
		}
		if (full) {                                    // no extra record slot left: the pair goes to the host
			atomicAdd(&a.ctl[8 + WHY_CAPACITY], 1ull);
			a.records[r].kind = KG_ALN_HOST; a.records[r + 1].kind = KG_ALN_HOST;
			add_unmapped = 0; add_unique = 0; add_paired = 0; add_dist = 0;      // (nothing of a pair handed back is counted here)
			break;
		}
		} while (false);
		// ---- into the chunks' statistics: one set of atomics per wave where its lanes share a chunk ----
		const uint64_t have = __ballot(ck >= 0);
		if (have == 0) continue;
		const int ck0 = __shfl(ck, __ffsll((unsigned long long)have) - 1);
		if (__ballot(ck >= 0 && ck != ck0) == 0) {
			for (int off = 32; off > 0; off >>= 1) {
				add_paired += __shfl_xor(add_paired, off); add_dist += __shfl_xor(add_dist, off);
				add_unmapped += __shfl_xor(add_unmapped, off); add_unique += __shfl_xor(add_unique, off); add_host += __shfl_xor(add_host, off);
			}
			if ((threadIdx.x & 63) != 0) ck = -1;
		}
		if (ck >= 0) {
			kg_chunk_stats &cs = a.chunk_stats[ck];
			if (add_host) atomicAdd(&cs.host_pairs, add_host);
			if (add_unmapped) atomicAdd(&cs.unmapped, add_unmapped);
			if (add_unique) atomicAdd(&cs.unique, add_unique);
			if (add_paired) atomicAdd((unsigned long long *)&cs.paired, (unsigned long long)add_paired);
			if (add_dist) atomicAdd((unsigned long long *)&cs.distance, (unsigned long long)add_dist);
		}
	}
}


// The order in which aln_plan_kernel takes the candidates: four bins by the number of seeds (none or one / two / three / more), each
// block a contiguous range of candidates -- pass 0 counts the bins (ctl[24..27]), pass 1 places the indices (ctl[28..31] run along).
__device__ __forceinline__ int plan_bin(const AlnArgs &a, int64_t cand)
{
	const int64_t r = a.c_read[cand];
	if (a.r_host[r] || a.c_score[cand] == 0) return 0;
	const int n = cand < a.n_cands ? a.cands[cand].count : a.resc_count[cand - a.n_cands];
	return n <= 1 ? 0 : n == 2 ? 1 : n == 3 ? 2 : 3;
}

__global__ __launch_bounds__(256) void aln_bin_kernel(AlnArgs a, int pass)
{
	__shared__ unsigned int s_cnt[4];
	__shared__ unsigned long long s_next[4];
	const int64_t n_all = plan_slots(a);
	const int64_t per = (n_all + gridDim.x - 1) / gridDim.x;
	const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n_all ? b0 + per : n_all;
	if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	unsigned int mine[4] = {0, 0, 0, 0};
	for (int64_t c = b0 + threadIdx.x; c < b1; c += blockDim.x) mine[plan_bin(a, slot_cand(a, c))]++;
#pragma unroll
	for (int b = 0; b < 4; ++b) {
		unsigned int v = mine[b];
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
		if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt[b], v);
	}
	__syncthreads();
	if (pass == 0) {
		if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&a.ctl[24 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
		return;
	}
	if (threadIdx.x < 4) {
		unsigned long long base = 0;
		for (int b = 0; b < (int)threadIdx.x; ++b) base += a.ctl[24 + b];
		s_next[threadIdx.x] = base + (s_cnt[threadIdx.x] ? atomicAdd(&a.ctl[28 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]) : 0ull);
	}
	__syncthreads();
	for (int64_t base = b0; base < b1; base += blockDim.x) {
		const int64_t c = base + threadIdx.x < b1 ? slot_cand(a, base + threadIdx.x) : -1;
		const int bin = c >= 0 ? plan_bin(a, c) : -1;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const uint64_t mask = __ballot(bin == b);
			if (mask == 0) continue;
			const int leader = __ffsll((unsigned long long)mask) - 1;
			unsigned long long at = 0;
			if ((int)(threadIdx.x & 63) == leader) at = atomicAdd(&s_next[b], (unsigned long long)__popcll(mask));
			at = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(at >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)at, leader);
			const uint64_t below = (threadIdx.x & 63) == 0 ? 0ull : (~0ull >> (64 - (threadIdx.x & 63)));
			if (bin == b) a.plan_order[at + (unsigned long long)__popcll(mask & below)] = (int32_t)c;
		}
	}
}

__global__ void aln_reset_kernel(AlnArgs a)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	// (ctl[8..23]: running tallies, never reset: [8..20] why pairs went back to the host; [21..23] parked candidates, NW jobs and
	// partition plans of the batches before this one)
	if (i == 0) { a.ctl[21] += a.ctl[0]; a.ctl[22] += a.ctl[1]; a.ctl[23] += a.ctl[5]; }
	__syncthreads();
	if (i == 0) a.ctl[33] += a.ctl[32];                  // (running tally: candidates the fast plan kernel left to the general one)
	if (i < 8) a.ctl[i] = 0;
	if (i >= 24 && i < 33) a.ctl[i] = 0;
	if (i >= 34 && i <= 37) a.ctl[i] = 0;                // (aln_trivial_kernel: candidates / pairs it leaves to the general kernels, pairs it decided)
	for (int c = i; c < a.n_chunks; c += gridDim.x * blockDim.x) {
		kg_chunk_stats z;
		z.paired = 0; z.distance = 0; z.lo = -1; z.hi = 0x7fffffffffffffffll; z.unmapped = 0; z.unique = 0; z.host_pairs = 0; z.rescue_wanted = 0;
		a.chunk_stats[c] = z;
	}
	for (int64_t r = i; r < a.n_reads; r += (int64_t)gridDim.x * blockDim.x) { a.r_host[r] = 0; a.r_pending[r] = 0; a.resc_n[r] = 0; a.resc_off[r] = 0; }
}

static inline int grid_for_aln(int64_t items, int block, int max_blocks)
{
	int64_t g = (items + block - 1) / block;
	if (g < 1) g = 1;
	if (g > max_blocks) g = max_blocks;
	return (int)g;
}

hipError_t launch_align_front(const AlnArgs &a, int n_cu, hipStream_t stream)
{
	hipLaunchKernelGGL(aln_reset_kernel, dim3(grid_for_aln(a.n_reads, 256, n_cu * 8)), dim3(256), 0, stream, a);
	if (a.slow_pairs) {
		kt_begin(KT_ALN_TRIVIAL, stream);
		hipLaunchKernelGGL(aln_trivial_kernel, dim3(grid_for_aln(a.n_reads / 2 + 1, 256, n_cu * 16)), dim3(256), 0, stream, a);
		kt_end(KT_ALN_TRIVIAL, stream);
	}
	kt_begin(KT_ALN_PAIR, stream);
	hipLaunchKernelGGL(aln_pair_kernel, dim3(grid_for_aln(a.all_paired ? a.n_reads / 2 + 1 : a.n_reads, 256, n_cu * 16)), dim3(256), 0, stream, a);
	kt_end(KT_ALN_PAIR, stream);
	kt_begin(KT_ALN_RESCUE, stream);
	hipLaunchKernelGGL(aln_rescue_kernel, dim3(grid_for_aln(a.task_capacity, 1, n_cu * 32)), dim3(64), 0, stream, a);
	hipLaunchKernelGGL(aln_post_rescue_kernel, dim3(grid_for_aln(a.n_reads, 256, n_cu * 16)), dim3(256), 0, stream, a);
	kt_end(KT_ALN_RESCUE, stream);
	if (a.n_cands > 0) {
		if (a.plan_order) {
			hipLaunchKernelGGL(aln_bin_kernel, dim3(grid_for_aln(a.n_cands + a.task_capacity / 8, 256, n_cu * 8)), dim3(256), 0, stream, a, 0);
			hipLaunchKernelGGL(aln_bin_kernel, dim3(grid_for_aln(a.n_cands + a.task_capacity / 8, 256, n_cu * 8)), dim3(256), 0, stream, a, 1);
		}
		kt_begin(KT_ALN_PLAN_FAST, stream);
		if (a.plan_slow) hipLaunchKernelGGL(aln_plan_fast_kernel, dim3(grid_for_aln(a.n_cands + a.task_capacity / 8, 256, n_cu * 16)), dim3(256), 0, stream, a);
		kt_end(KT_ALN_PLAN_FAST, stream);
		kt_begin(KT_ALN_PLAN, stream);
		if (a.dbg_plan_group && a.dbg_no_inline) hipLaunchKernelGGL(aln_plan_group_kernel, dim3(n_cu * 16), dim3(256), 0, stream, a);
		else hipLaunchKernelGGL(aln_plan_kernel, dim3(grid_for_aln(a.n_cands + a.task_capacity / 8, 256, n_cu * 16)), dim3(256), 0, stream, a);
		kt_end(KT_ALN_PLAN, stream);
		kt_begin(KT_ALN_PARTITION, stream);
		hipLaunchKernelGGL(aln_partition_kernel, dim3(grid_for_aln(a.n_cands / 8 + 1, 256, n_cu * 8)), dim3(256), 0, stream, a);
		kt_end(KT_ALN_PARTITION, stream);
	}
	return hipGetLastError();
}

hipError_t launch_align_back(const AlnArgs &a, int n_cu, hipStream_t stream)
{
	kt_begin(KT_ALN_FINISH, stream);
	if (a.dbg_finish_lanes == 1) hipLaunchKernelGGL(aln_finish_kernel, dim3(grid_for_aln(a.spill_capacity, 256, n_cu * 8)), dim3(256), 0, stream, a);
	else if (a.dbg_finish_lanes == 2) hipLaunchKernelGGL(aln_finish_wave_kernel, dim3(n_cu * 8), dim3(256), 0, stream, a);
	else if (a.dbg_finish_lanes == 3) hipLaunchKernelGGL(aln_finish_group_kernel<16>, dim3(n_cu * 16), dim3(256), 0, stream, a);
	else hipLaunchKernelGGL(aln_finish_group_kernel<8>, dim3(n_cu * 16), dim3(256), 0, stream, a);
	kt_end(KT_ALN_FINISH, stream);
	kt_begin(KT_ALN_FINAL, stream);
	hipLaunchKernelGGL(aln_final_kernel, dim3(grid_for_aln(a.n_reads, 256, n_cu * 16)), dim3(256), 0, stream, a);
	kt_end(KT_ALN_FINAL, stream);
	return hipGetLastError();
}

}  // namespace kg
