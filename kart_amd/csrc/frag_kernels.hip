// frag_kernels.hip -- GenerateNormalPairAlignment for long fragments on the device (gfx950): the -pacbio branch.
//
// For a read fragment and a genome fragment that both exceed 30 bases the reference (src/tools.cpp:142-223) finds their common
// 8-mers within a shift limit -- min(50, 20 % of the longer side) for -pacbio, MaxGaps otherwise -- merges them into exact matches
// (GenerateSimplePairsFromFragmentPair, src/KmerAnalysis.cpp:104-179), runs IdentifyNormalPairs(rLen, gLen, ...) on those
// (src/AlignmentCandidates.cpp:420-490 with its three seed filters, :235-418), and aligns what lies between them: literal
// stretches, nw_alignment for the sub-fragments, and -- -pacbio only -- the same procedure again for a sub-fragment with a side
// above 300 (the recursion at :197).  aln_partition_kernel (align_kernels.hip) does this for short-read fragments (<= 255
// bases, <= 12 matches, no recursion); this file is the general form:
//   frag_partition_kernel  one wave per task (a fragment pair).  The read fragment and the text window go into the LDS as 2-bit
//                          codes; every lane scans diagonals for runs of >= 8 equal bases (32 bases per step); the runs are
//                          rank-sorted by (gPos, rPos); lane 0 runs the reference's IdentifyNormalPairs on the LDS arrays (the
//                          filters are sequential by nature) and writes the task's pieces: literal columns, NW jobs (for the
//                          NW kernels, nw_kernels.hip), sub-tasks for the next level.
//   frag_stitch_kernel     one lane per request: the op string of the whole fragment from its pieces, depth first (what the
//                          reference leaves in frag1 / frag2 at the end of GenerateNormalPairAlignment).
// Integer / bit work, no MFMA.  Outside the envelope (a read character other than A/C/G/T, more than kFragMaxRuns matches,
// fragments beyond kFragMaxLen, recursion deeper than kFragMaxDepth) the request is handed back (status 1) and the caller
// plans it with its own implementation of the same reference code.
#include "frag_kernels.hpp"

namespace kg {

namespace {

struct __attribute__((packed, aligned(1))) FrU64u { uint64_t v; };

__device__ __forceinline__ uint64_t text_word32_at(const uint8_t *text, int64_t two_l, int64_t p)     // 32 bases of the 2-bit text from position p
{
	if (p < 0 || p > two_l) return 0;
	const uint8_t *tp = text + ((uint64_t)p >> 2);
	uint64_t lo = reinterpret_cast<const FrU64u *>(tp)->v, hi = tp[8];
	int sh = ((int)p & 3) << 1;
	return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

// 32 read characters as four 8-byte words (the uploaded characters have 64 bytes of slack behind them); words at or beyond rL are not loaded
__device__ __forceinline__ void load_chars32(const uint8_t *f1, int w, int rL, uint64_t c[4])
{
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const int i0 = (w << 5) + 8 * q;
		c[q] = i0 < rL ? reinterpret_cast<const FrU64u *>(f1 + i0)->v : 0ull;
	}
}

// ... converted eight at a time: code in bits 2:1 of the letter (Gray), packed to 2 bits per base; `bad`: a character other than A/C/G/T in either case
__device__ __forceinline__ uint64_t pack_chars32(const uint64_t c[4], int w, int rL, uint64_t &bad)
{
	uint64_t word = 0;
	bad = 0;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const int i0 = (w << 5) + 8 * q;
		if (i0 >= rL) break;
		const uint64_t c8 = c[q];
		const uint64_t u = c8 & 0xDFDFDFDFDFDFDFDFull;
		auto zero_bytes = [](uint64_t t) { return ~(((t & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | t | 0x7F7F7F7F7F7F7F7Full); };
		uint64_t ok = zero_bytes(u ^ 0x4141414141414141ull) | zero_bytes(u ^ 0x4343434343434343ull) | zero_bytes(u ^ 0x4747474747474747ull) | zero_bytes(u ^ 0x5454545454545454ull);
		const int n = rL - i0 < 8 ? rL - i0 : 8;
		const uint64_t in = n >= 8 ? 0x8080808080808080ull : ((1ull << (8 * n)) - 1) & 0x8080808080808080ull;
		bad |= ~ok & in;
		uint64_t g2 = (c8 >> 1) & 0x0303030303030303ull;
		uint64_t code = g2 ^ ((g2 >> 1) & 0x0101010101010101ull);                 // A 0, C 1, G 2, T 3
		code = (code | (code >> 6)) & 0x000F000F000F000Full;
		code = (code | (code >> 12)) & 0x000000FF000000FFull;
		code = (code | (code >> 24)) & 0xFFFFull;
		word |= code << (16 * q);
	}
	return word;
}

__device__ __forceinline__ unsigned long long shfl_u64_frag(unsigned long long v, int src)
{
	return ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src);
}

__device__ __forceinline__ bool key_less(int g1, int r1, int g2, int r2) { return g1 == g2 ? r1 < r2 : g1 < g2; }   // CompByGenomePos

// vector<SeedPair_t> of one fragment in the LDS (fragment-relative coordinates)
typedef int16_t frp_t;               // fragment-relative positions and lengths (a side holds at most kFragMaxLen = 8192 bases)
struct FragPairs {
	frp_t *gPos, *rPos, *rLen, *gLen;
	uint8_t *simple;
	int num;
};

__device__ void erase_empty(FragPairs &v)
{
	int w = 0;
	for (int i = 0; i < v.num; ++i)
		if (v.rLen[i] != 0) {
			if (w != i) { v.gPos[w] = v.gPos[i]; v.rPos[w] = v.rPos[i]; v.rLen[w] = v.rLen[i]; v.gLen[w] = v.gLen[i]; v.simple[w] = v.simple[i]; }
			w++;
		}
	v.num = w;
}

// CheckSeedOverlapping, src/AlignmentCandidates.cpp:323-373
__device__ bool resolve_overlap(FragPairs &v, int i, int j)
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
	if (v.rLen[i] > 0 && v.rLen[j] > 0 && (ov = v.gPos[i] + v.gLen[i] - v.gPos[j]) > 0) {
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

// IdentifyNormalPairs(rlen, glen, v) for a fragment (glen > 0), src/AlignmentCandidates.cpp:420-490, by ONE lane on the LDS arrays, in two parts:
// filter_pairs = the three seed filters (:426-428), gap_pairs = the gap pairs between neighbours and the head / tail pairs (:437-488).
// byr: scratch of v.num entries (read-position order).  gap_pairs returns false for more pairs than `cap`.
__device__ void filter_pairs(FragPairs &v, uint16_t *byr, bool tandem_done = false, bool byr_done = false)
{
	if (v.num > 1) {
		// RemoveTandemRepeatSeeds, :235-260: every read position hit by more than one seed goes
		if (!tandem_done) {
			bool any = false;
			for (int i = 0; i < v.num; ++i) byr[i] = 0;
			for (int i = 0; i < v.num; ++i)
				for (int j = i + 1; j < v.num; ++j)
					if (v.rPos[i] == v.rPos[j]) { byr[i] = byr[j] = 1; any = true; }
			if (any) {
				for (int i = 0; i < v.num; ++i)
					if (byr[i]) v.rLen[i] = v.gLen[i] = 0;
				erase_empty(v);
			}
		}
		// RemoveTranslocatedSeeds, :262-321: byr[k] = index (in genome order) of the seed with the k-th smallest read position
		if (v.num > 1) {
			const int num = v.num;
			for (int i = 0; i < num && !byr_done; ++i) {
				int p = i;
				while (p > 0 && v.rPos[byr[p - 1]] > v.rPos[i]) { byr[p] = byr[p - 1]; --p; }
				byr[p] = (uint16_t)i;
			}
			bool any = false;
			for (int i = 0; i < num; ++i) {
				if (byr[i] == i) continue;
				any = true;
				int hi = byr[i];
				for (int j = i + 1; j <= hi; ++j)
					if (byr[j] > hi) hi = byr[j];
				int s1 = 0, s2 = 0;
				for (int k = i; k <= hi; ++k) {
					if (k < byr[k]) s1 += v.rLen[byr[k]];
					else s2 += v.rLen[byr[k]];
				}
				for (int k = i; k <= hi; ++k) {
					bool drop = s1 > s2 ? k > byr[k] : k < byr[k];
					if (drop) v.rLen[byr[k]] = v.gLen[byr[k]] = 0;
				}
				i = hi;
			}
			if (any) erase_empty(v);
		}
		// CheckOverlappingSeeds, :375-418
		if (v.num > 1) {
			const int num = v.num;
			bool any = false;
			for (int i = 0; i < num;) {
				if (v.rLen[i] > 0) {
					int r_end = v.rPos[i] + v.rLen[i] - 1, g_end = v.gPos[i] + v.gLen[i] - 1;
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
	}
}

// filter_pairs for up to 64 seeds by the whole wave, the seeds one per lane in REGISTERS (lane i holds seed i of the genome order): the same
// sequential algorithm, statement for statement, but every array access is a v_readlane (a write: compare + select) with a wave-uniform index (a few cycles)
// instead of a trip to the LDS by one lane (~100 cycles each, ~500 of them per task with 14 seeds: 53 k of a task's 84 k wave cycles in four
// tasks of ten, profiles/r05ze_frag_prof.log).  Only the two O(n^2) steps are done the parallel way (same result): the tandem marks (a lane
// compares its read position with everybody's) and the read-position order (rank by counting; the positions are distinct by then).
// Returns the new number of seeds, lane i holding seed i again.
__device__ __forceinline__ int rl_(int v, int idx) { return __builtin_amdgcn_readlane(v, idx); }
__device__ __forceinline__ void wl_(int &v, int idx, int val) { v = (int)(threadIdx.x & 63) == idx ? val : v; }     // (compare + select: this compiler has no writelane builtin)

__device__ int compact_seeds_wave(int num, int lane, int &G, int &R, int &L, int &GL)
{
	const uint64_t keep = __ballot(lane < num && L != 0);
	const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	const int cnt = __popcll(keep);
	const bool mine = (keep >> lane) & 1;
	const int dst = mine ? __popcll(keep & below) : cnt + __popcll(~keep & below);      // a permutation of the 64 lanes: the kept seeds first, in order
	G = __builtin_amdgcn_ds_permute(dst << 2, G); R = __builtin_amdgcn_ds_permute(dst << 2, R);
	L = __builtin_amdgcn_ds_permute(dst << 2, L); GL = __builtin_amdgcn_ds_permute(dst << 2, GL);
	return cnt;
}

__device__ bool resolve_overlap_wave(int &G, int &R, int &L, int &GL, int i, int j)      // resolve_overlap() on the lanes' registers
{
	bool master = true;
	int ov;
	if ((ov = rl_(R, i) + rl_(L, i) - rl_(R, j)) > 0) {
		const int li = rl_(L, i), lj = rl_(L, j);
		if (li < lj) {
			master = false;
			if (li > ov) { wl_(L, i, li - ov); wl_(GL, i, li - ov); }
			else { wl_(L, i, 0); wl_(GL, i, 0); }
		} else if (lj > ov) {
			wl_(R, j, rl_(R, j) + ov); wl_(G, j, rl_(G, j) + ov); wl_(L, j, lj - ov); wl_(GL, j, lj - ov);
		} else { wl_(L, j, 0); wl_(GL, j, 0); }
	}
	if (rl_(L, i) > 0 && rl_(L, j) > 0 && (ov = rl_(G, i) + rl_(GL, i) - rl_(G, j)) > 0) {
		const int li = rl_(L, i), lj = rl_(L, j);
		if (rl_(GL, i) < rl_(GL, j)) {
			master = false;
			if (li > ov) { wl_(L, i, li - ov); wl_(GL, i, li - ov); }
			else { wl_(L, i, 0); wl_(GL, i, 0); }
		} else if (lj > ov) {
			wl_(R, j, rl_(R, j) + ov); wl_(G, j, rl_(G, j) + ov); wl_(L, j, lj - ov); wl_(GL, j, lj - ov);
		} else { wl_(L, j, 0); wl_(GL, j, 0); }
	}
	return master;
}

__device__ int filter_pairs_wave(int num, int lane, int &G, int &R, int &L, int &GL)
{
	num = __builtin_amdgcn_readfirstlane(num);          // (wave-uniform, and the compiler must know: the loops below are scalar loops then, not masked vector ones)
	if (num <= 1) return num;
	// RemoveTandemRepeatSeeds, :235-260
	{
		bool dup = false;
		for (int j = 0; j < num; ++j) {
			const int rj = rl_(R, j);
			dup = dup || (lane < num && j != lane && rj == R);
		}
		if (__ballot(dup)) {
			if (dup) L = GL = 0;
			num = compact_seeds_wave(num, lane, G, R, L, GL);
		}
	}
	// RemoveTranslocatedSeeds, :262-321: B = byr, lane k holds the index (in genome order) of the seed with the k-th smallest read position
	if (num > 1) {
		int rank = 0;
		for (int j = 0; j < num; ++j) rank += rl_(R, j) < R ? 1 : 0;
		const int B = __builtin_amdgcn_ds_permute((lane < num ? rank : lane) << 2, lane);
		bool any = false;
		for (int i = 0; i < num; ++i) {
			const int bi = rl_(B, i);
			if (bi == i) continue;
			any = true;
			int hi = bi;
			for (int j = i + 1; j <= hi; ++j) { const int bj = rl_(B, j); if (bj > hi) hi = bj; }
			int s1 = 0, s2 = 0;
			for (int k = i; k <= hi; ++k) {
				const int bk = rl_(B, k), lk = rl_(L, bk);
				if (k < bk) s1 += lk; else s2 += lk;
			}
			for (int k = i; k <= hi; ++k) {
				const int bk = rl_(B, k);
				const bool drop = s1 > s2 ? k > bk : k < bk;
				if (drop) { wl_(L, bk, 0); wl_(GL, bk, 0); }
			}
			i = hi;
		}
		if (any) num = compact_seeds_wave(num, lane, G, R, L, GL);
	}
	// CheckOverlappingSeeds, :375-418
	if (num > 1) {
		bool any = false;
		for (int i = 0; i < num;) {
			const int li = rl_(L, i);
			if (li > 0) {
				const int r_end = rl_(R, i) + li - 1, g_end = rl_(G, i) + rl_(GL, i) - 1;
				for (int j = i + 1; j < num; ++j) {
					if (rl_(L, j) == 0) continue;
					if (r_end < rl_(R, j) && g_end < rl_(G, j)) break;
					if (!resolve_overlap_wave(G, R, L, GL, i, j)) break;
				}
				if (rl_(L, i) == 0) {
					any = true;
					int q = i - 1;
					while (q > 0 && rl_(L, q) == 0) q--;
					i = q < 0 ? 0 : q;
				} else i++;
			} else {
				any = true;
				i++;
			}
		}
		if (any) num = compact_seeds_wave(num, lane, G, R, L, GL);
	}
	return num;
}

// filter_pairs for more than 64 seeds (a fragment of thousands of bases inside a repeat): the two quadratic steps by the whole wave on the LDS arrays -- the
// tandem marks (every lane compares the read positions of its seeds with everybody's) and the read-position order (rank by counting: the positions
// are distinct once the tandem seeds are gone) --, the sequential rest by lane 0 as before.  By one lane the insertion sort alone was ~10 M wave cycles
// for 380 seeds: a handful of such tasks set the duration of the whole launch.  n_shared: the seed count as lane 0 publishes it.
__device__ void filter_pairs_wide(FragPairs &v, uint16_t *byr, int lane, int *n_shared)
{
	int num = __builtin_amdgcn_readfirstlane(v.num);
	if (num > 1) {
		bool mine = false;
		for (int i = lane; i < num; i += 64) {
			const int ri = v.rPos[i];
			bool dup = false;
			for (int j = 0; j < num; ++j) dup = dup || (j != i && v.rPos[j] == ri);
			byr[i] = dup ? 1 : 0;
			mine = mine || dup;
		}
		const bool any = __ballot(mine) != 0;
		__syncthreads();
		if (any) {
			if (lane == 0) {
				for (int i = 0; i < num; ++i)
					if (byr[i]) v.rLen[i] = v.gLen[i] = 0;
				erase_empty(v);
				*n_shared = v.num;
			}
			__syncthreads();
			num = __builtin_amdgcn_readfirstlane(*n_shared);
		}
		__syncthreads();
		for (int i = lane; i < num; i += 64) {
			const int ri = v.rPos[i];
			int rank = 0;
			for (int j = 0; j < num; ++j) rank += v.rPos[j] < ri ? 1 : 0;
			byr[rank] = (uint16_t)i;
		}
		__syncthreads();
	}
	if (lane == 0) {
		v.num = num;
		filter_pairs(v, byr, true, true);
		*n_shared = v.num;
	}
}

__device__ bool gap_pairs(int rlen, int glen, FragPairs &v, int cap)
{
	if (v.num > 1) {
		// the gaps between consecutive seeds, appended and then moved to their place in (gPos, rPos) order (:437-455; the keys are distinct)
		const int num = v.num;
		int added = 0;
		for (int i = 0, j = 1; j < num; ++i, ++j) {
			int r_gap = v.rPos[j] - (v.rPos[i] + v.rLen[i]);
			if (r_gap < 0) r_gap = 0;
			int g_gap = v.gPos[j] - (v.gPos[i] + v.gLen[i]);
			if (g_gap < 0) g_gap = 0;
			if (r_gap > 0 || g_gap > 0) {
				if (num + added >= cap) return false;
				int t = num + added++;
				v.simple[t] = 0;
				v.rPos[t] = v.rPos[i] + v.rLen[i];
				v.gPos[t] = v.gPos[i] + v.gLen[i];
				v.rLen[t] = r_gap; v.gLen[t] = g_gap;
			}
		}
		for (int t = num; t < num + added; ++t) {
			int xg = v.gPos[t], xr = v.rPos[t], xrl = v.rLen[t], xgl = v.gLen[t];
			uint8_t xs = v.simple[t];
			int p = t;
			while (p > 0 && key_less(xg, xr, v.gPos[p - 1], v.rPos[p - 1])) {
				v.gPos[p] = v.gPos[p - 1]; v.rPos[p] = v.rPos[p - 1]; v.rLen[p] = v.rLen[p - 1]; v.gLen[p] = v.gLen[p - 1]; v.simple[p] = v.simple[p - 1];
				--p;
			}
			v.gPos[p] = xg; v.rPos[p] = xr; v.rLen[p] = xrl; v.gLen[p] = xgl; v.simple[p] = xs;
		}
		v.num = num + added;
	}
	if (v.num > 0) {
		if (v.num + 2 > cap) return false;
		int r_gap = v.rPos[0] > 0 ? v.rPos[0] : 0;
		int g_gap = glen > 0 ? v.gPos[0] : r_gap;
		if (r_gap > 0 || g_gap > 0) {
			for (int p = v.num; p > 0; --p) {
				v.gPos[p] = v.gPos[p - 1]; v.rPos[p] = v.rPos[p - 1]; v.rLen[p] = v.rLen[p - 1]; v.gLen[p] = v.gLen[p - 1]; v.simple[p] = v.simple[p - 1];
			}
			int g = v.gPos[1] - g_gap;
			v.gPos[0] = g < 0 ? 0 : g;
			v.rPos[0] = 0; v.rLen[0] = r_gap; v.gLen[0] = g_gap; v.simple[0] = 0;
			v.num++;
		}
		int last = v.num - 1;
		r_gap = rlen - (v.rPos[last] + v.rLen[last]);
		g_gap = glen > 0 ? glen - (v.gPos[last] + v.gLen[last]) : r_gap;
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

// IdentifyNormalPairs for the common case, by the whole wave: the matches are already in read order as well (so the tandem and
// translocation filters, :235-321, find nothing) and no two neighbours overlap in either sequence (so CheckOverlappingSeeds, :375-418,
// changes nothing).  Then the result is the matches with the gap pairs between neighbours interleaved -- a gap pair sorts right behind
// the match it follows: its genome position is the match's end, at most the next match's start, and at a tie its read position is
// smaller -- plus the head and tail pairs (:437-488).  Returns the new count, or -1 without touching anything when the case is not that.
template <int kC>
__device__ int identify_normal_pairs_wave(int rlen, int glen, int n, int lane, int16_t *, frp_t *gPos, frp_t *rPos, frp_t *rLen, frp_t *gLen, uint8_t *simple)
{
	bool ok = true;
	for (int i = lane; i + 1 < n; i += 64)
		ok = ok && rPos[i] < rPos[i + 1] && rPos[i] + rLen[i] - 1 < rPos[i + 1] && gPos[i] + gLen[i] - 1 < gPos[i + 1];
	if (__ballot(!ok)) return -1;
	const bool head = rPos[0] > 0 || gPos[0] > 0;             // (:457-470; glen > 0 here: the genome side's gap is gPos[0])
	// every lane takes its matches (one per 64: n <= 64 kC) and the gap pair behind each into registers, then all write
	int mg[kC], mr[kC], ml_r[kC], ml_g[kC], dst[kC], gr[kC], gg[kC];
	bool has[kC];
	int before = head ? 1 : 0;                                  // output slots in front of this chunk
	for (int c = 0; c < kC; ++c) {
		const int i = c * 64 + lane;
		has[c] = false; dst[c] = -1; gr[c] = gg[c] = 0; mg[c] = mr[c] = ml_r[c] = ml_g[c] = 0;
		bool gap = false;
		if (i < n) {
			mg[c] = gPos[i]; mr[c] = rPos[i]; ml_r[c] = rLen[i]; ml_g[c] = gLen[i];
			if (i + 1 < n) {
				gr[c] = rPos[i + 1] - (mr[c] + ml_r[c]);
				gg[c] = gPos[i + 1] - (mg[c] + ml_g[c]);
				gap = gr[c] > 0 || gg[c] > 0;
			}
		}
		const uint64_t mgap = __ballot(gap);
		const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
		if (i < n) dst[c] = before + (i - c * 64) + __popcll(mgap & below);
		has[c] = gap;
		const int in_chunk = n - c * 64 < 64 ? (n - c * 64 > 0 ? n - c * 64 : 0) : 64;
		before += in_chunk + __popcll(mgap);
	}
	const int last = n - 1;
	const int t_r = rlen - (rPos[last] + rLen[last]), t_g = glen - (gPos[last] + gLen[last]);
	const int t_rp = rPos[last] + rLen[last], t_gp = gPos[last] + gLen[last];
	const int h_r = rPos[0] > 0 ? rPos[0] : 0, h_g = gPos[0];
	__syncthreads();
	for (int c = 0; c < kC; ++c) {
		if (dst[c] < 0) continue;
		const int d = dst[c];
		gPos[d] = mg[c]; rPos[d] = mr[c]; rLen[d] = ml_r[c]; gLen[d] = ml_g[c]; simple[d] = 1;
		if (has[c]) { gPos[d + 1] = mg[c] + ml_g[c]; rPos[d + 1] = mr[c] + ml_r[c]; rLen[d + 1] = gr[c]; gLen[d + 1] = gg[c]; simple[d + 1] = 0; }
	}
	int total = before;
	if (lane == 0) {
		if (head) { gPos[0] = 0; rPos[0] = 0; rLen[0] = h_r; gLen[0] = h_g; simple[0] = 0; }
		if (t_r > 0 || t_g > 0) { gPos[total] = t_gp; rPos[total] = t_rp; rLen[total] = t_r; gLen[total] = t_g; simple[total] = 0; }
	}
	if (t_r > 0 || t_g > 0) total++;
	__syncthreads();
	return total;
}

}  // namespace

// One wave per task of the level [level_begin[level], level_begin[level + 1]).
// Two instantiations can share a level (round 5, KG_FRAG_TWO_TIERS=1; off by default -- no gain, see launch_frag_partition): the SMALL one (sides up to kFragSmallLen, up to kFragSmallRuns matches: 4 KB of LDS, the CU's wave slots full)
// takes what fits it -- nearly every task: a fragment pair of a 7 kb read is ~350 x 350 bases with ~14 matches -- and leaves the rest (task.status 2) to the
// full-size one (15 KB, 10 waves per CU), which runs behind it on the same stream.  (It was built on the reading that the kernel waits on the LDS in two
// thirds of its wave cycles, profiles/r05o_pacbio_pmc_summary.json; what it waited for were the work lists' device-wide atomics -- see `pooled` below.)
template <int kMaxLen, int kMaxRuns, bool kSmall>
__global__ __launch_bounds__(64) void frag_partition_kernel(FragArgs a, int level)
{
	constexpr int kMaxPairs = 2 * kMaxRuns + 2;
	constexpr int kC = (kMaxRuns + 63) / 64;
	__shared__ uint64_t s_rd[kMaxLen / 32 + 2], s_tx[kMaxLen / 32 + 2];
	__shared__ frp_t s_gPos[kMaxPairs], s_rPos[kMaxPairs], s_rLen[kMaxPairs], s_gLen[kMaxPairs];
	__shared__ uint8_t s_simple[kMaxPairs];
	__shared__ frp_t s_run_r[kMaxRuns], s_run_d[kMaxRuns], s_run_l[kMaxRuns];
	__shared__ uint16_t s_byr[kMaxPairs];
	__shared__ int s_n, s_bad;
	const int lane = threadIdx.x;
	const unsigned long long t0 = a.ctl[FC_LEVEL0 + level], t1 = a.ctl[FC_LEVEL0 + level + 1];
	// The wave's own stretch of each work list (pieces, NW jobs, their op bytes): one atomic per ~50 tasks instead of three per task.  The counters
	// are device-scope atomics on ONE line that all eight XCDs share -- 2.3 M of them per 780 k tasks were what the kernel's two memory phases waited
	// for (100 k + 129 k of a task's 295 k wave cycles, profiles/r05w_frag_prof.log).  What a wave leaves unused stays behind: pieces and op bytes
	// nobody refers to, job slots as empty jobs.  Level 0 only -- the levels below hold a few hundred tasks.
	const bool pooled = level == 0;
	unsigned long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // [8] tasks whose seeds went through the filters, [9] ... whose gap pairs lane 0 made, [10] / [11] the wave cycles of either
	unsigned long long pc_next = 0, pc_end = 0, jb_next = 0, jb_end = 0, op_next = 0, op_end = 0;
	auto empty_jobs = [&](unsigned long long from, unsigned long long to) {
		if (to > (unsigned long long)a.job_capacity) to = (unsigned long long)a.job_capacity;
		for (unsigned long long j = from + (unsigned long long)lane; j < to; j += 64) { NwJobDesc jd; jd.o1 = 0; jd.o2 = 0; jd.ops = 0; jd.m = 0; jd.n = 0; a.jobs[j] = jd; }
	};
	// 64 tasks are looked at per step, one per lane: which of them are this instantiation's (the small one marks what outgrows it, status 2; with
	// a.one_tier the full-size one takes everything); the wave then works through those one after the other
	for (unsigned long long tb = t0 + (unsigned long long)blockIdx.x * 64; tb < t1; tb += (unsigned long long)gridDim.x * 64) {
	bool mine = false;
	long long my_f1 = 0, my_g = 0;                     // this lane's task of the 64: what the wave needs of it when its turn comes
	int my_rL = 0, my_gL = 0;
	if (tb + lane < t1) {
		FragTask &tk = a.tasks[tb + lane];
		my_f1 = tk.f1_off; my_g = tk.g; my_rL = tk.rL; my_gL = tk.gL;
		const bool small_fits = my_rL <= kFragSmallLen && my_gL <= kFragSmallLen;
		if (kSmall) { mine = small_fits; if (!small_fits) tk.status = 2; }
		else mine = a.one_tier || tk.status == 2 || !small_fits;
	}
	uint64_t todo = __ballot(mine);
	// The first 64 words (2048 bases) of both sides of the NEXT task are fetched while the wave works on this one: a task was a chain of six
	// dependent trips to memory -- its record, four character words behind a loop exit each, the text -- with two or three waves per SIMD to hide them
	// (the LDS arrays), 120 k of its 220 k wave cycles (profiles/r05x_ab_long_2m.log)
	uint64_t pf_c[4] = {0, 0, 0, 0}, pf_t = 0;
	int pf_for = -1;
	auto fetch = [&](int idx) {
		const long long f1o = (long long)shfl_u64_frag((unsigned long long)my_f1, idx), gg = (long long)shfl_u64_frag((unsigned long long)my_g, idx);
		const int rl = __shfl(my_rL, idx), gl = __shfl(my_gL, idx);
		pf_for = idx;
		if (!(rl > 30 && gl > 30) || rl > kMaxLen || gl > kMaxLen) return;
		if (lane < ((rl + 31) >> 5)) load_chars32(reinterpret_cast<const uint8_t *>(a.f1) + f1o, lane, rl, pf_c);
		if (lane < ((gl + 31) >> 5)) pf_t = text_word32_at(a.text, a.two_genome_size, gg + ((int64_t)lane << 5));
	};
	while (todo) {
		const int cur = __ffsll((unsigned long long)todo) - 1;
		const unsigned long long ti = tb + (unsigned long long)cur;
		todo &= todo - 1;
		FragTask &task = a.tasks[ti];
		const long long c0 = a.prof ? clock64() : 0;
		long long c1 = c0, c2 = c0, c3 = c0, c4 = c0;
		const int rL = __builtin_amdgcn_readlane(my_rL, cur), gL = __builtin_amdgcn_readlane(my_gL, cur);       // (uniform, in scalar registers)
		const long long task_f1_off = (long long)shfl_u64_frag((unsigned long long)my_f1, cur);
		const uint8_t *f1 = reinterpret_cast<const uint8_t *>(a.f1) + task_f1_off;
		const int64_t g = (int64_t)shfl_u64_frag((unsigned long long)my_g, cur);
		if (pf_for != cur) fetch(cur);
		const uint64_t cc0 = pf_c[0], cc1 = pf_c[1], cc2 = pf_c[2], cc3 = pf_c[3], ct = pf_t;
		if (todo) fetch(__ffsll((unsigned long long)todo) - 1);       // in flight until the next turn of this loop
		// lane 0 decides the task's pieces; everybody else helps with the runs
		bool whole_job = !(rL > 30 && gL > 30);            // src/tools.cpp:146
		bool host = false;
		int why = -1;                                      // (diagnostics: which limit sent the request back, FC_WHY)
		if (!whole_job && (rL > kMaxLen || gL > kMaxLen)) { host = true; why = 0; }
		int n_runs = 0;
		if (!whole_job && !host) {
			int max_shift;
			if (a.pacbio) {                                // :149-153
				max_shift = rL > gL ? (int)(rL * 0.2) : (int)(gL * 0.2);
				if (max_shift > 50) max_shift = 50;
			} else max_shift = a.max_gaps;
			if (lane == 0) { s_n = 0; s_bad = 0; }
			__syncthreads();
			// ---- the two fragments as 2-bit codes (the 8-mer code maps characters through nst_nt4_table and skips 'N': plain
			// A/C/G/T in either case is what a comparison of 2-bit codes reproduces; anything else goes back to the caller) ----
			const int rw = (rL + 31) >> 5, gw = (gL + 31) >> 5;
			for (int w = lane; w < rw; w += 64) {
				// 32 characters = four unaligned 8-byte loads, converted eight at a time (pack_chars32); the first 64 words came with the prefetch
				uint64_t c[4], bad;
				if (w < 64) { c[0] = cc0; c[1] = cc1; c[2] = cc2; c[3] = cc3; }
				else load_chars32(f1, w, rL, c);
				s_rd[w] = pack_chars32(c, w, rL, bad);
				if (bad) s_bad = 1;
			}
			for (int w = lane; w < gw; w += 64) s_tx[w] = w < 64 ? ct : text_word32_at(a.text, a.two_genome_size, g + ((int64_t)w << 5));
			__syncthreads();
			if (a.prof) c1 = clock64();
			if (s_bad) { host = true; why = 1; }
			else {
				// ---- runs of >= 8 equal bases along the diagonals |gpos - rpos| < max_shift (= the merged common 8-mers) ----
				for (int d = -(max_shift - 1) + lane; d <= max_shift - 1; d += 64) {
					const int t_lo = d < 0 ? -d : 0, t_hi = rL < gL - d ? rL : gL - d;      // read positions t with 0 <= t + d < gL
					int run = 0;
					// (the read side is word-aligned: one LDS read; the text side at base + d keeps the upper word of a step as the lower one of the next:
					//  two LDS reads per step instead of four.  Everything after the loads is 32-bit arithmetic: the 64-bit shifts this loop was made of --
					//  the funnel shift of the text, the range masks, the test for eight equal bases in a row -- run at a quarter of the rate; the text
					//  is shifted by v_alignbit on its 32-bit halves, the equal bases of each half are compressed to 16 bits, the rest works on one 32-bit mask)
					int base = t_lo & ~31;
					int wi = (base + d) >> 5;                           // (arithmetic shift: -1 for a start left of the text)
					const int sh = ((base + d) & 31) << 1;
					const bool wide = sh >= 32;
					const uint32_t s5 = (uint32_t)sh & 31u;
					uint64_t tlo = (wi >= 0 && wi < gw) ? s_tx[wi] : 0;
					auto eq16 = [](uint32_t x) {                        // 16 bases of 2 bits: bit i = base i is 0 (equal)
						uint32_t y = ~(x | (x >> 1)) & 0x55555555u;
						y = (y | (y >> 1)) & 0x33333333u;
						y = (y | (y >> 2)) & 0x0F0F0F0Fu;
						y = (y | (y >> 4)) & 0x00FF00FFu;
						return (y | (y >> 8)) & 0xFFFFu;
					};
					for (; base < t_hi; base += 32, ++wi) {
						const uint64_t thi = (wi + 1 >= 0 && wi + 1 < gw) ? s_tx[wi + 1] : 0;
						const uint64_t rdw = s_rd[base >> 5];
						const uint32_t t0 = (uint32_t)tlo, t1 = (uint32_t)(tlo >> 32), t2 = (uint32_t)thi, t3 = (uint32_t)(thi >> 32);
						tlo = thi;
						const uint32_t a0 = wide ? t1 : t0, a1 = wide ? t2 : t1, a2 = wide ? t3 : t2;
						const uint32_t xl = (uint32_t)rdw ^ __builtin_amdgcn_alignbit(a1, a0, s5), xh = (uint32_t)(rdw >> 32) ^ __builtin_amdgcn_alignbit(a2, a1, s5);
						uint32_t m = eq16(xl) | (eq16(xh) << 16);                           // bit i: read base base + i equals text base base + d + i
						const int lo = t_lo > base ? t_lo - base : 0, hi = t_hi - base < 32 ? t_hi - base : 32;
						if (lo != 0 || hi != 32) m &= (hi >= 32 ? ~0u : (1u << hi) - 1u) & ~((1u << lo) - 1u);
						if (m == 0xffffffffu) { run += 32; continue; }
						const int t = __ffs((int)~m) - 1;                                   // equal bases from position 0 up
						// the common case -- 98 of 99 diagonals are not the alignment's -- has no 8 equal bases in a row in this word and the
						// run carried in does not reach 8 either: only the equal bases at the top of the word are carried on
						{
							uint32_t q8 = m & (m >> 1);
							q8 &= q8 >> 2;
							q8 &= q8 >> 4;
							if (q8 == 0 && run + t < 8) { run = __clz((int)~m); continue; }
						}
						// the runs of equal bases in this word, without walking every run and gap (a 15 %-error diagonal alternates ~10 times per word, and
						// the whole wave waited for the one lane on the alignment's diagonal: round 5): the run entering the word ends at the first zero; runs
						// of >= 8 inside the word start where eight ones in a row begin; the run touching the top is carried on
						auto emit = [&](int start, int len) {
							int k = atomicAdd(&s_n, 1);
							if (k < kMaxRuns) { s_run_r[k] = start; s_run_d[k] = d; s_run_l[k] = len; }
						};
						if (run + t >= 8) emit(base - run, run + t);
						const int top = __clz((int)~m);                                     // equal bases from position 31 down (m != all ones)
						uint32_t in = m & ~((2u << t) - 1u);                                // what lies strictly inside: not the run at the bottom, not the one at the top
						if (top) in &= ~(0xffffffffu << (32 - top));
						uint32_t r8 = in & (in >> 1);
						r8 &= r8 >> 2;
						r8 &= r8 >> 4;                                                      // bit p: the bases p .. p + 7 are equal
						uint32_t starts = r8 & ~(r8 << 1);
						while (starts) {
							const int pp = __ffs((int)starts) - 1;
							starts &= starts - 1;
							emit(base + pp, __ffs((int)~(in >> pp)) - 1);
						}
						run = top;
					}
					if (run >= 8) {
						int k = atomicAdd(&s_n, 1);
						if (k < kMaxRuns) { s_run_r[k] = t_hi - run; s_run_d[k] = d; s_run_l[k] = run; }
					}
				}
				__syncthreads();
				n_runs = __builtin_amdgcn_readfirstlane(s_n);
				if (n_runs > kMaxRuns) {
					if (kSmall) { if (lane == 0) task.status = 2; __syncthreads(); continue; }       // (more matches than the small arrays hold: the full-size kernel's)
					host = true; why = 2;
				}
				if (a.prof) c2 = clock64();
			}
			if (!host && n_runs > 0) {
				// sort(SimplePairVec, CompByGenomePos), src/KmerAnalysis.cpp:177: rank of every run among the others (keys are distinct)
				for (int i = lane; i < n_runs; i += 64) {
					const int gi = s_run_r[i] + s_run_d[i], ri = s_run_r[i];
					int rank = 0;
					for (int j = 0; j < n_runs; ++j)
						if (key_less(s_run_r[j] + s_run_d[j], s_run_r[j], gi, ri)) rank++;
					s_gPos[rank] = gi; s_rPos[rank] = ri; s_rLen[rank] = s_gLen[rank] = s_run_l[i]; s_simple[rank] = 1;
				}
			}
			__syncthreads();
			if (a.prof) c3 = clock64();
		}
		// IdentifyNormalPairs on the LDS arrays: by the whole wave where the seed filters have nothing to do, else by lane 0 (they are
		// sequential by nature)
		// (round 5: where the filters DO have something to do -- four tasks in ten carry a chance match on a foreign diagonal -- lane 0 runs the filters
		//  alone and the whole wave then interleaves the gap pairs as in the clean case; inserting them one by one into the sorted LDS arrays was most
		//  of what the kernel waited for: 2.3 k LDS instructions per task, 67 % of its wave cycles waiting, profiles/r05o_pacbio_pmc_summary.json)
		int fast_num = -1;
		const bool eligible = !host && !whole_job && n_runs > 0;
		if (eligible && !a.no_fast_pairs) fast_num = identify_normal_pairs_wave<kC>(rL, gL, n_runs, lane, nullptr, s_gPos, s_rPos, s_rLen, s_gLen, s_simple);
		if (fast_num < 0) {
			FragPairs v;
			v.gPos = s_gPos; v.rPos = s_rPos; v.rLen = s_rLen; v.gLen = s_gLen; v.simple = s_simple; v.num = n_runs;
			const long long cf0 = a.prof ? clock64() : 0;
			if (a.prof && eligible) prof_acc[8] += 1;
			if (eligible && n_runs <= 64 && !a.no_fast_pairs) {
				int G = 0, R = 0, L = 0, GL = 0;
				if (lane < n_runs) { G = s_gPos[lane]; R = s_rPos[lane]; L = s_rLen[lane]; GL = s_gLen[lane]; }
				const int kept = filter_pairs_wave(n_runs, lane, G, R, L, GL);
				__syncthreads();
				if (lane < kept) { s_gPos[lane] = (frp_t)G; s_rPos[lane] = (frp_t)R; s_rLen[lane] = (frp_t)L; s_gLen[lane] = (frp_t)GL; s_simple[lane] = 1; }
				if (lane == 0) s_n = kept;
			} else if (eligible && !a.no_fast_pairs) filter_pairs_wide(v, s_byr, lane, &s_n);
			else if (lane == 0) {
				if (eligible) filter_pairs(v, s_byr);
				s_n = eligible ? v.num : 0;
			}
			__syncthreads();
			if (a.prof) prof_acc[10] += (unsigned long long)(clock64() - cf0);
			const int n2 = __builtin_amdgcn_readfirstlane(s_n);
			__syncthreads();
			if (eligible && n2 > 0 && !a.no_fast_pairs) fast_num = identify_normal_pairs_wave<kC>(rL, gL, n2, lane, nullptr, s_gPos, s_rPos, s_rLen, s_gLen, s_simple);
			const long long cg0 = a.prof ? clock64() : 0;
			if (a.prof && fast_num < 0 && eligible && n2 > 0) prof_acc[9] += 1;
			if (lane == 0 && fast_num < 0) {
				v.num = n2;
				bool h2 = host, wj = whole_job;
				if (eligible && n2 > 0 && !gap_pairs(rL, gL, v, kMaxPairs)) { h2 = true; s_bad = 2; }
				if (eligible && v.num == 0) wj = true;                       // no common 8-mer survived: the whole fragment is one alignment (:214-221)
				s_n = h2 ? -1 : wj ? 0 : v.num;
			}
			if (a.prof && fast_num < 0) { __syncthreads(); prof_acc[11] += (unsigned long long)(clock64() - cg0); }
		}
		if (lane == 0 && fast_num >= 0) s_n = fast_num;
		__syncthreads();
		if (a.prof) c4 = clock64();
		// ... and the whole wave turns the pairs into the task's pieces: literal runs, NW jobs, sub-tasks -- counted first, reserved with ONE
		// atomic per list, then written, every lane its own pair (a lane-0 loop over ~27 pairs with a global store each was 95 k of a task's 320 k cycles)
		const int num = __builtin_amdgcn_readfirstlane(s_n);     // -1: outside the envelope; 0: one alignment for the whole fragment
		host = num < 0;
		if (host && why < 0) why = s_bad == 2 ? 3 : 2;
		int first = 0, count = 0;
		if (!host) {
			// what pair i becomes: 0 nothing, 1 literal, 2 NW job, 3 sub-task
			auto kind_of = [&](int i, int &prl, int &pgl) {
				if (num == 0) { prl = rL; pgl = gL; return i == 0 ? 2 : 0; }
				if (i >= num) { prl = pgl = 0; return 0; }
				prl = s_rLen[i]; pgl = s_gLen[i];
				if (prl <= 0 && pgl <= 0) return 0;
				if (pgl == 0 || prl == 0 || (prl == 1 && pgl == 1) || s_simple[i]) return 1;
				if (a.pacbio && (prl > 300 || pgl > 300)) return 3;                   // the recursion, :197
				return 2;
			};
			const int n_items = num == 0 ? 1 : num;
			int n_pieces = 0, n_jobs = 0, n_sub = 0;
			long long n_ops = 0;
			for (int base = 0; base < n_items; base += 64) {
				int prl, pgl;
				const int kd = kind_of(base + lane, prl, pgl);
				n_pieces += __popcll(__ballot(kd != 0)); n_jobs += __popcll(__ballot(kd == 2)); n_sub += __popcll(__ballot(kd == 3));
				long long o = kd == 2 ? prl + pgl : 0;
				for (int off = 32; off > 0; off >>= 1) o += __shfl_xor(o, off);
				n_ops += o;
			}
			unsigned long long piece_at = 0, job_at = 0, ops_at = 0, task_at = 0;
			{
				const bool np = pc_next + (unsigned long long)n_pieces > pc_end;
				const bool nj = n_jobs && jb_next + (unsigned long long)n_jobs > jb_end;
				const bool no = n_jobs && op_next + (unsigned long long)n_ops > op_end;
				const unsigned long long wp = pooled && n_pieces < kFragPieceChunk ? (unsigned long long)kFragPieceChunk : (unsigned long long)n_pieces;
				const unsigned long long wj = pooled && n_jobs < kFragJobChunk ? (unsigned long long)kFragJobChunk : (unsigned long long)n_jobs;
				const unsigned long long wo = pooled && n_ops < kFragOpsChunk ? (unsigned long long)kFragOpsChunk : (unsigned long long)n_ops;
				if (nj) empty_jobs(jb_next, jb_end);                 // what is left of the old stretch
				unsigned long long bp = 0, bj = 0, bo = 0;
				if (lane == 0) {
					if (np) bp = atomicAdd(&a.ctl[FC_PIECES], wp);
					if (nj) bj = atomicAdd(&a.ctl[FC_JOBS], wj);
					if (no) bo = atomicAdd(&a.ctl[FC_OPS], wo);
					if (n_sub) task_at = atomicAdd(&a.ctl[FC_TASKS], (unsigned long long)n_sub);
				}
				if (np) { pc_next = shfl_u64_frag(bp, 0); pc_end = pc_next + wp; }
				if (nj) { jb_next = shfl_u64_frag(bj, 0); jb_end = jb_next + wj; }
				if (no) { op_next = shfl_u64_frag(bo, 0); op_end = op_next + wo; }
				task_at = shfl_u64_frag(task_at, 0);
				piece_at = pc_next; job_at = jb_next; ops_at = op_next;
			}
			const bool jobs_fit = job_at + (unsigned long long)n_jobs <= (unsigned long long)a.job_capacity;
			const bool room = piece_at + (unsigned long long)n_pieces <= (unsigned long long)a.piece_capacity && jobs_fit &&
			                  ops_at + (unsigned long long)n_ops <= (unsigned long long)a.ops_capacity &&
			                  (n_sub == 0 || (task_at + (unsigned long long)n_sub <= (unsigned long long)a.task_capacity && level + 1 < kFragMaxDepth));
			if (!room) {
				host = true;
				why = (n_sub != 0 && level + 1 >= kFragMaxDepth) ? 5 : 4;
				// (nothing is taken from the wave's stretches -- what is left of them ends as empty jobs, so that the NW kernels find nothing there; sub-task
				//  slots beyond the capacity were never written, the ones inside it become tasks of nothing)
				for (int j = lane; j < n_sub; j += 64)
					if (task_at + (unsigned long long)j < (unsigned long long)a.task_capacity) {
						FragTask sub;
						sub.f1_off = task_f1_off; sub.g = g; sub.rL = 0; sub.gL = 0; sub.first = 0; sub.count = 0; sub.status = 0; sub.root = task.root;
						a.tasks[task_at + (unsigned long long)j] = sub;
					}
			} else {
				first = (int)piece_at;
				count = n_pieces;
				pc_next += (unsigned long long)n_pieces; jb_next += (unsigned long long)n_jobs; op_next += (unsigned long long)n_ops;
				unsigned long long p_run = piece_at, j_run = job_at, o_run = ops_at, t_run = task_at;
				for (int base = 0; base < n_items; base += 64) {
					const int i = base + lane;
					int prl, pgl;
					const int kd = kind_of(i, prl, pgl);
					const uint64_t mp = __ballot(kd != 0), mj = __ballot(kd == 2), ms = __ballot(kd == 3);
					const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
					long long o = kd == 2 ? prl + pgl : 0, incl = o;                    // exclusive prefix of the op bytes of the jobs before this lane
					for (int off = 1; off < 64; off <<= 1) { long long t = __shfl_up(incl, off); if (lane >= off) incl += t; }
					const int64_t rp = num == 0 ? 0 : s_rPos[i < num ? i : 0], gp = num == 0 ? 0 : s_gPos[i < num ? i : 0];
					if (kd != 0) {
						FragPiece pc;
						if (kd == 1) {
							if (pgl == 0) { pc.kind = KG_OP_GAP2; pc.v = prl; }                 // read characters against '-' (:172-176)
							else if (prl == 0) { pc.kind = KG_OP_GAP1; pc.v = pgl; }            // '-' against genome characters (:177-181)
							else { pc.kind = KG_OP_DIAG; pc.v = prl; }                          // one base each, or an exact match
						} else if (kd == 2) {
							const unsigned long long jx = j_run + (unsigned long long)__popcll(mj & below);
							NwJobDesc jd;
							jd.o1 = task_f1_off + rp; jd.o2 = g + gp; jd.ops = (int64_t)(o_run + (unsigned long long)(incl - o)); jd.m = prl; jd.n = pgl;
							a.jobs[jx] = jd;
							pc.kind = FP_JOB; pc.v = (int32_t)jx;
						} else {
							const unsigned long long tx = t_run + (unsigned long long)__popcll(ms & below);
							FragTask sub;
							sub.f1_off = task_f1_off + rp; sub.g = g + gp; sub.rL = prl; sub.gL = pgl;
							sub.first = 0; sub.count = 0; sub.status = 0; sub.root = task.root;
							a.tasks[tx] = sub;
							pc.kind = FP_TASK; pc.v = (int32_t)tx;
						}
						a.pieces[p_run + (unsigned long long)__popcll(mp & below)] = pc;
					}
					p_run += (unsigned long long)__popcll(mp); j_run += (unsigned long long)__popcll(mj); t_run += (unsigned long long)__popcll(ms);
					o_run += (unsigned long long)__shfl(incl, 63);
				}
			}
		}
		if (lane == 0) {
			task.first = first; task.count = host ? 0 : count; task.status = host ? 1 : 0;
			if (host) { a.status[task.root] = 1; atomicAdd(&a.ctl[FC_WHY + (why < 0 ? 4 : why)], 1ull); }                                              // the whole request goes back to the caller
			if (a.prof) {
				const long long c5 = clock64();
				prof_acc[0] += (unsigned long long)(c1 - c0); prof_acc[1] += (unsigned long long)(c2 - c1); prof_acc[2] += (unsigned long long)(c3 - c2);
				prof_acc[3] += (unsigned long long)(c4 - c3); prof_acc[4] += (unsigned long long)(c5 - c4); prof_acc[5] += 1ull;
				prof_acc[6] += (unsigned long long)n_runs; prof_acc[7] += (unsigned long long)(rL + gL);
			}
		}
		__syncthreads();
	}
	}
	empty_jobs(jb_next, jb_end);
	if (a.prof && lane == 0)                        // (per wave, not per task: eight atomics per task on one line were a fifth of the profiled run)
		for (int k = 0; k < 12; ++k) atomicAdd(&a.ctl[FC_PROF + k], prof_acc[k]);
}

// the tasks appended while `level` was processed are the next level
__global__ void frag_level_kernel(FragArgs a, int level)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		unsigned long long n = a.ctl[FC_TASKS];
		if (n > (unsigned long long)a.task_capacity) n = (unsigned long long)a.task_capacity;
		a.ctl[FC_LEVEL0 + level + 2] = n;
	}
}

__global__ void frag_reset_kernel(FragArgs a)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0) {
		a.ctl[FC_TASKS] = (unsigned long long)a.n; a.ctl[FC_PIECES] = 0; a.ctl[FC_JOBS] = 0; a.ctl[FC_OPS] = 0;
		a.ctl[FC_LEVEL0] = 0; a.ctl[FC_LEVEL0 + 1] = (unsigned long long)a.n;
		for (int k = 0; k < 12; ++k) a.ctl[FC_PROF + k] = 0;
		for (int k = 0; k < 6; ++k) a.ctl[FC_WHY + k] = 0;
		for (int l = 2; l <= kFragMaxDepth + 1; ++l) a.ctl[FC_LEVEL0 + l] = (unsigned long long)a.n;
	}
	for (int64_t r = i; r < a.n; r += (int64_t)gridDim.x * blockDim.x) {
		FragTask t;
		t.f1_off = a.off1[r]; t.g = a.gpos[r]; t.rL = a.rlen ? a.rlen[r] : (int32_t)(a.off1[r + 1] - a.off1[r]); t.gL = a.glen[r];
		t.first = 0; t.count = 0; t.status = 0; t.root = (int32_t)r;
		a.tasks[r] = t;
		a.status[r] = 0;
	}
}

// One lane per request: its op string from the pieces, depth first.  The bytes leave eight at a time (an accumulator per lane, unaligned 8-byte
// stores; literal runs as a fill pattern, job op strings through unaligned 8-byte loads): one byte store per column and lane -- 64 different
// cache lines per store instruction -- was 73 ms per 400 k long reads (profiles/r05f_pacbio_kernel_stats.csv).
__global__ __launch_bounds__(256) void frag_stitch_kernel(FragArgs a)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n; r += (int64_t)gridDim.x * blockDim.x) {
		if (a.status[r]) { a.aln_len[r] = 0; if (a.runs) a.runs[r] = 0; continue; }
		uint8_t *out = a.ops + a.ops_off[r];
		int at = 0, na = 0;                    // bytes stored; bytes waiting in acc
		uint64_t acc = 0;
		int nr = 0;                            // runs of equal ops so far
		uint64_t last = 255;                   // the op written last
		auto put = [&](uint64_t b) {
			acc |= b << (8 * na);
			if (++na == 8) { reinterpret_cast<FrU64u *>(out + at)->v = acc; at += 8; acc = 0; na = 0; }
		};
		int stack_task[kFragMaxDepth + 1], stack_piece[kFragMaxDepth + 1];
		int sp = 0;
		stack_task[0] = (int)r; stack_piece[0] = 0;
		while (sp >= 0) {
			const FragTask &t = a.tasks[stack_task[sp]];
			if (stack_piece[sp] >= t.count) { sp--; continue; }
			const FragPiece pc = a.pieces[t.first + stack_piece[sp]++];
			if (pc.kind <= KG_OP_GAP2) {
				int v = pc.v;
				if (v > 0 && (uint64_t)pc.kind != last) { nr++; last = (uint64_t)pc.kind; }
				while (na != 0 && v > 0) { put((uint64_t)pc.kind); --v; }
				const uint64_t pat = (uint64_t)pc.kind * 0x0101010101010101ull;
				for (; v >= 8; v -= 8) { reinterpret_cast<FrU64u *>(out + at)->v = pat; at += 8; }
				for (; v > 0; --v) put((uint64_t)pc.kind);
			} else if (pc.kind == FP_JOB) {
				const uint8_t *src = a.job_ops + a.jobs[pc.v].ops;
				int L = a.job_len[pc.v], k = 0;
				while (na != 0 && k < L) { const uint64_t b = src[k++]; if (b != last) { nr++; last = b; } put(b); }
				for (; k + 8 <= L; k += 8) {
					const uint64_t w = reinterpret_cast<const FrU64u *>(src + k)->v;
					const uint64_t y = w ^ ((w << 8) | last);                    // every byte against the one before it (ops are 0 .. 2)
					nr += __popcll((y | (y >> 1)) & 0x0101010101010101ull);
					last = w >> 56;
					reinterpret_cast<FrU64u *>(out + at)->v = w; at += 8;
				}
				for (; k < L; ++k) { const uint64_t b = src[k]; if (b != last) { nr++; last = b; } put(b); }
			} else if (sp < kFragMaxDepth) { sp++; stack_task[sp] = pc.v; stack_piece[sp] = 0; }
		}
		for (int k = 0; k < na; ++k) out[at + k] = (uint8_t)(acc >> (8 * k));
		a.aln_len[r] = at + na;
		if (a.runs) a.runs[r] = nr;
	}
}

static inline int grid_of(int64_t items, int block, int max_blocks)
{
	int64_t g = (items + block - 1) / block;
	if (g < 1) g = 1;
	if (g > max_blocks) g = max_blocks;
	return (int)g;
}

int64_t frag_pool_waves(int64_t n_requests, int n_cu)
{
	const int64_t g = n_requests / 64 + 1;
	const int64_t big = std::min<int64_t>(g, (int64_t)n_cu * kFragWavesPerCu), small = getenv("KG_FRAG_TWO_TIERS") ? std::min<int64_t>(g, (int64_t)n_cu * kFragSmallWavesPerCu) : 0;
	return big + small;
}

hipError_t launch_frag_partition(const FragArgs &a, int n_cu, hipStream_t stream)
{
	hipLaunchKernelGGL(frag_reset_kernel, dim3(grid_of(a.n, 256, n_cu * 8)), dim3(256), 0, stream, a);
	for (int level = 0; level < kFragMaxDepth; ++level) {
		// (the level's task count lives on the device: the grid is sized for the requests at level 0 and for a share of them below)
		const int64_t guess = level == 0 ? a.n : a.n / 4 + 1024;
		// (tried in round 5 and left as an A/B aid, KG_FRAG_TWO_TIERS=1: the small instantiation in front -- 4 KB of LDS, the CU's wave slots full -- changes
		//  nothing, 4.83 s against 4.74 s per 2 M long reads, profiles/r05r_ab_long_2m.log -- more waves only queued for the same three atomic counters,
		//  which is what bound the kernel then; measured again with the counters pooled: 4.54 against 4.64 s, profiles/r05y_ab_long_2m.log, inside the runs' spread)
		FragArgs b = a;
		b.one_tier = getenv("KG_FRAG_TWO_TIERS") != nullptr ? 0 : 1;
		if (!b.one_tier) hipLaunchKernelGGL((frag_partition_kernel<kFragSmallLen, kFragSmallRuns, true>), dim3(grid_of(guess / 64 + 1, 1, n_cu * kFragSmallWavesPerCu)), dim3(64), 0, stream, b, level);
		hipLaunchKernelGGL((frag_partition_kernel<kFragMaxLen, kFragMaxRuns, false>), dim3(grid_of(guess / 64 + 1, 1, n_cu * kFragWavesPerCu)), dim3(64), 0, stream, b, level);
		hipLaunchKernelGGL(frag_level_kernel, dim3(1), dim3(64), 0, stream, a, level);
	}
	return hipGetLastError();
}

hipError_t launch_frag_stitch(const FragArgs &a, int n_cu, hipStream_t stream)
{
	hipLaunchKernelGGL(frag_stitch_kernel, dim3(grid_of(a.n, 256, n_cu * 16)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

}  // namespace kg
