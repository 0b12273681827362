// seed_kernels.hip -- batched maximal-exact-match seeding on gfx950.
//
// Replaces, for a whole batch of reads at once, the reference's per-read call chain
//   IdentifySeedPairs_FastMode / _SensitiveMode  (src/AlignmentCandidates.cpp:49-80, 132-169)
//     -> BWT_Search                              (src/bwt_search.cpp:140-184)
//          -> bwt_2occ4 / bwt_occ4               (src/bwt_search.cpp:68-118)
//          -> bwt_sa -> bwt_invPsi -> bwt_occ    (src/bwt_search.cpp:44-66, 120-138)
//   + the final std::sort by (PosDiff,rPos) / (gPos,rPos).
//
// Three kernels:
//   search_kernel : persistent lanes, one read per lane.  A read is a chain of dependent
//                   searches (the next start depends on the previous match length), a search is a
//                   chain of dependent LF steps; each loop iteration performs at most one LF step
//                   per lane = one or two 64-byte Occ-block gathers.  Lanes whose read is
//                   exhausted pull the next read from a global queue with a wave-aggregated
//                   atomic (ballot + mbcnt), so a wave keeps its 64 intervals live until the batch
//                   is drained.  Only the reverse-complement side of BWA's bi-interval is tracked:
//                   on a forward+revcomp text it is a plain backward search and needs ONE base's
//                   rank at two positions per step instead of all four (SURVEY.md App. C check).
//                   Output: a dense list of hits {interval start, size, rPos, len, read, seed slot}.
//   locate_kernel : persistent lanes over (hit, i) items.  SAMPLED mode walks LF until a sampled
//                   rank (bwt_sa); FULL mode is one gather from the expanded suffix array.  The
//                   text position of the pattern is 2L - SA[x1+i] - len.
//   sort_kernel   : per-read ordering with the mode's comparator (total order, so the result is
//                   unique and equals the reference's std::sort output).
#include "seed_kernels.hpp"

#include <hipcub/hipcub.hpp>
#include <cstdlib>

namespace kg {

__device__ __forceinline__ uint64_t l2_of(const FmView &ix, int c)
{
	return c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : c == 3 ? ix.L2[3] : ix.L2[4];
}

// index of this lane among the set bits of `mask` below it
__device__ __forceinline__ int lane_rank(uint64_t mask)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Wave-local ticket pool.  A single global counter hit by every wave every iteration saturates
// at ~90 M atomics/s on one word (MI355X_MICROARCH.md "dequeue"), which was the whole kernel's
// ceiling; instead one atomic reserves kPoolChunk consecutive tickets for the wave and lanes draw
// from that pool with ballot-prefix arithmetic.  `next`/`end` are wave-uniform.
constexpr unsigned long long kPoolChunk = 256;

struct WavePool {
	unsigned long long next = 0, end = 0;
};

__device__ __forceinline__ unsigned long long pool_take(WavePool &p, unsigned long long *counter, bool want)
{
	uint64_t mask = __ballot(want);
	if (mask == 0) return 0;
	unsigned long long cnt = (unsigned long long)__popcll(mask);
	unsigned long long avail = p.end - p.next;
	unsigned long long rank = (unsigned long long)lane_rank(mask);
	unsigned long long ticket = p.next + rank;
	if (cnt > avail) {  // wave-uniform branch
		unsigned long long base = 0;
		int leader = __ffsll((unsigned long long)mask) - 1;
		if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, kPoolChunk);
		base = __shfl(base, leader);
		if (rank >= avail) ticket = base + (rank - avail);
		p.next = base + (cnt - avail);
		p.end = base + kPoolChunk;
	} else {
		p.next += cnt;
	}
	return ticket;
}

// ---- read window ------------------------------------------------------------------------------
// The ABI hands over one byte per base (the reference's EncodeSeq).  The search loop keeps the
// read's codes in registers: a "window word" holds 16 positions, 4 bits each.  A word is made from
// ONE unaligned 16-byte load of the raw bytes (gfx950 global loads need no alignment) and ~60 bit
// ops; bytes > 3 become nibbles > 3 (ambiguous) and positions >= rlen read as 4, which is also what
// makes SensitiveMode's look past the read end behave like 'N' (SURVEY.md App. B-10).
struct __attribute__((packed, aligned(1))) Raw16 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t nibbles_of(uint32_t x)   // 4 bytes -> 4 nibbles in the low 16 bits
{
	uint32_t y = x & 0xFCFCFCFCu;                                            // anything above 3?
	uint32_t nz = (((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;     // 0x80 per non-zero byte
	uint32_t nib = (x & 0x03030303u) | (nz >> 5);                            // 0..3, or 4..7 when ambiguous
	nib = (nib | (nib >> 4)) & 0x00FF00FFu;
	return (nib | (nib >> 8)) & 0x0000FFFFu;
}

__device__ __forceinline__ uint64_t window_word(const uint8_t *enc, int64_t n_bases, int64_t base, int rlen, int w)
{
	int64_t at = base + ((int64_t)w << 4);
	int valid = rlen - (w << 4);                  // positions of this word that exist
	uint32_t a, b, c, d;
	if (at + 16 <= n_bases) {
		Raw16 v = *reinterpret_cast<const Raw16 *>(enc + at);
		a = v.x; b = v.y; c = v.z; d = v.w;
	} else {                                      // last bytes of the batch: stay inside the buffer
		uint32_t t[4] = {0, 0, 0, 0};
		for (int j = 0; j < 16; ++j)
			if (at + j < n_bases) t[j >> 2] |= (uint32_t)enc[at + j] << ((j & 3) << 3);
		a = t[0]; b = t[1]; c = t[2]; d = t[3];
	}
	uint64_t word = (uint64_t)(nibbles_of(a) | (nibbles_of(b) << 16)) | ((uint64_t)(nibbles_of(c) | (nibbles_of(d) << 16)) << 32);
	uint64_t keep = valid >= 16 ? ~0ull : valid <= 0 ? 0ull : ((1ull << (valid << 2)) - 1);
	return (word & keep) | (0x4444444444444444ull & ~keep);
}

// Pack pre-pass: 16 lanes per read, lane w of a group builds window word w (one unaligned 16-byte
// load each, consecutive lanes read consecutive bytes).  Read r's words start at packed word
// (read_off[r] >> 4) + 3 r -- a closed form, no scan -- which leaves room for the two all-'N'
// padding words every read gets.  Cost: 150 B in + 104 B out per read, against ~10 KB of index
// gathers per read in the search kernel.
__global__ __launch_bounds__(256) void pack_reads_kernel(SeedArgs a)
{
	int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	int64_t n_groups = ((int64_t)gridDim.x * blockDim.x) >> 4;
	int w0 = threadIdx.x & 15;
	for (int64_t r = group; r < a.n_reads; r += n_groups) {
		int64_t base = a.read_off[r];
		int rlen = (int)(a.read_off[r + 1] - base);
		uint64_t *out = a.packed + (base >> 4) + 3 * r;
		int words = (rlen >> 4) + 3;
		for (int w = w0; w < words; w += 16) out[w] = window_word(a.enc, a.n_bases, base, rlen, w);
	}
}

// ---- search ---------------------------------------------------------------------------------------
// Two-level persistent loop, one read per lane:
//   tight loop : nothing but extension steps -- window nibble, two 12/16-byte rank gathers, popcount,
//                interval update (~45 instructions per wave iteration).  A lane whose search ends
//                parks (`pending`) and the wave keeps stepping the others.
//   slow path  : entered when kRefill lanes are parked (or none is live): D records hits and
//                advances the read position, A draws new reads from the wave's pool, B starts the
//                next search (q-mer table jump).  Its ~350 instructions are amortised over ~8
//                iterations; parked lanes cost ~6 % of the lane-iterations.
// History (profiles/): a single state-machine loop spent 57 % of its wave cycles in
// SQ_WAIT_INST_ANY -- ~300 instructions and ~30 exec-mask branches per iteration, because at wave
// level some lane always needs the rare path.
constexpr int kRefill = 8;

template <typename idx_t>
__global__ __launch_bounds__(256) void search_kernel(SeedArgs a)
{
	__shared__ idx_t l2s[8];
	if (threadIdx.x < 5) l2s[threadIdx.x] = (idx_t)a.ix.L2[threadIdx.x];
	__syncthreads();
	const FmView &ix = a.ix;
	const idx_t primary = (idx_t)ix.primary;
	const bool fast = a.mode == KG_MODE_FAST;
	const int msl = a.min_seed_len;
	const idx_t occ_thr = (idx_t)a.occ_thr;
	const bool have_tab = sizeof(idx_t) == 4 ? ix.qtab32 != nullptr : ix.qtab64 != nullptr;

	bool have_read = false, done = false, active = false, pending = false;
	int r = 0, rlen = 0, pos = 0, stop_pos = 0, end_pos = 0, seed_cnt = 0;
	int cur = 0, stop = 0, wword = 0;
	const uint64_t *pw = a.packed;
	uint64_t win = 0x4444444444444444ull, wnext = 0x4444444444444444ull;
	idx_t k = 0, n = 0;
	uint32_t c_search = 0, c_lf = 0, c_lf2 = 0;
	WavePool read_pool, hit_pool;

	for (;;) {
		// ================= slow path =================
		// ---- D: searches that ended in the tight loop -------------------------------------------------
		{
			int len = cur - pos;
			bool hit = pending && len >= msl && n <= occ_thr;
			unsigned long long slot = pool_take(hit_pool, a.hit_count, hit);
			if (hit) {
				uint4 *dst = reinterpret_cast<uint4 *>(a.hits + slot);
				uint64_t k64 = (uint64_t)k;
				dst[0] = make_uint4((uint32_t)k64, (uint32_t)(k64 >> 32), (uint32_t)r, (uint32_t)pos);
				dst[1] = make_uint4((uint32_t)len, (uint32_t)n, (uint32_t)seed_cnt, 0u);
				seed_cnt += (int)n;
			}
			if (pending) {
				int adv = fast ? len + 1 : (hit ? len : msl);                 // :74 / :157-161
				pos += adv;
				stop_pos += fast ? 0 : adv;
				stop_pos = stop_pos > rlen ? rlen : stop_pos;               // :163
				pending = false;
			}
		}
		// ---- A/B passes until every lane is live or done ----------------------------------------------
		for (;;) {
			bool want_read = !have_read && !done;
			if (__ballot(want_read)) {
				unsigned long long t = pool_take(read_pool, a.read_queue, want_read);
				if (want_read) {
					if (t >= (unsigned long long)a.n_reads) done = true;
					else {
						r = (int)t;
						int64_t base = a.read_off[t];
						rlen = (int)(a.read_off[t + 1] - base);
						pw = a.packed + (base >> 4) + 3 * (int64_t)t;
						win = pw[0];
						wnext = pw[1];
						wword = 0;
						pos = 0; stop_pos = 30; end_pos = rlen - msl; seed_cnt = 0;
						have_read = true;
					}
				}
			}
			bool idle = have_read && !active;
			bool finished = idle && pos >= end_pos;
			if (finished) {
				a.seeds_per_read[r] = seed_cnt;
				have_read = false;
			}
			bool starting = idle && !finished;
			int w = pos >> 4;
			if (starting && w != wword) {           // the new start lies outside the window's first word
				if (w == wword + 1) win = wnext; else win = pw[w];
				wnext = pw[w + 1];
				wword = w;
			}
			int code0 = (int)((win >> ((pos & 15) << 2)) & 15);
			bool skip = starting && code0 > 3;          // ambiguous base: FastMode :59, SensitiveMode :142
			pos += skip ? 1 : 0;
			stop_pos += skip ? 1 : 0;
			bool go = starting && !skip;
			if (go) {
				// the next kQmer codes as a 2-bit packed index (code at pos in the lowest bits)
				int sh = (pos & 15) << 2;
				uint64_t x = sh ? (win >> sh) | (wnext << (64 - sh)) : win;
				x &= (1ull << (4 * kQmer)) - 1;
				bool clean = (x & 0x4444444444444444ull) == 0 && pos + kQmer <= (fast ? rlen : stop_pos) && have_tab;
				uint64_t y = x & 0x3333333333333333ull;
				y = (y | (y >> 2)) & 0x0F0F0F0F0F0F0F0Full;
				y = (y | (y >> 4)) & 0x00FF00FF00FF00FFull;
				y = (y | (y >> 8)) & 0x0000FFFF0000FFFFull;
				y = (y | (y >> 16)) & 0x00000000FFFFFFFFull;
				idx_t tk = 0, tn = 0;
				uint32_t tlf2 = 0;
				if (clean) {
					if (sizeof(idx_t) == 4) {
						uint2 e = ix.qtab32[y];
						tk = (idx_t)e.x; tn = (idx_t)(e.y & 0x0FFFFFFFu); tlf2 = e.y >> 28;
					} else {
						uint4 e = ix.qtab64[y];
						tk = (idx_t)(((uint64_t)e.y << 32) | e.x); tn = (idx_t)e.z; tlf2 = e.w;
					}
				}
				bool jump = clean && tn != 0;
				k = jump ? tk : l2s[3 - code0] + 1;                            // x[1], :149
				n = jump ? tn : l2s[code0 + 1] - l2s[code0];                  // x[2], :150
				cur = pos + (jump ? kQmer : 1);
				// the reference performs these kQmer-1 steps one by one; keep its block accounting
				c_lf += jump ? (uint32_t)(kQmer - 1) : 0u;
				c_lf2 += jump ? tlf2 : 0u;
				stop = fast ? rlen : stop_pos;
				active = true;
				c_search++;
				if ((cur >> 4) != wword) { win = wnext; wword++; wnext = pw[wword + 1]; }
			}
			if (__ballot(!done && !active) == 0) break;
		}
		if (__ballot(active) == 0) break;               // every lane is done

		// ================= tight loop =================
		for (;;) {
			int code = (int)((win >> ((cur & 15) << 2)) & 15);
			bool do_step = active && cur < stop && code <= 3;
			idx_t nn = 0, ok = 0;
			int c = 3 - code;
			if (do_step) {
				idx_t kk = k - 1, ll = k - 1 + n;                          // bwt_2occ4(x1-1, x1-1+x2), :157
				kk -= (kk >= primary);
				ll -= (ll >= primary);
				uint4 vk = ix.planes[(((uint64_t)(kk >> 6)) << 2) + (uint32_t)c];
				uint4 vl = ix.planes[(((uint64_t)(ll >> 6)) << 2) + (uint32_t)c];
				uint64_t mk = (2ull << (kk & 63)) - 1, ml = (2ull << (ll & 63)) - 1;
				ok = (idx_t)(((uint64_t)vk.w << 32) | vk.z) + (idx_t)__popcll((((uint64_t)vk.y << 32) | vk.x) & mk);
				idx_t ol = (idx_t)(((uint64_t)vl.w << 32) | vl.z) + (idx_t)__popcll((((uint64_t)vl.y << 32) | vl.x) & ml);
				nn = ol - ok;
				c_lf++;
				c_lf2 += (kk >> 7) != (ll >> 7) ? 1u : 0u;                 // reference 128-symbol block accounting
			}
			bool cont = nn != 0;                                            // implies do_step
			if (cont) {
				k = l2s[c] + 1 + ok;
				n = nn;
				cur++;
				if ((cur & 15) == 0) { win = wnext; wword++; wnext = pw[wword + 1]; }
			}
			pending = pending || (active && !cont);
			active = cont;
			uint64_t parked = __ballot(pending);
			if (__popcll(parked) >= kRefill || __ballot(active) == 0) break;
		}
	}
	// the slots this wave reserved but never filled are marked empty for the locate kernel
	for (unsigned long long x = hit_pool.next + (threadIdx.x & 63); x < hit_pool.end; x += 64) a.hits[x].n = 0;
	// work counters: one atomic per wave
	uint64_t s0 = c_search, s1 = c_lf - c_lf2, s2 = c_lf2;
	for (int off = 32; off > 0; off >>= 1) {
		s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off);
	}
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&a.counters[0], (unsigned long long)s0);
		atomicAdd(&a.counters[1], (unsigned long long)s1);
		atomicAdd(&a.counters[2], (unsigned long long)s2);
	}
}

// hit -> first output slot of the hit, given the scanned per-read seed offsets
__device__ __forceinline__ int64_t hit_out_base(const SeedArgs &a, const Hit &h)
{
	return a.seed_off[h.read] + h.seed_start;
}

__global__ __launch_bounds__(256) void locate_sampled_kernel(SeedArgs a)
{
	const FmView &ix = a.ix;
	const unsigned long long n_hits = *a.hit_count;
	bool done = false, walking = false;
	Hit h;
	h.n = 0;
	int i = 0;
	uint64_t k = 0, steps = 0;
	uint32_t c_inv = 0, c_sa = 0;
	WavePool pool;
	for (;;) {
		for (;;) {
			bool idle = !done && !walking;
			if (__ballot(idle) == 0) break;
			bool want = idle && i >= h.n;
			unsigned long long t = pool_take(pool, a.locate_queue, want);
			if (want) {
				if (t >= n_hits) done = true;
				else { h = a.hits[t]; i = 0; }
			}
			if (idle && !done && i < h.n) {
				k = h.k + (uint64_t)i;
				steps = 0;
				walking = true;
			}
		}
		if (__ballot(!done) == 0) break;
		if (walking) {
			if ((k & 31) == 0) {
				uint64_t sa = steps + ix.sa[k >> 5];                       // bwt_sa :128-138
				int64_t out = hit_out_base(a, h) + i;
				if (out < a.seed_capacity) {
					kg_seed s;
					s.gPos = (int64_t)(ix.seq_len - sa - (uint64_t)h.len);  // revcomp side -> pattern side
					s.rPos = h.rpos; s.len = h.len;
					a.seeds[out] = s;
				}
				c_sa++;
				i++;
				walking = false;
			} else {
				k = lf_step_plane(ix, k);
				steps++;
				c_inv++;
			}
		}
	}
	uint64_t s0 = c_inv, s1 = c_sa;
	for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); }
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&a.counters[3], (unsigned long long)s0);
		atomicAdd(&a.counters[4], (unsigned long long)s1);
	}
}

__global__ __launch_bounds__(256) void locate_full_kernel(SeedArgs a)
{
	const FmView &ix = a.ix;
	const unsigned long long n_hits = *a.hit_count;
	unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (; t < n_hits; t += stride) {
		Hit h = a.hits[t];
		int64_t out = hit_out_base(a, h);
		for (int i = 0; i < h.n; ++i) {
			uint64_t sa = ix.fsa32 ? (uint64_t)ix.fsa32[h.k + i] : ix.fsa64[h.k + i];
			if (out + i < a.seed_capacity) {
				kg_seed s;
				s.gPos = (int64_t)(ix.seq_len - sa - (uint64_t)h.len);
				s.rPos = h.rpos; s.len = h.len;
				a.seeds[out + i] = s;
			}
		}
	}
}

// comparators of the two seeding modes (reference src/AlignmentCandidates.cpp:11-21)
__device__ __forceinline__ bool seed_less(const kg_seed &x, const kg_seed &y, int mode)
{
	if (mode == KG_MODE_FAST) {
		int64_t dx = x.gPos - x.rPos, dy = y.gPos - y.rPos;
		return dx == dy ? x.rPos < y.rPos : dx < dy;
	}
	return x.gPos == y.gPos ? x.rPos < y.rPos : x.gPos < y.gPos;
}

__global__ __launch_bounds__(256) void sort_kernel(SeedArgs a)
{
	// One read per lane, in place.  Most reads have 2-4 seeds (plain insertion sort = the last gap), but
	// reads inside repeat families carry hundreds (up to 50 per search), so the passes are Shell's
	// (Ciura gaps): ~n^1.3 moves instead of n^2/4 for the lane that would otherwise hold up its wave.
	const int gaps[8] = {701, 301, 132, 57, 23, 10, 4, 1};
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; r < a.n_reads; r += stride) {
		int64_t lo = a.seed_off[r], hi = a.seed_off[r + 1];
		if (hi > a.seed_capacity) hi = a.seed_capacity;
		int64_t n = hi - lo;
		if (n < 2) continue;
		kg_seed *s = a.seeds + lo;
#pragma unroll 1
		for (int gi = 0; gi < 8; ++gi) {
			int64_t gap = gaps[gi];
			if (gap >= n) continue;
			for (int64_t i = gap; i < n; ++i) {
				kg_seed v = s[i];
				int64_t j = i - gap;
				while (j >= 0 && seed_less(v, s[j], a.mode)) { s[j + gap] = s[j]; j -= gap; }
				s[j + gap] = v;
			}
		}
	}
}

// ---- chaining: GenerateAlignmentCandidateForIlluminaSeq / ForPacBioSeq --------------------------------
// (reference src/AlignmentCandidates.cpp:82-130, 171-224).  One read per lane; the seed lists are short
// (2-4 seeds for 150 bp reads, ~170 for a 7 kb PacBio read), the logic is a sequential greedy scan.
__device__ __forceinline__ int64_t contig_end_of(const ChainArgs &a, int64_t g)   // GetAlignmentBoundary, src/tools.cpp:399-404
{
	int lo = 0, hi = a.n_ends - 1;
	while (lo < hi) {
		int mid = (lo + hi) >> 1;
		if (a.contig_end[mid] < g) lo = mid + 1; else hi = mid;
	}
	return a.contig_end[lo];
}

__global__ __launch_bounds__(256) void chain_kernel(ChainArgs a)
{
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; r < a.n_reads; r += stride) {
		const int64_t base = a.seed_off[r];
		const int num = (int)(a.seed_off[r + 1] - base);
		const int rlen = (int)(a.read_off[r + 1] - a.read_off[r]);
		const kg_seed *s = a.seeds + base;
		kg_candidate *out = a.cands + base;
		kg_seed *cs = a.cand_seeds + base;
		int nc = 0;
		int64_t used = 0;
		int i = 0;
		while (i < num && s[i].gPos - s[i].rPos < 0) i++;
		if (!a.pacbio) {
			int thr = rlen / 5;                       // (int)(rlen*0.2) for non-negative rlen
			if (thr > 50) thr = 50;
			while (i < num) {
				int score = s[i].len;
				int64_t g_end = contig_end_of(a, s[i].gPos);
				int j = i, k = i + 1;
				for (; k < num; ++k) {
					int64_t dk = s[k].gPos - s[k].rPos, dj = s[j].gPos - s[j].rPos;
					if (s[k].gPos > g_end || dk - dj > a.max_gaps) break;
					score += s[k].len;
					j = k;
				}
				if (score > thr) {
					if (score - 50 > thr) thr = score - 50;
					int64_t d = s[i].gPos - s[i].rPos;
					kg_candidate c;
					c.posDiff = d < 0 ? 0 : d; c.score = score; c.count = k - i; c.first = base + used;
					// the candidate's seeds, re-sorted by (gPos, rPos) (:118): insertion sort while copying
					for (int q = i; q < k; ++q) {
						kg_seed v = s[q];
						int64_t p = used + (q - i);
						while (p > used && (cs[p - 1].gPos > v.gPos || (cs[p - 1].gPos == v.gPos && cs[p - 1].rPos > v.rPos))) { cs[p] = cs[p - 1]; --p; }
						cs[p] = v;
					}
					used += k - i;
					out[nc++] = c;
				}
				i = k;
			}
		} else {
			uint8_t *taken = a.taken + base;
			for (int q = 0; q < num; ++q) taken[q] = 0;
			int thr = 0;
			for (; i < num; ++i) {
				if (taken[i]) continue;
				int score = s[i].len;
				taken[i] = 1;
				int64_t first = used;
				// tentative: the picked seeds are written at cs[used..]; kept only if the score qualifies.  A seed
				// is taken at most once whether or not its candidate is kept (reference TakenArr), so the
				// slots of a rejected candidate are simply reused by the next one.
				cs[used++] = s[i];
				int j = i;
				for (int k = i + 1; k < num; ++k) {
					if (taken[k]) continue;
					int64_t dk = s[k].gPos - s[k].rPos, dj = s[j].gPos - s[j].rPos;
					int64_t dd = dk - dj;
					if ((dd < 0 ? -dd : dd) < 300) {
						if (s[k].rPos > s[j].rPos) {
							score += s[k].len;
							cs[used++] = s[k];
							taken[k] = 1;
							j = k;
						}
					} else if (s[k].gPos - s[j].gPos > 1000) break;
				}
				if (score >= thr) {
					thr = score;
					int64_t d = s[i].gPos - s[i].rPos;
					kg_candidate c;
					c.posDiff = d < 0 ? 0 : d; c.score = score; c.count = (int32_t)(used - first); c.first = base + first;
					out[nc++] = c;
				} else used = first;
			}
		}
		a.n_cands[r] = nc;
	}
}

hipError_t launch_chain_batch(const ChainArgs &a, int n_cu, hipStream_t stream)
{
	if (a.n_reads <= 0) return hipSuccess;
	int64_t g = (a.n_reads + 255) / 256;
	if (g > (int64_t)n_cu * 16) g = (int64_t)n_cu * 16;
	hipLaunchKernelGGL(chain_kernel, dim3((unsigned)g), dim3(256), 0, stream, a);
	return hipGetLastError();
}

__global__ void finish_offsets_kernel(SeedArgs a)
{
	// seed_off[n_reads] = total; record overflow and the output counters
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		int64_t total = a.seed_off[a.n_reads - 1] + a.seeds_per_read[a.n_reads - 1];
		a.seed_off[a.n_reads] = total;
		a.counters[5] = (unsigned long long)total;
		a.counters[6] = (unsigned long long)a.n_bases;
		a.counters[7] = total > a.seed_capacity ? (unsigned long long)total : 0ull;
	}
}

// full suffix array expansion at index load: one chain per sample, walking LF from a sampled
// rank until the next sampled rank, writing SA[k] = SA[sample] - steps on the way.
__global__ __launch_bounds__(256) void expand_sa_kernel(FmView ix, uint64_t n_sa, uint32_t *fsa32, uint64_t *fsa64)
{
	uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; j < n_sa; j += stride) {
		uint64_t k = j << 5;
		uint64_t p = j == 0 ? ix.seq_len : ix.sa[j];
		for (;;) {
			if (fsa32) fsa32[k] = (uint32_t)p; else fsa64[k] = p;
			k = lf_step_plane(ix, k);
			p--;
			if ((k & 31) == 0) break;
		}
	}
}

// Index-load-time conversion of the reference's 2-bit Occ/BWT blocks into the bit-plane layout.
// One thread per 64-symbol block.
__global__ __launch_bounds__(256) void build_planes_kernel(const uint32_t *occ, uint64_t n_blocks64, uint4 *planes)
{
	uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; b < n_blocks64; b += stride) {
		const uint32_t *rb = occ + ((b >> 1) << 4);      // reference block of 128 symbols
		int half = (int)(b & 1);
		uint64_t cnt[4];
		for (int c = 0; c < 4; ++c) cnt[c] = (uint64_t)rb[2 * c] | ((uint64_t)rb[2 * c + 1] << 32);
		uint64_t plane[4] = {0, 0, 0, 0};
		for (int j = 0; j < 128; ++j) {
			uint32_t w = rb[8 + (j >> 4)];
			int sym = (w >> (30 - 2 * (j & 15))) & 3;
			if (j < 64 && half) cnt[sym]++;              // first half counted into the second half's base
			else if ((j >> 6) == half) plane[sym] |= 1ull << (j & 63);
		}
		for (int c = 0; c < 4; ++c)
			planes[(b << 2) + c] = make_uint4((uint32_t)plane[c], (uint32_t)(plane[c] >> 32), (uint32_t)cnt[c], (uint32_t)(cnt[c] >> 32));
	}
}

// q-mer table: entry id encodes the codes LSB first (code of the first base in bits 1:0); the value
// is the state BWT_Search (reference src/bwt_search.cpp:147-168) reaches after those kQmer bases.
__global__ __launch_bounds__(256) void build_qtab_kernel(FmView ix, uint2 *t32, uint4 *t64)
{
	uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; id < (1ull << (2 * kQmer)); id += stride) {
		int p = (int)(id & 3);
		uint64_t k = l2_of(ix, 3 - p) + 1, n = l2_of(ix, p + 1) - l2_of(ix, p);
		uint32_t lf2 = 0;
		for (int j = 1; j < kQmer && n != 0; ++j) {
			int c = 3 - (int)((id >> (2 * j)) & 3);
			uint64_t kk = k - 1, ll = k - 1 + n;
			kk -= (kk >= ix.primary);
			ll -= (ll >= ix.primary);
			lf2 += (kk >> 7) != (ll >> 7);
			uint64_t ok = rank_plane(ix, kk, c), ol = rank_plane(ix, ll, c);
			n = ol - ok;
			k = l2_of(ix, c) + 1 + ok;
		}
		if (t32) {
			if (n >= (1ull << 28)) n = 0;      // not representable: the search falls back to single steps
			t32[id] = make_uint2((uint32_t)k, (uint32_t)n | (lf2 << 28));
		} else {
			if (n >= (1ull << 32)) n = 0;
			t64[id] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)n, lf2);
		}
	}
}

__global__ void seed_reset_kernel(unsigned long long *ctl)
{
	if (threadIdx.x < kCtlWords) ctl[threadIdx.x] = 0;
}

static inline int grid_for(int64_t items, int block, int max_blocks)
{
	int64_t g = (items + block - 1) / block;
	if (g < 1) g = 1;
	if (g > max_blocks) g = max_blocks;
	return (int)g;
}

hipError_t launch_build_planes(const uint32_t *occ, uint64_t n_blocks64, uint4 *planes, hipStream_t stream)
{
	hipLaunchKernelGGL(build_planes_kernel, dim3(grid_for((int64_t)n_blocks64, 256, 256 * 64)), dim3(256), 0, stream, occ, n_blocks64, planes);
	return hipGetLastError();
}

hipError_t launch_build_qtab(const FmView &ix, uint2 *t32, uint4 *t64, hipStream_t stream)
{
	hipLaunchKernelGGL(build_qtab_kernel, dim3(256 * 32), dim3(256), 0, stream, ix, t32, t64);
	return hipGetLastError();
}

hipError_t launch_expand_sa(const FmView &ix, uint64_t n_sa, uint32_t *fsa32, uint64_t *fsa64, hipStream_t stream)
{
	hipLaunchKernelGGL(expand_sa_kernel, dim3(grid_for((int64_t)n_sa, 256, 256 * 32)), dim3(256), 0, stream, ix, n_sa, fsa32, fsa64);
	return hipGetLastError();
}

struct WidenOp {
	__host__ __device__ __forceinline__ int64_t operator()(const int32_t &x) const { return (int64_t)x; }
};
using WideIter = hipcub::TransformInputIterator<int64_t, WidenOp, const int32_t *>;

size_t scan_temp_bytes(int64_t max_reads)
{
	size_t bytes = 0;
	WideIter it((const int32_t *)nullptr, WidenOp());
	hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, it, (int64_t *)nullptr, (int)max_reads);
	return bytes;
}

hipError_t launch_seed_batch(const SeedArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream, hipEvent_t *ev)
{
	if (a.n_reads <= 0) return hipSuccess;
	hipError_t e;
	// queue heads, hit count and counters are zeroed on the stream every call
	hipLaunchKernelGGL(seed_reset_kernel, dim3(1), dim3(64), 0, stream, a.read_queue);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	// persistent lanes: 8 blocks of 256 threads per CU (= 32 waves/CU) unless the batch is smaller
	int per_cu = 8;
	if (const char *env = getenv("KG_SEARCH_BLOCKS_PER_CU")) per_cu = atoi(env) > 0 ? atoi(env) : 8;  // tuning knob
	int blocks = grid_for(a.n_reads, 256, n_cu * per_cu);
	hipLaunchKernelGGL(pack_reads_kernel, dim3(grid_for(a.n_reads * 16, 256, n_cu * 32)), dim3(256), 0, stream, a);
	if (ev) (void)hipEventRecord(ev[0], stream);
	// 32-bit interval arithmetic whenever the text allows it; KG_FORCE_U64 exercises the wide instantiation
	// (the one hg38-sized indexes use) on small test indexes
	static const bool force_wide = getenv("KG_FORCE_U64") != nullptr;
	if (a.ix.seq_len < 0xFFFFFF00ull && !(force_wide && a.ix.qtab64))
		hipLaunchKernelGGL(search_kernel<uint32_t>, dim3(blocks), dim3(256), 0, stream, a);
	else
		hipLaunchKernelGGL(search_kernel<uint64_t>, dim3(blocks), dim3(256), 0, stream, a);
	if (ev) (void)hipEventRecord(ev[1], stream);
	size_t tb = scan_temp_bytes;
	WideIter it(a.seeds_per_read, WidenOp());
	if ((e = hipcub::DeviceScan::ExclusiveSum(scan_temp, tb, it, a.seed_off, (int)a.n_reads, stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(finish_offsets_kernel, dim3(1), dim3(64), 0, stream, a);
	if (ev) (void)hipEventRecord(ev[2], stream);
	if (a.ix.fsa32 || a.ix.fsa64)
		hipLaunchKernelGGL(locate_full_kernel, dim3(grid_for(a.max_hits, 256, n_cu * 8)), dim3(256), 0, stream, a);
	else
		hipLaunchKernelGGL(locate_sampled_kernel, dim3(grid_for(a.max_hits, 256, n_cu * 8)), dim3(256), 0, stream, a);
	if (ev) (void)hipEventRecord(ev[3], stream);
	hipLaunchKernelGGL(sort_kernel, dim3(grid_for(a.n_reads, 256, n_cu * 16)), dim3(256), 0, stream, a);
	if (ev) (void)hipEventRecord(ev[4], stream);
	return hipGetLastError();
}

}  // namespace kg
