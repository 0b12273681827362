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
//                   When the interval has shrunk to a single suffix (and the full SA is resident) the
//                   lane stops ranking: one gather from the SA names the text position, and the
//                   rest of the match is a 48-bases-per-iteration comparison of the read against
//                   the 2-bit text -- same match length by construction (an interval of one extends
//                   iff the next text base equals the next read base), ~4x fewer cache lines.
//                   Output: a dense list of hits {interval start, size, rPos, len, read, seed slot}.
//   locate_kernel : persistent lanes over (hit, i) items.  SAMPLED mode walks LF until a sampled
//                   rank (bwt_sa); FULL mode is one gather from the expanded suffix array.  The
//                   text position of the pattern is 2L - SA[x1+i] - len.
//   sort_*_kernel : per-read ordering with the mode's comparator (total order, so the result is
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

__device__ __forceinline__ unsigned long long pool_take(WavePool &p, unsigned long long *counter, bool want, unsigned long long chunk = kPoolChunk)
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
		if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, chunk);
		base = __shfl(base, leader);
		if (rank >= avail) ticket = base + (rank - avail);
		p.next = base + (cnt - avail);
		p.end = base + chunk;
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

// nst_nt4_table on four characters at once: A/C/G/T in either case carry their code in bits 2:1 (00, 01, 11, 10 -- one
// Gray-to-binary step away), everything else becomes 4
__device__ __forceinline__ uint32_t ascii_to_codes(uint32_t x)
{
	uint32_t g = (x >> 1) & 0x03030303u;
	uint32_t code = g ^ ((g >> 1) & 0x01010101u);
	uint32_t u = x & 0xDFDFDFDFu;
	auto nonzero = [](uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };   // 0x80 per non-zero byte
	uint32_t other = nonzero(u ^ 0x41414141u) & nonzero(u ^ 0x43434343u) & nonzero(u ^ 0x47474747u) & nonzero(u ^ 0x54545454u);   // 0x80: none of ACGT
	uint32_t m = (other >> 7) * 0xFFu;                                         // 0xFF per such byte
	return (code & ~m) | (0x04040404u & m);
}

__device__ __forceinline__ uint64_t window_word(const uint8_t *enc, int64_t n_bases, int64_t base, int rlen, int w, bool ascii = false)
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
	if (ascii) { a = ascii_to_codes(a); b = ascii_to_codes(b); c = ascii_to_codes(c); d = ascii_to_codes(d); }
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
		for (int w = w0; w < words; w += 16) out[w] = window_word(a.enc, a.n_bases, base, rlen, w, a.ascii != 0);
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

struct __attribute__((packed, aligned(1))) U64u { uint64_t v; };

// 16 bases of 2 bits -> 16 nibbles (the base in the low bits of its nibble)
__device__ __forceinline__ uint64_t spread_2bit(uint32_t v)
{
	uint64_t x = v;
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
	x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
	x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x << 2)) & 0x3333333333333333ull;
	return x;
}

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
	const int q = ix.qmer;
	const uint64_t qmask = q >= 16 ? ~0ull : (1ull << (4 * q)) - 1;
	const bool direct = ix.text != nullptr && (ix.fsa32 != nullptr || ix.fsa64 != nullptr);

	bool have_read = false, done = false, active = false, pending = false;
	int r = 0, rlen = 0, pos = 0, stop_pos = 0, end_pos = 0, seed_cnt = 0;
	int cur = 0, stop = 0, wword = 0;
	const uint64_t *pw = a.packed;
	int64_t rbase = 0;                    // raw-code variant (a.packed == nullptr): window words are made on the fly
	const bool raw = a.packed == nullptr;
#define KG_WORD(w) (raw ? window_word(a.enc, a.n_bases, rbase, rlen, (w), a.ascii != 0) : pw[(w)])
	uint64_t win = 0x4444444444444444ull, wnext = 0x4444444444444444ull;
	idx_t k = 0, n = 0;
	// how the lane extends its match: 0 = rank (LF) steps, 1 = interval of one, fetch its suffix, 2 = compare
	// the read with the text at tpos (the text position facing read position `cur`)
	int mode = 0;
	idx_t tpos = 0;
	uint32_t c_search = 0, c_lf = 0, c_lf2 = 0, c_sa = 0, c_dbg = 0;
	const int dbg = a.debug_count;        // 1 table lookups, 2 LF steps executed, 3 of them with kk/ll in different 128-byte lines, 4 text rounds
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
				bool at_text = mode == 2;                                   // finished against the text: position known
				uint64_t k64 = at_text ? (uint64_t)tpos - (uint64_t)len : (uint64_t)k;
				dst[0] = make_uint4((uint32_t)k64, (uint32_t)(k64 >> 32), (uint32_t)r, (uint32_t)pos);
				dst[1] = make_uint4((uint32_t)len, (uint32_t)n, (uint32_t)seed_cnt, at_text ? 1u : 0u);
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
						rbase = base;
						win = KG_WORD(0);
						wnext = KG_WORD(1);
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
				if (w == wword + 1) win = wnext; else win = KG_WORD(w);
				wnext = KG_WORD(w + 1);
				wword = w;
			}
			int code0 = (int)((win >> ((pos & 15) << 2)) & 15);
			bool skip = starting && code0 > 3;          // ambiguous base: FastMode :59, SensitiveMode :142
			pos += skip ? 1 : 0;
			stop_pos += skip ? 1 : 0;
			bool go = starting && !skip;
			if (go) {
				// the next q codes as a 2-bit packed index (code at pos in the lowest bits)
				int sh = (pos & 15) << 2;
				uint64_t x = sh ? (win >> sh) | (wnext << (64 - sh)) : win;
				x &= qmask;
				bool clean = (x & 0x4444444444444444ull) == 0 && pos + q <= (fast ? rlen : stop_pos) && have_tab;
				uint64_t y = x & 0x3333333333333333ull;
				y = (y | (y >> 2)) & 0x0F0F0F0F0F0F0F0Full;
				y = (y | (y >> 4)) & 0x00FF00FF00FF00FFull;
				y = (y | (y >> 8)) & 0x0000FFFF0000FFFFull;
				y = (y | (y >> 16)) & 0x00000000FFFFFFFFull;
				idx_t tk = 0, tn = 0;
				uint32_t tlf2 = 0;
				bool tsa = false;                              // the entry names the suffix itself (interval of one)
				if (clean) {
					if (sizeof(idx_t) == 4) {
						uint2 e = ix.qtab32[y];
						tk = (idx_t)e.x; tn = (idx_t)(e.y & 0x07FFFFFFu); tsa = (e.y >> 27) & 1; tlf2 = e.y >> 28;
					} else {
						uint64_t e = ix.qtab64[y];
						tk = (idx_t)(e & 0x3FFFFFFFFull); tn = (idx_t)((e >> 34) & 0x1FFFFFFull); tsa = (e >> 59) & 1; tlf2 = (uint32_t)(e >> 60);
					}
				}
				c_dbg += (dbg == 1 && clean) ? 1u : 0u;
				bool jump = clean && tn != 0;
				k = jump ? tk : l2s[3 - code0] + 1;                            // x[1], :149
				n = jump ? tn : l2s[code0 + 1] - l2s[code0];                  // x[2], :150
				cur = pos + (jump ? q : 1);
				// the reference performs these q-1 steps one by one; keep its block accounting
				c_lf += jump ? (uint32_t)(q - 1) : 0u;
				c_lf2 += jump ? tlf2 : 0u;
				stop = fast ? rlen : stop_pos;
				mode = direct && n == 1 ? 1 : 0;
				if (jump && tsa) { tpos = (idx_t)(ix.seq_len - (uint64_t)tk); mode = 2; }
				active = true;
				c_search++;
				if ((cur >> 4) != wword) { win = wnext; wword++; wnext = KG_WORD(wword + 1); }
			}
			if (__ballot(!done && !active) == 0) break;
		}
		if (__ballot(active) == 0) break;               // every lane is done

		// ================= tight loop =================
		for (;;) {
			int sh = (cur & 15) << 2;
			int code = (int)((win >> sh) & 15);
			// :153-154; otherwise the search is over.  In text-comparison mode the window registers are not kept
			// in step with cur (the comparison loads its own read words, and an ambiguous code simply differs)
			bool alive = active && cur < stop && (mode == 2 || code <= 3);
			bool lf = alive && mode == 0, sa = alive && mode == 1, cmp = alive && mode == 2;
			int c = 3 - code;
			// the gathers of this iteration -- issued together, used below
			idx_t kk = 0, ll = 0;
			uint4 vk = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
			uint64_t sav = 0, t0 = 0, t1 = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
			if (lf) {
				kk = k - 1; ll = k - 1 + n;                                 // bwt_2occ4(x1-1, x1-1+x2), :157
				kk -= (kk >= primary);
				ll -= (ll >= primary);
				vk = ix.planes[(((uint64_t)(kk >> 6)) << 2) + (uint32_t)c];
				vl = ix.planes[(((uint64_t)(ll >> 6)) << 2) + (uint32_t)c];
			}
			if (sa) sav = ix.fsa32 ? (uint64_t)ix.fsa32[k] : ix.fsa64[k];
			if (cmp) {
				const U64u *tp = reinterpret_cast<const U64u *>(ix.text + ((uint64_t)tpos >> 2));
				t0 = tp[0].v; t1 = tp[1].v;                                   // 64 text bases from the byte holding tpos
				int w = cur >> 4;
				r0 = KG_WORD(w); r1 = KG_WORD(w + 1); r2 = KG_WORD(w + 2); r3 = KG_WORD(w + 3);   // 64 read codes from the word holding cur
			}
			bool cont = false;
			if (lf) {
				uint64_t mk = (2ull << (kk & 63)) - 1, ml = (2ull << (ll & 63)) - 1;
				idx_t ok = (idx_t)(((uint64_t)vk.w << 32) | vk.z) + (idx_t)__popcll((((uint64_t)vk.y << 32) | vk.x) & mk);
				idx_t ol = (idx_t)(((uint64_t)vl.w << 32) | vl.z) + (idx_t)__popcll((((uint64_t)vl.y << 32) | vl.x) & ml);
				idx_t nn = ol - ok;
				c_lf++;
				c_lf2 += (kk >> 7) != (ll >> 7) ? 1u : 0u;                 // reference 128-symbol block accounting
				c_dbg += (dbg == 2 || (dbg == 3 && (kk >> 7) != (ll >> 7))) ? 1u : 0u;
				cont = nn != 0;
				if (cont) {
					k = l2s[c] + 1 + ok;
					n = nn;
					cur++;
					mode = direct && nn == 1 ? 1 : 0;
				}
			}
			if (sa) {
				// SA[k] = where the reverse complement of the match starts; the match itself then ends right
				// before text position 2L - SA[k], which is the base the next read base must equal
				tpos = (idx_t)(ix.seq_len - sav);
				c_sa++;
				mode = 2;
				cont = true;
			}
			if (cmp) {
				// 48 bases per round: read codes (one per nibble) against text bases (2 bits each, spread to nibbles);
				// an ambiguous read code (> 3) always differs
				c_dbg += dbg == 4 ? 1u : 0u;
				uint64_t q0 = sh ? (r0 >> sh) | (r1 << (64 - sh)) : r0;
				uint64_t q1 = sh ? (r1 >> sh) | (r2 << (64 - sh)) : r1;
				uint64_t q2 = sh ? (r2 >> sh) | (r3 << (64 - sh)) : r2;
				int ts = ((int)((uint32_t)tpos & 3)) << 1;
				uint64_t ta = ts ? (t0 >> ts) | (t1 << (64 - ts)) : t0;           // text bases 0..31 from tpos
				uint64_t tb = t1 >> ts;                                            // bases 32..(63 - ts/2)
				uint64_t d0 = spread_2bit((uint32_t)ta) ^ q0, d1 = spread_2bit((uint32_t)(ta >> 32)) ^ q1, d2 = spread_2bit((uint32_t)tb) ^ q2;
				int m = d0 ? (__ffsll((unsigned long long)d0) - 1) >> 2
				      : d1 ? 16 + ((__ffsll((unsigned long long)d1) - 1) >> 2)
				      : d2 ? 32 + ((__ffsll((unsigned long long)d2) - 1) >> 2) : 48;
				uint64_t left = ix.seq_len - (uint64_t)tpos;                       // the text ends: the reference's step finds nothing
				m = (uint64_t)m > left ? (int)left : m;
				int lim = stop - cur < 48 ? stop - cur : 48;
				if (m < lim) {
					// the reference extends m times, then either stops at an ambiguous base (no step) or
					// performs the step that empties the interval
					uint64_t qw = m < 16 ? q0 : m < 32 ? q1 : q2;
					int code2 = (int)((qw >> ((m & 15) << 2)) & 15);
					c_lf += (uint32_t)m + (code2 <= 3 ? 1u : 0u);
					cur += m;
					tpos += (idx_t)m;
				} else {
					c_lf += (uint32_t)lim;
					cur += lim;
					tpos += (idx_t)lim;
					cont = cur < stop;
				}
			}
			if (cont && mode != 2 && (cur >> 4) != wword) { win = wnext; wword++; wnext = KG_WORD(wword + 1); }
			pending = pending || (active && !cont);
			active = cont;
			uint64_t parked = __ballot(pending);
			if (__popcll(parked) >= kRefill || __ballot(active) == 0) break;
		}
	}
	// the slots this wave reserved but never filled are marked empty for the locate kernel
	for (unsigned long long x = hit_pool.next + (threadIdx.x & 63); x < hit_pool.end; x += 64) a.hits[x].n = 0;
	// work counters: one atomic per wave
	uint64_t s0 = c_search, s1 = c_lf - c_lf2, s2 = c_lf2, s3 = c_sa;
	for (int off = 32; off > 0; off >>= 1) {
		s0 += __shfl_down(s0, off); s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); s3 += __shfl_down(s3, off);
	}
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&a.counters[0], (unsigned long long)s0);
		atomicAdd(&a.counters[1], (unsigned long long)s1);
		atomicAdd(&a.counters[2], (unsigned long long)s2);
		if (s3) atomicAdd(&a.counters[4], (unsigned long long)s3);
	}
	if (dbg) {
		uint64_t s4 = c_dbg;
		for (int off = 32; off > 0; off >>= 1) s4 += __shfl_down(s4, off);
		if ((threadIdx.x & 63) == 0 && s4) atomicAdd(&a.counters[3], (unsigned long long)s4);
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

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
	return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src);
}

// KG_SA_FULL: one gather per seed.  A wave takes 64 hits at a time; hits that were finished against the text
// carry their position already, the others expand to n seeds each (up to 50): those (hit, i) items are spread
// over the lanes -- prefix sum of n across the wave, binary search by shuffle -- so that a 50-seed hit costs
// its wave one round instead of 50, and the suffix-array reads and seed writes of one hit are contiguous.
__global__ __launch_bounds__(256) void locate_full_kernel(SeedArgs a)
{
	const FmView &ix = a.ix;
	const unsigned long long n_hits = *a.hit_count;
	const int lane = threadIdx.x & 63;
	const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const unsigned long long n_waves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
	uint32_t c_sa = 0;
	for (unsigned long long t0 = wave * 64; t0 < n_hits; t0 += n_waves * 64) {
		Hit h;
		h.n = 0; h.k = 0; h.len = 0; h.rpos = 0; h.direct = 0; h.read = 0; h.seed_start = 0;
		if (t0 + lane < n_hits) h = a.hits[t0 + lane];
		int64_t out = h.n ? hit_out_base(a, h) : 0;
		if (h.n && h.direct && out < a.seed_capacity) {
			kg_seed s;
			s.gPos = (int64_t)h.k; s.rPos = h.rpos; s.len = h.len;
			a.seeds[out] = s;
		}
		int cnt = (h.n && !h.direct) ? h.n : 0;
		int incl = cnt;
		for (int d = 1; d < 64; d <<= 1) {
			int v = __shfl_up(incl, d);
			if (lane >= d) incl += v;
		}
		const int total = __shfl(incl, 63);
		for (int base = 0; base < total; base += 64) {
			int t = base + lane;
			int lo = 0, hi = 63;                                  // smallest lane whose inclusive prefix exceeds t
#pragma unroll
			for (int step = 0; step < 6; ++step) {
				int mid = (lo + hi) >> 1;
				int v = __shfl(incl, mid);
				if (v > t) hi = mid; else lo = mid + 1;
			}
			int src = lo > 63 ? 63 : lo;
			int s_incl = __shfl(incl, src), s_cnt = __shfl(cnt, src);
			uint64_t s_k = shfl_u64(h.k, src);
			int64_t s_out = (int64_t)shfl_u64((uint64_t)out, src);
			int s_len = __shfl(h.len, src), s_rpos = __shfl(h.rpos, src);
			int i = t - (s_incl - s_cnt);
			if (t < total) {
				uint64_t sa = ix.fsa32 ? (uint64_t)ix.fsa32[s_k + (uint64_t)i] : ix.fsa64[s_k + (uint64_t)i];
				c_sa++;
				if (s_out + i < a.seed_capacity) {
					kg_seed s;
					s.gPos = (int64_t)(ix.seq_len - sa - (uint64_t)s_len);
					s.rPos = s_rpos; s.len = s_len;
					a.seeds[s_out + i] = s;
				}
			}
		}
	}
	uint64_t s1 = c_sa;
	for (int off = 32; off > 0; off >>= 1) s1 += __shfl_down(s1, off);
	if (lane == 0 && s1) atomicAdd(&a.counters[4], (unsigned long long)s1);
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

// ---- per-read ordering -----------------------------------------------------------------------------
// Most reads carry 2-4 seeds, reads inside repeat families hundreds (up to 50 per search), long reads
// more.  Sorting a long list from one lane is a chain of dependent global-memory accesses that holds up
// its wave, so the work is split by size:
//   sort_small_kernel : one read per lane; lists of <= 8 are sorted in registers (19-comparator network,
//                       no dependent memory traffic); longer lists go to one of two work lists (one
//                       atomic per wave and list)
//   sort_wave_kernel  : 9..64 seeds: bitonic network across 16, 32 or 64 lanes by shuffle, i.e. four, two or one
//                       list(s) per wave
//   sort_lds_kernel   : > 64 seeds, one read per wave: bitonic in LDS, in two size classes (<= 256 seeds with a
//                       4 KB buffer so that many waves fit a CU, <= kSortLds with 32 KB); Shell sort in
//                       place beyond (rare)
// Both comparators are total orders on the data, so any correct sort reproduces std::sort's result.
constexpr int kSortLds = 2048;

__device__ __forceinline__ kg_seed seed_sentinel()
{
	kg_seed s;
	s.gPos = INT64_MAX; s.rPos = 0; s.len = 0;       // greater than every real seed under both comparators
	return s;
}

// work-list append through a wave-local ticket pool (see pool_take: one atomic per 256 tickets -- a plain
// wave-aggregated atomic per iteration runs into the single-word atomic ceiling); tickets a wave reserved
// but never used are marked -1 by list_close and skipped by the consumers
__device__ __forceinline__ void list_append(int32_t *list, WavePool &pool, unsigned long long *count, bool want, int32_t value)
{
	unsigned long long t = pool_take(pool, count, want, 64);
	if (want) list[t] = value;
}

__device__ __forceinline__ void list_close(int32_t *list, const WavePool &pool)
{
	for (unsigned long long x = pool.next + (threadIdx.x & 63); x < pool.end; x += 64) list[x] = -1;
}

// each of the five work lists has room for every read plus one 64-ticket pool chunk per wave of sort_small_kernel's
// grid (max_hits >= max_reads + 8192 per CU; the lists share the 32-byte hit records' memory: 5 x 4 bytes each)
__device__ __forceinline__ int64_t sort_list_stride(const SeedArgs &a) { return a.max_hits; }

// a seed as the sort sees it: key = PosDiff (FastMode) or gPos (SensitiveMode), then rPos; scalars only, so
// the networks below stay in registers
struct SortItem { int64_t key; int32_t rpos, len; };

__device__ __forceinline__ SortItem sort_item(const kg_seed &s, int mode)
{
	SortItem t;
	t.key = mode == KG_MODE_FAST ? s.gPos - s.rPos : s.gPos;
	t.rpos = s.rPos; t.len = s.len;
	return t;
}

__device__ __forceinline__ kg_seed sort_seed(int64_t key, int32_t rpos, int32_t len, int mode)
{
	kg_seed s;
	s.gPos = mode == KG_MODE_FAST ? key + rpos : key;
	s.rPos = rpos; s.len = len;
	return s;
}

#define KG_LOAD(i)                                                                           \
	int64_t k##i = INT64_MAX; int32_t r##i = 0, l##i = 0;                                    \
	if (n > i) { SortItem t_ = sort_item(s[i], mode); k##i = t_.key; r##i = t_.rpos; l##i = t_.len; }
#define KG_STORE(i) if (n > i) s[i] = sort_seed(k##i, r##i, l##i, mode)
#define KG_CE(i, j)                                                                          \
	do {                                                                                     \
		bool sw_ = k##j < k##i || (k##j == k##i && r##j < r##i);                             \
		int64_t ka_ = sw_ ? k##j : k##i, kb_ = sw_ ? k##i : k##j;                            \
		int32_t ra_ = sw_ ? r##j : r##i, rb_ = sw_ ? r##i : r##j;                            \
		int32_t la_ = sw_ ? l##j : l##i, lb_ = sw_ ? l##i : l##j;                            \
		k##i = ka_; k##j = kb_; r##i = ra_; r##j = rb_; l##i = la_; l##j = lb_;              \
	} while (0)

__global__ __launch_bounds__(256) void sort_small_kernel(SeedArgs a)
{
	int32_t *lists = reinterpret_cast<int32_t *>(a.hits);          // the hit records are dead once located
	const int64_t list_stride = sort_list_stride(a);
	unsigned long long *counts = a.read_queue + 12;
	WavePool pool0, pool1, pool2, pool3, pool4;
	const int mode = a.mode;
	int64_t r0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	int64_t rounds = (a.n_reads + stride - 1) / stride;          // wave-uniform trip count (ballots below)
	for (int64_t it = 0; it < rounds; ++it) {
		int64_t r = r0 + it * stride;
		int64_t lo = 0;
		int n = 0;
		if (r < a.n_reads) {
			lo = a.seed_off[r];
			int64_t hi = a.seed_off[r + 1];
			if (hi > a.seed_capacity) hi = a.seed_capacity;
			n = (int)(hi - lo);
		}
		list_append(lists, pool0, counts + 0, n > 8 && n <= 16, (int32_t)r);
		list_append(lists + list_stride, pool1, counts + 1, n > 16 && n <= 32, (int32_t)r);
		list_append(lists + 2 * list_stride, pool2, counts + 2, n > 32 && n <= 64, (int32_t)r);
		list_append(lists + 3 * list_stride, pool3, counts + 3, n > 64 && n <= 256, (int32_t)r);
		list_append(lists + 4 * list_stride, pool4, counts + 4, n > 256, (int32_t)r);
		if (n < 2 || n > 8) continue;
		kg_seed *s = a.seeds + lo;
		KG_LOAD(0) KG_LOAD(1) KG_LOAD(2) KG_LOAD(3) KG_LOAD(4) KG_LOAD(5) KG_LOAD(6) KG_LOAD(7)
		KG_CE(0, 2); KG_CE(1, 3); KG_CE(4, 6); KG_CE(5, 7);
		KG_CE(0, 4); KG_CE(1, 5); KG_CE(2, 6); KG_CE(3, 7);
		KG_CE(0, 1); KG_CE(2, 3); KG_CE(4, 5); KG_CE(6, 7);
		KG_CE(2, 4); KG_CE(3, 5);
		KG_CE(1, 4); KG_CE(3, 6);
		KG_CE(1, 2); KG_CE(3, 4); KG_CE(5, 6);
		KG_STORE(0); KG_STORE(1); KG_STORE(2); KG_STORE(3); KG_STORE(4); KG_STORE(5); KG_STORE(6); KG_STORE(7);
	}
	list_close(lists, pool0);
	list_close(lists + list_stride, pool1);
	list_close(lists + 2 * list_stride, pool2);
	list_close(lists + 3 * list_stride, pool3);
	list_close(lists + 4 * list_stride, pool4);
}

// kGroup lanes per list: 64 / kGroup lists are sorted side by side in one wave
template <int kGroup, int kList>
__global__ __launch_bounds__(256) void sort_wave_kernel(SeedArgs a)
{
	const int32_t *list = reinterpret_cast<const int32_t *>(a.hits) + (int64_t)kList * sort_list_stride(a);
	const unsigned long long n_list = a.read_queue[12 + kList];
	const int mode = a.mode;
	const int lane = threadIdx.x & 63;
	const int sub = lane & (kGroup - 1), grp = lane / kGroup;
	constexpr int kPer = 64 / kGroup;
	const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const unsigned long long n_waves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
	// 64 list entries per wave at a time (one coalesced load); unused pool tickets are -1
	for (unsigned long long t0 = wave * 64; t0 < n_list; t0 += n_waves * 64) {
		int32_t mine = t0 + lane < n_list ? list[t0 + lane] : -1;
		for (int base = 0; base < 64; base += kPer) {
			int64_t r = __shfl(mine, base + grp);
			if (__ballot(r >= 0) == 0) continue;
			int n = 0;
			kg_seed *s = a.seeds;
			if (r >= 0) {
				int64_t lo = a.seed_off[r], hi = a.seed_off[r + 1];
				if (hi > a.seed_capacity) hi = a.seed_capacity;
				n = (int)(hi - lo);
				s += lo;
			}
			int64_t key = INT64_MAX;
			int32_t rpos = 0, len = 0;
			if (sub < n) { SortItem t_ = sort_item(s[sub], mode); key = t_.key; rpos = t_.rpos; len = t_.len; }
#pragma unroll
			for (int k = 2; k <= kGroup; k <<= 1)
#pragma unroll
				for (int j = k >> 1; j > 0; j >>= 1) {
					int64_t okey = (int64_t)(((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)((uint64_t)key >> 32), j) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)(uint64_t)key, j));
					int32_t orpos = __shfl_xor(rpos, j), olen = __shfl_xor(len, j);
					bool keep_min = ((sub & j) == 0) == ((sub & k) == 0);
					bool o_less = okey < key || (okey == key && orpos < rpos);
					bool take = keep_min == o_less;
					key = take ? okey : key; rpos = take ? orpos : rpos; len = take ? olen : len;
				}
			if (sub < n) s[sub] = sort_seed(key, rpos, len, mode);
		}
	}
}

template <int kCap, int kList>
__global__ __launch_bounds__(64) void sort_lds_kernel(SeedArgs a)
{
	__shared__ kg_seed buf[kCap];
	const int32_t *list = reinterpret_cast<const int32_t *>(a.hits) + (int64_t)kList * sort_list_stride(a);
	const unsigned long long n_list = a.read_queue[12 + kList];
	const int mode = a.mode;
	const int lane = threadIdx.x;
	// one read per wave (= block); 64 list entries at a time, unused pool tickets are -1
	for (unsigned long long t0 = (unsigned long long)blockIdx.x * 64; t0 < n_list; t0 += (unsigned long long)gridDim.x * 64) {
	int32_t mine = t0 + lane < n_list ? list[t0 + lane] : -1;
	for (uint64_t todo = __ballot(mine >= 0); todo; todo &= todo - 1) {
		int64_t r = __shfl(mine, __ffsll((unsigned long long)todo) - 1);
		int64_t lo = a.seed_off[r], hi = a.seed_off[r + 1];
		if (hi > a.seed_capacity) hi = a.seed_capacity;
		int64_t n = hi - lo;
		kg_seed *s = a.seeds + lo;
		if (n <= kCap) {
			int p = 128;
			while (p < n) p <<= 1;
			for (int i = lane; i < p; i += 64) buf[i] = i < n ? s[i] : seed_sentinel();
			__syncthreads();
			for (int k = 2; k <= p; k <<= 1)
				for (int j = k >> 1; j > 0; j >>= 1) {
					for (int i = lane; i < (p >> 1); i += 64) {
						int x = ((i & ~(j - 1)) << 1) | (i & (j - 1));
						int y = x + j;
						kg_seed u = buf[x], w = buf[y];
						bool up = (x & k) == 0;
						if (seed_less(w, u, mode) == up) { buf[x] = w; buf[y] = u; }
					}
					__syncthreads();
				}
			for (int i = lane; i < n; i += 64) s[i] = buf[i];
			__syncthreads();
		} else if (lane == 0) {
			// Shell's passes (Ciura gaps) in place
			for (int64_t gap = 1750; gap > 0; gap = gap == 1750 ? 701 : gap == 701 ? 301 : gap == 301 ? 132 : gap == 132 ? 57 : gap == 57 ? 23 : gap == 23 ? 10 : gap == 10 ? 4 : gap == 4 ? 1 : 0) {
				for (int64_t i = gap; i < n; ++i) {
					kg_seed v = s[i];
					int64_t j = i - gap;
					while (j >= 0 && seed_less(v, s[j], mode)) { s[j + gap] = s[j]; j -= gap; }
					s[j + gap] = v;
				}
			}
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
		a.used[r] = (int32_t)used;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) { a.n_cands[a.n_reads] = 0; a.used[a.n_reads] = 0; }
}

// the candidates and their seeds, packed in read order (what goes back to the host)
__global__ __launch_bounds__(256) void compact_cands_kernel(ChainArgs a)
{
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; r < a.n_reads; r += stride) {
		const int64_t base = a.seed_off[r], co = a.cand_off[r], so = a.cseed_off[r];
		const int nc = a.n_cands[r], ns = a.used[r];
		for (int c = 0; c < nc; ++c) {
			kg_candidate v = a.cands[base + c];
			v.first = so + (v.first - base);
			a.dense_cands[co + c] = v;
		}
		for (int i = 0; i < ns; ++i) a.dense_seeds[so + i] = a.cand_seeds[base + i];
	}
}

hipError_t launch_chain_batch(const ChainArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream);

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
// Table construction, level by level: the entry of a j-base prefix y (first base in the lowest bits) is one
// extension step away from the entry of its first j-1 bases, so level j is built in place from level j-1 --
// thread y reads its parent T[y] and writes the four children T[y | c << 2(j-1)], its own slot (c = 0) last.
// Entries are exact while the table is built ({k, n, lf2} in 16 bytes); levels above kQtabWide are finished
// per entry from the widest exact level and written in the compact search format.
constexpr int kQtabWide = 12;

struct QEntry { uint64_t k, n; uint32_t lf2; };

__device__ __forceinline__ QEntry qentry_unpack(uint4 v)
{
	QEntry e;
	e.k = (uint64_t)v.x | ((uint64_t)(v.z & 0xFF) << 32);
	e.n = (uint64_t)v.y | ((uint64_t)((v.z >> 8) & 0xFF) << 32);
	e.lf2 = v.z >> 16;
	return e;
}

__device__ __forceinline__ uint4 qentry_pack(const QEntry &e)
{
	return make_uint4((uint32_t)e.k, (uint32_t)e.n, (uint32_t)(e.k >> 32) | ((uint32_t)(e.n >> 32) << 8) | (e.lf2 << 16), 0u);
}

__device__ __forceinline__ QEntry qentry_step(const FmView &ix, QEntry e, int read_code)   // one BWT_Search extension, :157-168
{
	if (e.n == 0) return e;
	int c = 3 - read_code;
	uint64_t kk = e.k - 1, ll = e.k - 1 + e.n;
	kk -= (kk >= ix.primary);
	ll -= (ll >= ix.primary);
	e.lf2 += (kk >> 7) != (ll >> 7);
	uint64_t ok = rank_plane(ix, kk, c), ol = rank_plane(ix, ll, c);
	e.n = ol - ok;
	e.k = l2_of(ix, c) + 1 + ok;
	return e;
}

__global__ __launch_bounds__(256) void qtab_level_kernel(FmView ix, uint4 *wide, int level)
{
	uint64_t y = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	if (level == 1) {
		if (y < 4) {
			QEntry e;
			e.k = l2_of(ix, 3 - (int)y) + 1; e.n = l2_of(ix, (int)y + 1) - l2_of(ix, (int)y); e.lf2 = 0;   // :149-150
			wide[y] = qentry_pack(e);
		}
		return;
	}
	for (; y < (1ull << (2 * (level - 1))); y += stride) {
		QEntry parent = qentry_unpack(wide[y]);
		for (int c = 3; c >= 0; --c)
			wide[y | ((uint64_t)c << (2 * (level - 1)))] = qentry_pack(qentry_step(ix, parent, c));
	}
}

__global__ __launch_bounds__(256) void qtab_finish_kernel(FmView ix, const uint4 *wide, int q_wide, int q, uint2 *t32, uint64_t *t64)
{
	uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; id < (1ull << (2 * q)); id += stride) {
		QEntry e = qentry_unpack(wide[id & ((1ull << (2 * q_wide)) - 1)]);
		for (int j = q_wide; j < q; ++j) e = qentry_step(ix, e, (int)((id >> (2 * j)) & 3));
		// an interval of one suffix: store the suffix itself (SA[k]) when the full SA is resident -- the search
		// then goes straight to comparing against the text, without the SA gather
		uint64_t flag = 0;
		if (e.n == 1 && ix.text != nullptr && (ix.fsa32 || ix.fsa64)) {
			e.k = ix.fsa32 ? (uint64_t)ix.fsa32[e.k] : ix.fsa64[e.k];
			flag = 1;
		}
		// not representable in the compact entry: n = 0, the search falls back to single steps
		if (t32) {
			if (e.n >= (1ull << 27)) e.n = 0;
			t32[id] = make_uint2((uint32_t)e.k, (uint32_t)e.n | ((uint32_t)flag << 27) | (e.lf2 << 28));
		} else {
			if (e.n >= (1ull << 25) || e.k >= (1ull << 34)) e.n = 0;
			t64[id] = e.k | (e.n << 34) | (flag << 59) | ((uint64_t)e.lf2 << 60);
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

// the 2L-base text the index was built over, from the forward-strand .pac bases (bwa packing: base i in bits
// (~i&3)*2 of byte i>>2): T[x] = pac[x] for x < L, 3 - pac[2L-1-x] above
__global__ __launch_bounds__(256) void build_text_kernel(const uint8_t *pac, uint64_t l_pac, uint8_t *text, uint64_t n_bytes)
{
	uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; b < n_bytes; b += stride) {
		uint32_t out = 0;
		for (int j = 0; j < 4; ++j) {
			uint64_t x = (b << 2) + (uint64_t)j;
			uint32_t v = 0;
			if (x < l_pac) v = (pac[x >> 2] >> ((~x & 3) << 1)) & 3;
			else if (x < 2 * l_pac) { uint64_t y = 2 * l_pac - 1 - x; v = 3 - ((pac[y >> 2] >> ((~y & 3) << 1)) & 3); }
			out |= v << (j << 1);
		}
		text[b] = (uint8_t)out;
	}
}

hipError_t launch_build_text(const uint8_t *pac, uint64_t l_pac, uint8_t *text, uint64_t n_bytes, hipStream_t stream)
{
	hipLaunchKernelGGL(build_text_kernel, dim3(grid_for((int64_t)n_bytes, 256, 256 * 64)), dim3(256), 0, stream, pac, l_pac, text, n_bytes);
	return hipGetLastError();
}

hipError_t launch_build_qtab(const FmView &ix, int q, uint2 *t32, uint64_t *t64, hipStream_t stream)
{
	int q_wide = q < kQtabWide ? q : kQtabWide;
	uint4 *wide = nullptr;
	hipError_t e = hipMalloc((void **)&wide, sizeof(uint4) << (2 * q_wide));
	if (e != hipSuccess) return e;
	for (int level = 1; level <= q_wide; ++level) {
		int64_t parents = level == 1 ? 4 : (int64_t)1 << (2 * (level - 1));
		hipLaunchKernelGGL(qtab_level_kernel, dim3(grid_for(parents, 256, 256 * 32)), dim3(256), 0, stream, ix, wide, level);
	}
	hipLaunchKernelGGL(qtab_finish_kernel, dim3(256 * 64), dim3(256), 0, stream, ix, wide, q_wide, q, t32, t64);
	e = hipGetLastError();
	hipError_t e2 = hipStreamSynchronize(stream);
	(void)hipFree(wide);
	return e != hipSuccess ? e : e2;
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
	(void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, it, (int64_t *)nullptr, (int)max_reads);   // size query only
	return bytes;
}

hipError_t launch_chain_batch(const ChainArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream)
{
	if (a.n_reads <= 0) return hipSuccess;
	int64_t g = (a.n_reads + 255) / 256;
	if (g > (int64_t)n_cu * 16) g = (int64_t)n_cu * 16;
	hipLaunchKernelGGL(chain_kernel, dim3((unsigned)g), dim3(256), 0, stream, a);
	hipError_t e;
	size_t tb = scan_temp_bytes;
	WideIter it1(a.n_cands, WidenOp());
	if ((e = hipcub::DeviceScan::ExclusiveSum(scan_temp, tb, it1, a.cand_off, (int)(a.n_reads + 1), stream)) != hipSuccess) return e;
	tb = scan_temp_bytes;
	WideIter it2(a.used, WidenOp());
	if ((e = hipcub::DeviceScan::ExclusiveSum(scan_temp, tb, it2, a.cseed_off, (int)(a.n_reads + 1), stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(compact_cands_kernel, dim3((unsigned)g), dim3(256), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_seed_batch(const SeedArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream, hipEvent_t *ev)
{
	if (a.n_reads <= 0) return hipSuccess;
	hipError_t e;
	// queue heads, hit count and counters are zeroed on the stream every call
	hipLaunchKernelGGL(seed_reset_kernel, dim3(1), dim3(64), 0, stream, a.read_queue);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	// persistent lanes: 4 blocks of 256 threads per CU (= 16 waves/CU, all resident: with ~90 VGPRs a SIMD holds 5 waves)
	// unless the batch is smaller.  More blocks change nothing (alternating A/B of 4 / 6 / 8 per CU: 22.0-22.6 ms each on
	// the hg38-sized workload, tools/ab_blocks.sh): the kernel is bound by the fabric, not by the number of waves in flight.
	int per_cu = 4;
	if (const char *env = getenv("KG_SEARCH_BLOCKS_PER_CU")) per_cu = atoi(env) > 0 ? atoi(env) : 4;  // tuning knob
	int blocks = grid_for(a.n_reads, 256, n_cu * per_cu);
	static const bool fused_pack = getenv("KG_FUSED_PACK") != nullptr;      // experiment: no pack pre-pass, raw codes read in the search
	SeedArgs a2 = a;
	if (fused_pack) a2.packed = nullptr;
	else hipLaunchKernelGGL(pack_reads_kernel, dim3(grid_for(a.n_reads * 16, 256, n_cu * 32)), dim3(256), 0, stream, a);
	if (ev) (void)hipEventRecord(ev[0], stream);
	// 32-bit interval arithmetic whenever the text allows it; KG_FORCE_U64 exercises the wide instantiation
	// (the one hg38-sized indexes use) on small test indexes
	static const bool force_wide = getenv("KG_FORCE_U64") != nullptr;
	if (a.ix.seq_len < 0xFFFFFF00ull && !(force_wide && a.ix.qtab64))
		hipLaunchKernelGGL(search_kernel<uint32_t>, dim3(blocks), dim3(256), 0, stream, a2);
	else
		hipLaunchKernelGGL(search_kernel<uint64_t>, dim3(blocks), dim3(256), 0, stream, a2);
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
	hipLaunchKernelGGL(sort_small_kernel, dim3(grid_for(a.n_reads, 256, n_cu * 16)), dim3(256), 0, stream, a);
	hipLaunchKernelGGL((sort_wave_kernel<16, 0>), dim3(n_cu * 8), dim3(256), 0, stream, a);
	hipLaunchKernelGGL((sort_wave_kernel<32, 1>), dim3(n_cu * 8), dim3(256), 0, stream, a);
	hipLaunchKernelGGL((sort_wave_kernel<64, 2>), dim3(n_cu * 8), dim3(256), 0, stream, a);
	hipLaunchKernelGGL((sort_lds_kernel<256, 3>), dim3(n_cu * 32), dim3(64), 0, stream, a);
	hipLaunchKernelGGL((sort_lds_kernel<kSortLds, 4>), dim3(n_cu * 5), dim3(64), 0, stream, a);
	if (ev) (void)hipEventRecord(ev[4], stream);
	return hipGetLastError();
}

}  // namespace kg
