// nw_kernels.hip -- batched Needleman-Wunsch gap closing on gfx950.
//
// Replaces nw_alignment() (reference src/nw_alignment.cpp:18-80) for a batch of fragment pairs.
// The reference computes three float matrices whose values are all multiples of 0.5
// (match +1.5, mismatch -1.5, first gap base -1.5, further gap bases -0.5, boundaries -1-0.5*i,
// :3-6,36-47,53-55), so the DP here is exact 32-bit integer arithmetic on doubled scores:
//   r(i,j) = max(r(i,j-1) - 1, s(i,j-1) - 3)      gap in sequence 1 (consumes a base of sequence 2)
//   t(i,j) = max(t(i-1,j) - 1, s(i-1,j) - 3)      gap in sequence 2
//   s(i,j) = max(s(i-1,j-1) +- 3, r, t)           bases compared through nst_nt4_table (case-blind, N==N)
// and the traceback keeps the reference's tie order (:60-72): s==r first, then s==t, else diagonal.
// Two direction bits per cell (s==r, s==t) are all the traceback needs.  It is a FULL DP (the
// reference is not banded); no MFMA -- this is integer max-plus work, not a contraction.
//
// Work is binned by max(m,n) in a classify pass (ballot-compacted lists):
//   class 0 (<= 8)   one pair per lane, everything in registers, 8x8 cells fully unrolled
//                    (97 % of Illumina gap fragments)
//   class 1 (<= 32)  one pair per lane, score rows in registers, direction words in LDS
//   class 2 (> 32)   one pair per wave: 64 lanes sweep the anti-diagonals of a 64-column stripe,
//                    neighbours exchanged with DPP shuffles, stripe boundaries in LDS, direction
//                    words in a per-wave HBM scratch slab
#include "seed_kernels.hpp"

namespace kg {

constexpr int NEG = -(1 << 20);

struct __attribute__((packed, aligned(1))) NwU64u { uint64_t v; };
struct __attribute__((packed, aligned(1))) NwU32u { uint32_t v; };

__device__ __forceinline__ int nt4_code(unsigned char ch)  // nst_nt4_table, src/BWT_Index/bntseq.c:40-57
{
	unsigned u = ch & 0xDFu;
	return u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
}

// A pair of the batch: offsets/lengths either from the two prefix-sum arrays of the C ABI (kg_nw_batch*) or from a job
// descriptor the alignment stage wrote on the device (align_kernels.hip).  In descriptor mode sequence 2 is read straight
// from the 2-bit text of the index (its codes ARE nst_nt4_table's), sequence 1 from the read characters.
struct NwPair {
	int64_t o1, o2, oo;
	int m, n;
};
__device__ __forceinline__ NwPair nw_pair(const NwArgs &a, int64_t p)
{
	NwPair q;
	if (a.desc) {
		const NwJobDesc d = a.desc[p];
		q.o1 = d.o1; q.o2 = d.o2; q.oo = d.ops; q.m = d.m; q.n = d.n;
	} else {
		q.o1 = a.off1[p]; q.o2 = a.off2[p];
		q.m = (int)(a.off1[p + 1] - q.o1); q.n = (int)(a.off2[p + 1] - q.o2);
		q.oo = q.o1 + q.o2;
	}
	return q;
}
__device__ __forceinline__ int nw_code2(const NwArgs &a, int64_t at)      // code of sequence-2 character `at`
{
	if (a.text2) return (a.text2[(uint64_t)at >> 2] >> (((uint32_t)at & 3) << 1)) & 3;
	return nt4_code((unsigned char)a.f2[at]);
}
__device__ __forceinline__ int64_t nw_count(const NwArgs &a) { return a.n_dev ? (int64_t)min(*a.n_dev, (unsigned long long)a.n) : a.n; }

// value of the lane below; lane 0, which has none, gets `first`: DPP wave shift right by one, no LDS traffic
__device__ __forceinline__ int wave_shr1(int v, int first)
{
	return __builtin_amdgcn_update_dpp(first, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ int lane_rank_nw(uint64_t mask)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__device__ __forceinline__ void append(bool want, unsigned long long *count, int32_t *list, int32_t value)
{
	uint64_t mask = __ballot(want);
	if (mask == 0) return;
	int leader = __ffsll((unsigned long long)mask) - 1;
	unsigned long long base = 0;
	if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(count, (unsigned long long)__popcll(mask));
	base = __shfl(base, leader);
	if (want) list[base + lane_rank_nw(mask)] = value;
}

// counter reset as a kernel: same-queue kernel->kernel ordering is what every later launch relies on anyway
// (a hipMemsetAsync of freshly pool-allocated words was observed to land after the classify kernel)
__global__ void nw_reset_kernel(unsigned long long *queue)
{
	if (threadIdx.x < kNwQueueWords) queue[threadIdx.x] = 0;
}

// queue words: [0..2] list sizes of the three classes, [3] class-2 work queue head, [4] the same for the launch of the longer pairs (two tiers),
// [5] the number of those longer pairs (a fourth list, a.tier_len > 0: they were part of list 2 until round 5, and the launch made for them -- two
// waves per CU -- walked all 720 k entries of a long-read batch's list to find its ~50: 15 of the batch's 90 ms, profiles/r05z_pacbio_kernel_trace)
// Every block takes ONE contiguous range of pairs: it counts its pairs per class, reserves its share of the lists with one atomic each, and fills
// it -- the waves drawing their places from counters in the LDS.  (One global atomic per wave and class, as
// before, was the kernel: 125 k same-address atomics for 8 M tiny pairs = 1.5 ms of the call's 3.4 ms.)
__global__ __launch_bounds__(256) void nw_classify_kernel(NwArgs a)
{
	__shared__ unsigned int s_cnt[4];
	__shared__ unsigned long long s_next[4];
	const int64_t count = nw_count(a);
	const int64_t per = (count + gridDim.x - 1) / gridDim.x;
	const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < count ? b0 + per : count;
	if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const int tier_len = a.tier_len > 0 ? a.tier_len : 0x7fffffff;
	auto cls_of = [&](int64_t p) {
		const NwPair q = nw_pair(a, p);
		const int mx = q.m > q.n ? q.m : q.n;
		return mx <= 8 ? 0 : mx <= 32 ? 1 : mx <= tier_len ? 2 : 3;
	};
	unsigned int mine[4] = {0, 0, 0, 0};
	for (int64_t p = b0 + threadIdx.x; p < b1; p += blockDim.x) mine[cls_of(p)]++;
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		unsigned int v = mine[c];
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
		if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt[c], v);
	}
	__syncthreads();
	if (threadIdx.x < 4) s_next[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(a.queue + (threadIdx.x < 3 ? threadIdx.x : 5), (unsigned long long)s_cnt[threadIdx.x]) : 0;
	__syncthreads();
	for (int64_t base = b0; base < b1; base += blockDim.x) {
		const int64_t p = base + threadIdx.x;
		const int cls = p < b1 ? cls_of(p) : -1;
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const uint64_t mask = __ballot(cls == c);
			if (mask == 0) continue;
			const int leader = __ffsll((unsigned long long)mask) - 1;
			unsigned long long at = 0;
			if ((int)(threadIdx.x & 63) == leader) at = atomicAdd(&s_next[c], (unsigned long long)__popcll(mask));
			at = __shfl(at, leader);
			if (cls == c) a.big_list[(int64_t)c * a.n + (int64_t)at + lane_rank_nw(mask)] = (int32_t)p;
		}
	}
}

// reverse ops[0..len) in place (traceback produces the columns right to left)
__device__ __forceinline__ void reverse_ops(uint8_t *ops, int len)
{
	for (int x = 0, y = len - 1; x < y; ++x, --y) { uint8_t t = ops[x]; ops[x] = ops[y]; ops[y] = t; }
}

// ---- class 0: up to 8x8, registers only -----------------------------------------------------------
__global__ __launch_bounds__(256) void nw_small8_kernel(NwArgs a)
{
	const unsigned long long count = a.queue[0];
	const int32_t *list = a.big_list;
	unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (; t < count; t += stride) {
		int64_t p = list[t];
		const NwPair q = nw_pair(a, p);
		const int64_t o1 = q.o1, o2 = q.o2;
		const int m = q.m, n = q.n;
		// descriptor mode (the alignment stage's jobs): both sequences lie in buffers with slack behind them (the read characters,
		// the 2-bit text), so each is ONE unaligned load instead of up to eight byte gathers
		const bool packed = a.desc != nullptr && a.text2 != nullptr;
		uint64_t w1 = 0, w2 = 0;
		uint32_t tw = 0;
		bool words = packed;                                            // sequence 1 (and, in offset mode, sequence 2) arrived as one word
		if (packed) {
			w1 = reinterpret_cast<const NwU64u *>(a.f1 + o1)->v;
			const uint32_t raw = reinterpret_cast<const NwU32u *>(a.text2 + ((uint64_t)o2 >> 2))->v;
			tw = raw >> (((uint32_t)o2 & 3) << 1);                      // (8 bases = 16 bits, at most 6 bits of shift: 22 bits needed)
		} else if (a.desc == nullptr && a.text2 == nullptr) {
			// offset mode: the same single load wherever eight bytes lie inside the caller's arrays (all pairs but the last few)
			if (o1 + 8 <= a.off1[a.n] && o2 + 8 <= a.off2[a.n]) {
				w1 = reinterpret_cast<const NwU64u *>(a.f1 + o1)->v;
				w2 = reinterpret_cast<const NwU64u *>(a.f2 + o2)->v;
				words = true;
			}
		}
		int c2[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) c2[j] = j < n ? (packed ? (int)((tw >> (2 * j)) & 3u) : words ? nt4_code((unsigned char)(w2 >> (8 * j))) : nw_code2(a, o2 + j)) : 8 + j;
		int S[9], T[9];
		S[0] = 0; T[0] = 0;
#pragma unroll
		for (int j = 1; j <= 8; ++j) { S[j] = -2 - j; T[j] = NEG; }
		uint64_t fr = 0, ft = 0;  // bit 8*(i-1)+(j-1)
#pragma unroll
		for (int i = 1; i <= 8; ++i) {
			if (i <= m) {
				int c1 = nt4_code(words ? (unsigned char)(w1 >> (8 * (i - 1))) : (unsigned char)a.f1[o1 + i - 1]);
				int diag = S[0];
				S[0] = -2 - i;
				int left_s = S[0], left_r = NEG;
#pragma unroll
				for (int j = 1; j <= 8; ++j) {
					int up_s = S[j], up_t = T[j];
					int r = max(left_r - 1, left_s - 3);
					int tt = max(up_t - 1, up_s - 3);
					int d = diag + (c1 == c2[j - 1] ? 3 : -3);
					int s = max(d, max(r, tt));
					fr |= (uint64_t)(s == r) << (8 * (i - 1) + (j - 1));
					ft |= (uint64_t)(s == tt) << (8 * (i - 1) + (j - 1));
					diag = up_s; S[j] = s; T[j] = tt; left_s = s; left_r = r;
				}
			}
		}
		// the traceback yields the columns right to left: they are collected in registers from the top byte of a 16-byte word down
		// (at most m + n <= 16 columns), shifted into place at the end and stored once -- no read-modify-write of the op string
		uint8_t *ops = a.ops + q.oo;
		int i = m, j = n, len = 0;
		uint64_t lo = 0, hi = 0;                                        // bytes 0..7 | 8..15 of the word
		while (i > 0 || j > 0) {
			int bit = 8 * (i - 1) + (j - 1);
			bool g1 = i == 0 || (j > 0 && ((fr >> bit) & 1));
			bool g2 = !g1 && (j == 0 || ((ft >> bit) & 1));
			const uint64_t op = g1 ? KG_OP_GAP1 : g2 ? KG_OP_GAP2 : KG_OP_DIAG;
			const int at = 15 - len;
			if (at >= 8) hi |= op << (8 * (at - 8)); else lo |= op << (8 * at);
			len++;
			if (g1) j--; else if (g2) i--; else { i--; j--; }
		}
		if (len > 0) {
			const int sh = 16 - len;                                    // bytes to shift down
			if (sh >= 8) { lo = sh == 8 ? hi : hi >> (8 * (sh - 8)); hi = 0; }
			else if (sh > 0) { lo = (lo >> (8 * sh)) | (hi << (8 * (8 - sh))); hi >>= 8 * sh; }
			// overlapping stores that never leave [0, len): the neighbouring jobs' op strings start right behind
			if (len >= 8) {
				reinterpret_cast<NwU64u *>(ops)->v = lo;
				if (len > 8) {
					const int t8 = len - 8;                             // the last eight bytes, bytes t8 .. t8 + 7 of the word
					reinterpret_cast<NwU64u *>(ops + t8)->v = t8 == 8 ? hi : (lo >> (8 * t8)) | (hi << (8 * (8 - t8)));
				}
			} else if (len >= 4) {
				reinterpret_cast<NwU32u *>(ops)->v = (uint32_t)lo;
				if (len > 4) reinterpret_cast<NwU32u *>(ops + len - 4)->v = (uint32_t)(lo >> (8 * (len - 4)));
			} else {
				for (int k = 0; k < len; ++k) ops[k] = (uint8_t)(lo >> (8 * k));
			}
		}
		a.aln_len[p] = len;
	}
}

// ---- class 1: up to 32x32, direction words in LDS ---------------------------------------------------
__global__ __launch_bounds__(256) void nw_small32_kernel(NwArgs a)
{
	__shared__ uint2 dirs[32][256];  // [row][thread] -> (s==r bits, s==t bits), 64 KB
	const unsigned long long count = a.queue[1];
	const int32_t *list = a.big_list + a.n;
	unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	for (; t < count; t += stride) {
		int64_t p = list[t];
		const NwPair q = nw_pair(a, p);
		const int64_t o1 = q.o1, o2 = q.o2;
		const int m = q.m, n = q.n;
		// descriptor mode (the alignment stage's / the fragment kernels' jobs): both sequences lie in buffers with slack behind them (the read
		// characters, the 2-bit text), so sequence 1 is four unaligned 8-byte loads and sequence 2 one 8-byte load + a byte instead of up to 32 + 32
		// byte gathers of 64 lanes at 64 different addresses each (round 5: 250 ms per 400 k long reads in product against ~30 ms in tools/bench_nw.py)
		const bool packed = a.desc != nullptr && a.text2 != nullptr;
		uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, tw = 0;
		if (packed) {
			r0 = reinterpret_cast<const NwU64u *>(a.f1 + o1)->v;
			if (m > 8) r1 = reinterpret_cast<const NwU64u *>(a.f1 + o1 + 8)->v;
			if (m > 16) r2 = reinterpret_cast<const NwU64u *>(a.f1 + o1 + 16)->v;
			if (m > 24) r3 = reinterpret_cast<const NwU64u *>(a.f1 + o1 + 24)->v;
			const uint8_t *tp = a.text2 + ((uint64_t)o2 >> 2);
			const uint64_t lo = reinterpret_cast<const NwU64u *>(tp)->v, hi = tp[8];
			const int sh = ((int)o2 & 3) << 1;
			tw = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;              // 32 bases from o2
		}
		int c2[32];
#pragma unroll
		for (int j = 0; j < 32; ++j) c2[j] = j < n ? (packed ? (int)((tw >> (2 * j)) & 3u) : nw_code2(a, o2 + j)) : 8 + j;
		int S[33], T[33];
		S[0] = 0; T[0] = 0;
#pragma unroll
		for (int j = 1; j <= 32; ++j) { S[j] = -2 - j; T[j] = NEG; }
		uint64_t cw = 0;
		for (int i = 1; i <= m; ++i) {
			int c1;
			if (packed) {
				const int k = i - 1;
				if ((k & 7) == 0) cw = k < 8 ? r0 : k < 16 ? r1 : k < 24 ? r2 : r3;
				c1 = nt4_code((unsigned char)(cw >> (8 * (k & 7))));
			} else c1 = nt4_code((unsigned char)a.f1[o1 + i - 1]);
			int diag = S[0];
			S[0] = -2 - i;
			int left_s = S[0], left_r = NEG;
			uint32_t fr = 0, ft = 0;
#pragma unroll
			for (int j = 1; j <= 32; ++j) {
				int up_s = S[j], up_t = T[j];
				int r = max(left_r - 1, left_s - 3);
				int tt = max(up_t - 1, up_s - 3);
				int d = diag + (c1 == c2[j - 1] ? 3 : -3);
				int s = max(d, max(r, tt));
				fr |= (uint32_t)(s == r) << (j - 1);
				ft |= (uint32_t)(s == tt) << (j - 1);
				diag = up_s; S[j] = s; T[j] = tt; left_s = s; left_r = r;
			}
			dirs[i - 1][threadIdx.x] = make_uint2(fr, ft);
		}
		// the traceback yields the columns right to left: one walk counts them, a second writes them where they belong, eight at a time (one
		// byte store per column and a reversal in place -- byte loads and stores again -- was the rest of the kernel's memory traffic)
		uint8_t *ops = a.ops + q.oo;
		int len = 0;
		for (int i = m, j = n; i > 0 || j > 0; ++len) {
			uint2 w = i > 0 ? dirs[i - 1][threadIdx.x] : make_uint2(0, 0);
			bool g1 = i == 0 || (j > 0 && ((w.x >> (j - 1)) & 1));
			bool g2 = !g1 && (j == 0 || ((w.y >> (j - 1)) & 1));
			if (g1) j--; else if (g2) i--; else { i--; j--; }
		}
		{
			int i = m, j = n, at = len, na = 0;
			uint64_t acc = 0;
			while (i > 0 || j > 0) {
				uint2 w = i > 0 ? dirs[i - 1][threadIdx.x] : make_uint2(0, 0);
				bool g1 = i == 0 || (j > 0 && ((w.x >> (j - 1)) & 1));
				bool g2 = !g1 && (j == 0 || ((w.y >> (j - 1)) & 1));
				acc = (acc << 8) | (uint64_t)(g1 ? KG_OP_GAP1 : g2 ? KG_OP_GAP2 : KG_OP_DIAG);
				--at;
				if (++na == 8) { reinterpret_cast<NwU64u *>(ops + at)->v = acc; acc = 0; na = 0; }
				if (g1) j--; else if (g2) i--; else { i--; j--; }
			}
			for (int k = 0; k < na; ++k) ops[k] = (uint8_t)(acc >> (8 * k));
		}
		a.aln_len[p] = len;
	}
}

// ---- class 2: one pair per wave, anti-diagonal sweep -------------------------------------------------
// LDS per wave: boundary column S,R for rows 0..m (column 64*stripe), updated in place: lane 63
// rewrites row i 63 steps after lane 0 consumed it.

// K columns per lane: a stripe is 64 K columns wide, lane l owns columns K l + 1 .. K l + K of it and computes the K cells of its
// row in one step (each takes the one before as its left neighbour) -- the per-step work that does not depend on the number of
// cells (lane shifts, boundary fetch, mask stores, loop) is paid once per 64 K cells, and a fragment of up to 64 K columns needs
// one sweep.  Direction bits: per stripe and step 2 K 64-bit lane masks (column k: s == r, s == t) -- the compare results
// themselves.
template <int K>
__device__ __forceinline__ void nw_sweep(const NwArgs &a, int lane, int m, int n, int64_t o2, int2 *bSR, const unsigned char *s1c, uint64_t *dir64)
{
	constexpr int W = 64 * K;
	const int n_stripes = (n + W - 1) / W;
	const int steps = m + 63;                          // anti-diagonal steps of one (full) stripe
	for (int st = 0; st < n_stripes; ++st) {
		const int j0 = st * W + K * lane + 1;           // 1-based first column of this lane
		int c2[K], up_s[K], up_t[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			c2[k] = j0 + k <= n ? nw_code2(a, o2 + j0 + k - 1) : 9 + k;
			up_s[k] = -2 - (j0 + k); up_t[k] = NEG;     // row 0
		}
		int res_s = 0, res_r = 0;            // this lane's last result of its last column (what lane+1 sees as "left")
		int prev_left_s = 0;                 // S(i-1, j0-1): the left neighbour's value of the previous step
		int c1 = 15;                         // sequence-1 code of this lane's row, handed on from lane to lane
		uint64_t *dw = dir64 + (int64_t)st * steps * (2 * K);
		// lane 0's inputs of a step (boundary values and the code of the row entering the stripe) are fetched one step ahead,
		// so that the LDS latency is off the dependent chain
		int2 b = bSR[m >= 1 ? 1 : 0];
		int code_in = s1c[0];
		// (a stripe narrower than 64 K columns is done once its last column has reached row m)
		const int width = n - st * W < W ? n - st * W : W;
		const int steps_here = m + (width + K - 1) / K - 1;
		for (int d = 1; d <= steps_here; ++d) {
			const int i = d - lane;
			// what comes in from the left: lane-1's result of the previous step (a wave shift by one lane); lane 0 takes the
			// boundary column instead
			const int left_s = wave_shr1(res_s, b.x), left_r = wave_shr1(res_r, b.y);
			c1 = wave_shr1(c1, code_in);
			{
				const int bn = d + 1 <= m ? d + 1 : m;             // next step's boundary row
				b = bSR[bn];
				code_in = s1c[bn - 1];
			}
			const bool valid = (unsigned)(i - 1) < (unsigned)m;
			int sv[K], rv[K], tv[K];
			uint64_t mr[K], mt[K];
#pragma unroll
			for (int k = 0; k < K; ++k) {
				// column k: its left neighbour is the cell just computed (column 0: what came in), its diagonal the previous
				// column's previous row (row 0 while the lane has not started: up_s still holds it)
				const int ls = k == 0 ? left_s : sv[k - 1], lr = k == 0 ? left_r : rv[k - 1];
				const int diag = k == 0 ? (i == 1 ? (j0 == 1 ? 0 : -2 - (j0 - 1)) : prev_left_s) : up_s[k - 1];
				rv[k] = max(lr - 1, ls - 3);
				tv[k] = max(up_t[k] - 1, up_s[k] - 3);
				sv[k] = max(diag + (c1 == c2[k] ? 3 : -3), max(rv[k], tv[k]));
				mr[k] = __builtin_amdgcn_ballot_w64(sv[k] == rv[k]);       // (rows outside 1..m are never visited by the traceback)
				mt[k] = __builtin_amdgcn_ballot_w64(sv[k] == tv[k]);
			}
			if (lane == 0) {
				uint64_t *o = dw + (2 * K) * (d - 1);
#pragma unroll
				for (int k = 0; k < K; ++k) { o[2 * k] = mr[k]; o[2 * k + 1] = mt[k]; }
			}
			prev_left_s = left_s;            // cell (i, j0-1) is the diagonal of the next row
			if (valid) {                     // (a lane that has not started keeps row 0)
#pragma unroll
				for (int k = 0; k < K; ++k) { up_s[k] = sv[k]; up_t[k] = tv[k]; }
			}
			res_s = sv[K - 1]; res_r = rv[K - 1];
			// lane 63 publishes its last column as the next stripe's boundary (row i of column W (st+1))
			if (lane == 63 && valid) bSR[i] = make_int2(sv[K - 1], rv[K - 1]);
		}
		// row 0 of the next boundary column
		if (lane == 63) bSR[0] = make_int2(-2 - (st * W + W), -2 - (st * W + W));
		__syncthreads();
	}
}

// Traceback, wave-cooperative: the path is one dependent chain; all lanes walk the same (uniform) path.  Cell (i, j) of stripe st
// was computed by lane ((j-1) mod 64 K) / K at step i + lane: the masks of the 64 steps below the current one are loaded by the 64
// lanes at once and read lane by lane (v_readlane on scalar lane / bit indices); every move lowers the step by one or two (or keeps
// it, between columns of one lane).  Returns the number of ops written (in reverse order).
template <int K>
__device__ __forceinline__ int nw_walk(int lane, int m, int n, const uint64_t *dir64, uint8_t *ops)
{
	constexpr int W = 64 * K;
	const int steps = m + 63;
	int len = 0, i = m, jj = n;
	while (i > 0 || jj > 0) {
		if (i == 0) {                                         // the rest of the top row: gaps in sequence 1
			for (int x = lane; x < jj; x += 64) ops[len + x] = KG_OP_GAP1;
			len += jj; jj = 0;
			break;
		}
		if (jj == 0) {
			for (int x = lane; x < i; x += 64) ops[len + x] = KG_OP_GAP2;
			len += i; i = 0;
			break;
		}
		const int st = (jj - 1) / W;
		const int d_hi = i + ((jj - 1) % W) / K;
		const uint64_t *dw = dir64 + (int64_t)st * steps * (2 * K);
		uint64_t cr[K], ct[K];
#pragma unroll
		for (int k = 0; k < K; ++k) { cr[k] = 0; ct[k] = 0; }
		if (d_hi - lane >= 1) {
			const uint64_t *o = dw + (2 * K) * (d_hi - lane - 1);
#pragma unroll
			for (int k = 0; k < K; ++k) { cr[k] = o[2 * k]; ct[k] = o[2 * k + 1]; }
		}
		while (i > 0 && jj > 0 && (jj - 1) / W == st) {
			const int col = (jj - 1) % W, cl = col / K, d = i + cl;
			const int src = d_hi - d;
			if (src > 63) break;
			const int us = __builtin_amdgcn_readfirstlane(src), ub = __builtin_amdgcn_readfirstlane(cl), kk = __builtin_amdgcn_readfirstlane(col % K);
			uint64_t wr = cr[0], wt = ct[0];
#pragma unroll
			for (int k = 1; k < K; ++k) { if (kk == k) { wr = cr[k]; wt = ct[k]; } }
			const uint32_t half_r = ub < 32 ? (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wr, us) : (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wr >> 32), us);
			const uint32_t half_t = ub < 32 ? (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wt, us) : (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wt >> 32), us);
			const bool g1 = (half_r >> (ub & 31)) & 1;
			const bool g2 = !g1 && ((half_t >> (ub & 31)) & 1);
			if (lane == 0) ops[len] = g1 ? KG_OP_GAP1 : g2 ? KG_OP_GAP2 : KG_OP_DIAG;
			len++;
			if (g1) jj--; else if (g2) i--; else { i--; jj--; }
		}
	}
	return len;
}

// kGlobal: fragments longer than kNwMaxLen -- the boundary column and the sequence-1 codes no longer fit the LDS and
// live in a per-wave HBM slab behind the direction words instead (same sweep; the reference's nw_alignment has no length
// limit, src/nw_alignment.cpp:24-33, so neither has this path).
// tier: 0 = every pair of the class; 1 = the pairs up to a.tier_len (list 2, slabs a.t1_*); 2 = the longer ones (list 3)
template <bool kGlobal>
__global__ __launch_bounds__(64) void nw_big_kernel(NwArgs a, int tier)
{
	extern __shared__ int lds_dyn[];
	const int lane = threadIdx.x;
	const unsigned long long count = a.queue[tier == 2 ? 5 : 2];
	const int32_t *list = a.big_list + (tier == 2 ? 3 : 2) * a.n;
	uint32_t *dir = tier == 1 ? a.t1_dir_scratch + (int64_t)blockIdx.x * a.t1_dir_words_per_wave : a.dir_scratch + (int64_t)blockIdx.x * a.dir_words_per_wave;
	unsigned long long *const ticket = a.queue + (tier == 2 ? 4 : 3);
	// tickets are drawn up to 16 at a time where the list is long: the counter is ONE address for the whole device, ~12 ns per atomic whoever asks --
	// 700 k pairs of a long-read batch took 8.4 of the launch's 9.1 ms to hand out one by one (profiles/r05q_pacbio_kernel_stats.csv, r05w)
	const unsigned long long share = count / ((unsigned long long)gridDim.x * 4ull);
	const unsigned long long take = share < 1 ? 1ull : share > 16 ? 16ull : share;
	unsigned long long t_next = 0, t_end = 0;
	for (;;) {
		if (t_next == t_end) {
			unsigned long long t0 = 0;
			if (lane == 0) t0 = atomicAdd(ticket, take);
			t0 = __shfl(t0, 0);
			if (t0 >= count) break;
			t_next = t0;
			t_end = t0 + take < count ? t0 + take : count;
		}
		const unsigned long long t = t_next++;
		int64_t p = list[t];
		const NwPair q = nw_pair(a, p);
		const int64_t o1 = q.o1, o2 = q.o2;
		const int m = q.m, n = q.n;
		int *lds = kGlobal ? reinterpret_cast<int *>(dir + a.gb_offset_words) : lds_dyn;
		int2 *bSR = reinterpret_cast<int2 *>(lds);                                   // boundary column: {S, R} of rows 0..m
		unsigned char *s1c = reinterpret_cast<unsigned char *>(lds + 2 * (m + 1));  // fits: see nw_big_lds_bytes()
		for (int i = lane; i <= m; i += 64) bSR[i] = i == 0 ? make_int2(0, 0) : make_int2(-2 - i, NEG);
		for (int i = lane; i < m; i += 64) s1c[i] = (unsigned char)nt4_code((unsigned char)a.f1[o1 + i]);
		__syncthreads();
		uint64_t *dir64 = reinterpret_cast<uint64_t *>(dir);
		uint8_t *ops = a.ops + q.oo;
		int len;
		// two columns per lane up to 128 columns (one sweep), four beyond
		if (n <= 128) { nw_sweep<2>(a, lane, m, n, o2, bSR, s1c, dir64); len = nw_walk<2>(lane, m, n, dir64, ops); }
		else { nw_sweep<4>(a, lane, m, n, o2, bSR, s1c, dir64); len = nw_walk<4>(lane, m, n, dir64, ops); }
		if (lane == 0) a.aln_len[p] = len;
		__threadfence_block();
		__syncthreads();
		for (int x = lane; x < len / 2; x += 64) {
			uint8_t t0 = ops[x], t1 = ops[len - 1 - x];
			ops[x] = t1; ops[len - 1 - x] = t0;
		}
		__syncthreads();
	}
}

static inline int grid_for_nw(int64_t items, int block, int max_blocks)
{
	int64_t g = (items + block - 1) / block;
	if (g < 1) g = 1;
	if (g > max_blocks) g = max_blocks;
	return (int)g;
}

hipError_t launch_nw_batch(const NwArgs &a, int n_cu, hipStream_t stream)
{
	if (a.n <= 0) return hipSuccess;
	hipError_t e;
	hipLaunchKernelGGL(nw_reset_kernel, dim3(1), dim3(64), 0, stream, a.queue);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	hipLaunchKernelGGL(nw_classify_kernel, dim3(grid_for_nw(a.n, 256, n_cu * 8)), dim3(256), 0, stream, a);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	hipLaunchKernelGGL(nw_small8_kernel, dim3(grid_for_nw(a.n, 256, n_cu * 8)), dim3(256), 0, stream, a);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	hipLaunchKernelGGL(nw_small32_kernel, dim3(grid_for_nw(a.n, 256, n_cu * 2)), dim3(256), 0, stream, a);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	if (a.tier_len > 0 && a.t1_dir_scratch && a.t1_waves > 0) {
		// the pairs up to tier_len first: a launch sized for them (nw_big_kernel<false>'s dynamic LDS limit stays what the other launch asked for)
		hipLaunchKernelGGL(nw_big_kernel<false>, dim3(a.t1_waves), dim3(64), a.t1_lds_bytes, stream, a, 1);
		if ((e = hipGetLastError()) != hipSuccess) return e;
	}
	const int rest = a.tier_len > 0 ? 2 : 0;
	if (a.dir_scratch && a.big_waves > 0) {
		if (a.gb_offset_words > 0) {
			hipLaunchKernelGGL(nw_big_kernel<true>, dim3(a.big_waves), dim3(64), 0, stream, a, rest);
			return hipGetLastError();
		}
		if (a.big_lds_bytes > 48 * 1024 &&
		    (e = hipFuncSetAttribute(reinterpret_cast<const void *>(nw_big_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, a.big_lds_bytes)) != hipSuccess)
			return e;
		hipLaunchKernelGGL(nw_big_kernel<false>, dim3(a.big_waves), dim3(64), a.big_lds_bytes, stream, a, rest);
	}
	return hipGetLastError();
}

}  // namespace kg
