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

__device__ __forceinline__ int code_at(const uint8_t *enc, int64_t base, int i, int rlen)
{
	// positions past the read end behave as ambiguous bases: the reference's SensitiveMode can ask
	// for them after an N run, where it reads past its own buffer
	return i < rlen ? (int)enc[base + i] : 4;
}

__global__ __launch_bounds__(256) void search_kernel(SeedArgs a)
{
	const FmView &ix = a.ix;
	// per-lane read state
	bool have_read = false, done = false, active = false;
	int64_t r = 0, base = 0;
	int rlen = 0, pos = 0, stop_pos = 0, end_pos = 0, seed_cnt = 0;
	// per-lane search state
	int cur = 0, stop = 0, next_code = 4;
	uint64_t k = 0, n = 0;
	// local work counters
	uint32_t c_search = 0, c_lf1 = 0, c_lf2 = 0;
	WavePool read_pool, hit_pool;

	for (;;) {
		// ---- phase A: lanes without a live interval start their next search / next read --------
		for (;;) {
			bool idle = !done && !active;
			if (__ballot(idle) == 0) break;
			bool want_read = idle && !have_read;
			unsigned long long t = pool_take(read_pool, a.read_queue, want_read);
			if (want_read) {
				if (t >= (unsigned long long)a.n_reads) done = true;
				else {
					r = (int64_t)t;
					base = a.read_off[r];
					rlen = (int)(a.read_off[r + 1] - base);
					pos = 0; stop_pos = 30; end_pos = rlen - a.min_seed_len; seed_cnt = 0;
					have_read = true;
				}
			} else if (idle) {
				// skip ambiguous bases (FastMode :59, SensitiveMode :142)
				while (pos < end_pos && a.enc[base + pos] > 3) { pos++; stop_pos++; }
				if (pos >= end_pos) {
					a.seeds_per_read[r] = seed_cnt;
					have_read = false;
				} else {
					int p = a.enc[base + pos];
					k = l2_of(ix, 3 - p) + 1;                 // x[1] of BWT_Search :149
					n = l2_of(ix, p + 1) - l2_of(ix, p);      // x[2] :150
					cur = pos + 1;
					stop = a.mode == KG_MODE_FAST ? rlen : stop_pos;
					next_code = cur < stop ? code_at(a.enc, base, cur, rlen) : 4;
					active = true;
					c_search++;
				}
			}
		}
		if (__ballot(!done) == 0) break;

		// ---- phase B: one extension step for every live interval -------------------------------
		bool ended = false, hit = false;
		int len = 0;
		if (active) {
			ended = true;
			if (cur < stop && next_code <= 3) {
				int c = 3 - next_code;
				uint32_t pat = (uint32_t)c * 0x55555555u;
				uint64_t kk = k - 1, ll = k - 1 + n;              // bwt_2occ4(x1-1, x1-1+x2) :157
				kk -= (kk >= ix.primary);
				ll -= (ll >= ix.primary);
				uint64_t bk = kk >> 7, bl = ll >> 7;
				OccBlock B = load_block(ix, bk, c);
				int pre = cur + 1 < stop ? code_at(a.enc, base, cur + 1, rlen) : 4;  // overlaps the gathers
				uint64_t ok, ol;
				if (bk == bl) {
					uint32_t n1, n2;
					count_head2(B, pat, (int)(kk & 127) + 1, (int)(ll & 127) + 1, n1, n2);
					ok = B.cnt + n1; ol = B.cnt + n2;
					c_lf1++;
				} else {
					OccBlock B2 = load_block(ix, bl, c);
					ok = B.cnt + count_head(B, pat, (int)(kk & 127) + 1);
					ol = B2.cnt + count_head(B2, pat, (int)(ll & 127) + 1);
					c_lf2++;
				}
				uint64_t nn = ol - ok;
				if (nn != 0) {
					k = l2_of(ix, c) + 1 + ok;
					n = nn;
					cur++;
					next_code = pre;
					ended = false;
				}
			}
			if (ended) {
				len = cur - pos;
				hit = len >= a.min_seed_len && n <= (uint64_t)a.occ_thr;
			}
		}
		// hit list: slots come from the wave's pool (called by the whole wave: ballot is convergent)
		unsigned long long slot = pool_take(hit_pool, a.hit_count, hit);
		if (hit) {
			Hit h;
			h.k = k; h.read = (int32_t)r; h.rpos = pos; h.len = len; h.n = (int32_t)n; h.seed_start = seed_cnt; h.pad = 0;
			a.hits[slot] = h;
			seed_cnt += (int)n;
		}
		if (ended) {
			if (a.mode == KG_MODE_FAST) pos += len + 1;               // :74
			else {                                                     // :157-163
				int adv = hit ? len : a.min_seed_len;
				pos += adv; stop_pos += adv;
				if (stop_pos > rlen) stop_pos = rlen;
			}
			active = false;
		}
	}
	// the slots this wave reserved but never filled are marked empty for the locate kernel
	for (unsigned long long x = hit_pool.next + (threadIdx.x & 63); x < hit_pool.end; x += 64) a.hits[x].n = 0;
	// work counters: one atomic per wave
	uint64_t s0 = c_search, s1 = c_lf1, s2 = c_lf2;
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
				k = lf_step(ix, k);
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
	int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; r < a.n_reads; r += stride) {
		int64_t lo = a.seed_off[r], hi = a.seed_off[r + 1];
		if (hi > a.seed_capacity) hi = a.seed_capacity;
		for (int64_t i = lo + 1; i < hi; ++i) {
			kg_seed v = a.seeds[i];
			int64_t j = i - 1;
			while (j >= lo && seed_less(v, a.seeds[j], a.mode)) { a.seeds[j + 1] = a.seeds[j]; --j; }
			a.seeds[j + 1] = v;
		}
	}
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
			k = lf_step(ix, k);
			p--;
			if ((k & 31) == 0) break;
		}
	}
}

static inline int grid_for(int64_t items, int block, int max_blocks)
{
	int64_t g = (items + block - 1) / block;
	if (g < 1) g = 1;
	if (g > max_blocks) g = max_blocks;
	return (int)g;
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
	if ((e = hipMemsetAsync(a.read_queue, 0, sizeof(unsigned long long) * kCtlWords, stream)) != hipSuccess) return e;
	// persistent lanes: 8 blocks of 256 threads per CU (= 32 waves/CU) unless the batch is smaller
	int blocks = grid_for(a.n_reads, 256, n_cu * 8);
	if (ev) (void)hipEventRecord(ev[0], stream);
	hipLaunchKernelGGL(search_kernel, dim3(blocks), dim3(256), 0, stream, a);
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
