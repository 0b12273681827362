// stream_kernels.hip -- FASTQ text in, SAM text out, on the device (gfx950).
//
// Input side = GetNextChunk / GetNextEntry (reference src/GetData.cpp:29-143) for a whole window of text at once:
//   fq_count_kernel / fq_index_kernel   the lines of the window: every '\n' ends one (getline); byte-parallel, 16 bytes per lane,
//                                       per-tile counts -> scan -> scatter of the line ends in order.
//   fq_record_kernel                    four lines = one record (header, sequence, '+', qualities) with the reference's own
//                                       arithmetic: name = IdentifyHeaderBegPos/EndPos (:29-49), rlen = sequence line length - 1
//                                       (the last character of a line is taken to be the newline, :66-69), qualities cut to rlen.
//   fq_plan_kernel                      how many whole chunks of ReadChunkSize reads the window yields (GetNextChunk's loop, :109-143),
//                                       where the next window starts, and whether something the device path does not take lies ahead
//                                       (an empty read ends a chunk early in the reference; such windows go to the host's reader).
//   fq_materialise_kernel               the reads as the reference holds them: characters as in the file, the second read of a
//                                       pair reverse-complemented (:125-135; GetComplementarySeq, src/tools.cpp:3-29).
// Output side = OutputPairedAlignments / OutputSingledAlignments (src/Mapping.cpp:177-315) for every kg_aln_record of the batch:
//   sam_size_kernel                     exact byte count of the line(s) of every read -> scan -> offsets
//   sam_format_kernel                   one wave per 64 reads: every lane prints the numeric fields of one read into the LDS, then all
//                                       lanes move each read's pieces -- name, FLAG .. TLEN, sequence (reverse-complemented for a
//                                       record shown on the other strand), qualities (reversed likewise), NM / AS / XS.
// Byte work, HBM-bound, no MFMA.  Everything is integer / character arithmetic; results are bit-identical to the host pipeline's
// text (tests/test_stream_gpu.py, and every SAM parity test runs through this path).
#include "stream_kernels.hpp"

#include <hipcub/hipcub.hpp>

namespace kg {

namespace {

// 4-bit mask of the bytes of x equal to the byte replicated in pat (bit i = byte i, lowest address first)
__device__ __forceinline__ uint32_t eq_mask4(uint32_t x, uint32_t pat)
{
	uint32_t t = x ^ pat;
	uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);       // 0x80 in every zero byte of t, exactly
	return ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u);
}

// the 16 bytes of this thread: newline mask (and whether a NUL byte lies among them), restricted to the window
__device__ __forceinline__ uint32_t newline_mask16(const FqWindow &w, int64_t p, bool &nul)
{
	nul = false;
	if (p >= w.end || p + 16 <= w.begin) return 0;
	const uint4 v = *reinterpret_cast<const uint4 *>(w.text + p);
	uint32_t nl = eq_mask4(v.x, 0x0A0A0A0Au) | (eq_mask4(v.y, 0x0A0A0A0Au) << 4) | (eq_mask4(v.z, 0x0A0A0A0Au) << 8) | (eq_mask4(v.w, 0x0A0A0A0Au) << 12);
	uint32_t z = eq_mask4(v.x, 0u) | (eq_mask4(v.y, 0u) << 4) | (eq_mask4(v.z, 0u) << 8) | (eq_mask4(v.w, 0u) << 12);
	int lo = (int)(w.begin > p ? w.begin - p : 0), hi = (int)(w.end - p < 16 ? w.end - p : 16);
	uint32_t keep = (hi >= 16 ? 0xFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
	nul = (z & keep) != 0;
	return nl & keep;
}

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
	const int lane = threadIdx.x & 63;
	for (int d = 1; d < 64; d <<= 1) {
		int t = __shfl_up(v, d);
		if (lane >= d) v += t;
	}
	return v;
}

}  // namespace

// sixteen bytes at any address (one unaligned load / store)
struct __attribute__((packed, aligned(1))) SamU128 { uint64_t lo, hi; };

// ---- lines ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fq_count_kernel(FqArgs a, int f, int64_t n_tiles)
{
	const FqWindow &w = a.w[f];
	const int64_t ta = w.begin & ~(int64_t)15;
	__shared__ int wave_sum[4];
	for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
		bool nul;
		uint32_t m = newline_mask16(w, ta + t * kFqTile + (int64_t)threadIdx.x * 16, nul);
		if (nul) a.meta[FQM_NUL] = 1;
		int c = __popc(m);
		for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
		if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
		__syncthreads();
		if (threadIdx.x == 0) w.tile_lines[t] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
		__syncthreads();
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) w.tile_lines[n_tiles] = 0;
}

__global__ __launch_bounds__(256) void fq_index_kernel(FqArgs a, int f, int64_t n_tiles)
{
	const FqWindow &w = a.w[f];
	const int64_t ta = w.begin & ~(int64_t)15;
	__shared__ int wave_sum[4];
	for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
		bool nul;
		const int64_t p = ta + t * kFqTile + (int64_t)threadIdx.x * 16;
		uint32_t m = newline_mask16(w, p, nul);
		int c = __popc(m);
		int incl = wave_inclusive_scan(c);
		if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
		__syncthreads();
		int64_t at = (int64_t)w.tile_lines[t] + (incl - c);
		for (int q = 0; q < (int)(threadIdx.x >> 6); ++q) at += wave_sum[q];
		while (m) {
			int b = __ffs(m) - 1;
			m &= m - 1;
			if (at < w.line_capacity) w.line_end[at] = (uint32_t)(p + b + 1);
			else a.meta[FQM_OVERFLOW] = 1;
			at++;
		}
		__syncthreads();
	}
	// the total, and the last line of a file that does not end in a newline (getline returns it as it is)
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		int64_t lines = w.tile_lines[n_tiles];
		if (w.eof && w.end > w.begin && w.text[w.end - 1] != '\n') {
			if (lines < w.line_capacity) w.line_end[lines] = (uint32_t)w.end;
			else a.meta[FQM_OVERFLOW] = 1;
			lines++;
		}
		a.meta[FQM_LINES0 + f] = lines;
	}
}

// ---- records -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fq_record_kernel(FqArgs a, int f)
{
	const FqWindow &w = a.w[f];
	// (launched behind fq_index_kernel of the same window on the same stream: the line count is final)
	int64_t lines = a.meta[FQM_LINES0 + f];
	if (lines > w.line_capacity) lines = w.line_capacity;
	const int64_t recs = lines >> 2;
	for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < recs; j += (int64_t)gridDim.x * blockDim.x) {
		const uint32_t l0 = j == 0 ? (uint32_t)w.begin : w.line_end[4 * j - 1], l1 = w.line_end[4 * j], l2 = w.line_end[4 * j + 1],
		               l3 = w.line_end[4 * j + 2], l4 = w.line_end[4 * j + 3];
		// IdentifyHeaderBegPos / IdentifyHeaderEndPos on the header line (its newline included), src/GetData.cpp:29-49
		const int len = (int)(l1 - l0);
		const uint8_t *h = w.text + l0;
		int p1 = len - 1, p2 = len - 1;
		bool f1 = false, f2 = false;
		// (sixteen characters per load -- a header is two or three of them -- instead of a byte load per character, each at a different memory line
		//  per lane; the window has 4 KB of slack behind its last byte)
		for (int base = 0; base < len && !(f1 && f2); base += 16) {
			const SamU128 u = *reinterpret_cast<const SamU128 *>(h + base);
			const uint32_t x0 = (uint32_t)u.lo, x1 = (uint32_t)(u.lo >> 32), x2 = (uint32_t)u.hi, x3 = (uint32_t)(u.hi >> 32);
			auto mask16 = [&](uint32_t pat) { return eq_mask4(x0, pat) | (eq_mask4(x1, pat) << 4) | (eq_mask4(x2, pat) << 8) | (eq_mask4(x3, pat) << 12); };
			const int lo = base == 0 ? 1 : 0, hi = len - base < 16 ? len - base : 16;              // characters 1 .. len - 1 of the line
			const uint32_t valid = (hi >= 16 ? 0xFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
			const uint32_t other = ~(mask16(0x3E3E3E3Eu) | mask16(0x40404040u)) & valid;          // not '>' and not '@'
			const uint32_t sep = (mask16(0x20202020u) | mask16(0x2F2F2F2Fu) | mask16(0x09090909u)) & valid;   // ' ', '/', '\t'
			if (!f1 && other) { p1 = base + __ffs((int)other) - 1; f1 = true; }
			if (!f2 && sep) { p2 = base + __ffs((int)sep) - 1; f2 = true; }
		}
		const int name_len = p2 > p1 ? p2 - p1 : 0;
		const int rlen = (int)(l2 - l1) - 1;
		int qlen = (int)(l4 - l3);
		if (qlen > rlen) qlen = rlen;
		if (qlen < 0) qlen = 0;
		w.rec_hdr[j] = l0;
		w.rec_name[j] = (uint32_t)(p1 & 0xFFFF) | ((uint32_t)name_len << 16);
		w.rec_seq[j] = l1;
		w.rec_qual[j] = l3;
		w.rec_rlen[j] = rlen;
		w.rec_qlen[j] = qlen;
		// an empty read ends a chunk early in the reference (src/GetData.cpp:117,121); a header too long for the 16-bit fields,
		// a read beyond any short-read length: not taken here
		bool bad = rlen <= 0 || rlen > (1 << 20) || p1 > 0xFFFF || name_len > 0xFFFF;
		// the text of a gz file: gzgets() hands out at most 999 bytes per call (a longer line arrives in pieces that the reference takes for the
		// record's next lines), and an entry whose first line does not start with '@' / '>' or names nothing ends after that line
		// (src/GetData.cpp:152-162): such a record is the caller's line reader's
		if (a.gz_lines) bad = bad || l1 - l0 > 999u || l2 - l1 > 999u || l3 - l2 > 999u || l4 - l3 > 999u || (h[0] != '@' && h[0] != '>') || name_len == 0;
		if (bad) atomicMin((long long *)&a.meta[FQM_BAD0 + f], (long long)j);
		const int64_t i = a.two_files ? 2 * j + f : j;
		if (i < a.max_reads) a.read_len[i] = rlen > 0 ? rlen : 0;
	}
}

__global__ void fq_reset_kernel(FqArgs a)
{
	const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i0 < FQM_WORDS) a.meta[i0] = (i0 == FQM_BAD0 || i0 == FQM_BAD1) ? 0x7fffffffffffffffll : 0;
	for (int64_t i = i0; i <= a.max_reads; i += (int64_t)gridDim.x * blockDim.x) a.read_len[i] = 0;
}

// GetNextChunk's loop over the window: whole chunks only, unless the files end here in a regular way
__global__ void fq_plan_kernel(FqArgs a)
{
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	const int64_t kNone = 0x7fffffffffffffffll;
	const int64_t L0 = a.meta[FQM_LINES0], L1 = a.two_files ? a.meta[FQM_LINES1] : 0;
	const int64_t recs0 = L0 >> 2, recs1 = L1 >> 2;
	int64_t avail, bad = kNone;
	if (a.two_files) {
		avail = 2 * (recs0 < recs1 ? recs0 : recs1);
		if (a.meta[FQM_BAD0] != kNone) bad = 2 * a.meta[FQM_BAD0];
		if (a.meta[FQM_BAD1] != kNone && 2 * a.meta[FQM_BAD1] + 1 < bad) bad = 2 * a.meta[FQM_BAD1] + 1;
	} else {
		avail = recs0;
		bad = a.meta[FQM_BAD0];
	}
	const bool at_eof = a.w[0].eof && (!a.two_files || a.w[1].eof);
	const bool regular_end = at_eof && (L0 & 3) == 0 && (!a.two_files || ((L1 & 3) == 0 && recs0 == recs1)) && (!(a.paired && !a.two_files) || (recs0 & 1) == 0);
	int64_t take = avail < a.want_reads ? avail : a.want_reads;
	int64_t stop = FQ_STOP_NONE;
	if (bad < take) { take = bad; stop = FQ_STOP_IRREGULAR; }
	if (a.meta[FQM_NUL] || a.meta[FQM_OVERFLOW]) { take = 0; stop = FQ_STOP_IRREGULAR; }
	const bool full = stop == FQ_STOP_NONE && take == avail && regular_end;
	if (!full) {
		take = take / a.chunk_reads * a.chunk_reads;
		// the rest of the file is less than a chunk and does not end in the regular way (a lone mate, a partial record): the host's
		// reader takes it from here
		if (stop == FQ_STOP_NONE && at_eof && avail - take < a.chunk_reads) stop = FQ_STOP_TAIL;
	}
	const int64_t t0 = a.two_files ? take >> 1 : take, t1 = a.two_files ? take >> 1 : 0;
	a.meta[FQM_READS] = take;
	a.meta[FQM_CHUNKS] = (take + a.chunk_reads - 1) / a.chunk_reads;
	a.meta[FQM_BASES] = a.read_off[take];
	a.meta[FQM_USED0] = t0 ? (int64_t)a.w[0].line_end[4 * t0 - 1] : a.w[0].begin;
	a.meta[FQM_USED1] = a.two_files ? (t1 ? (int64_t)a.w[1].line_end[4 * t1 - 1] : a.w[1].begin) : 0;
	a.meta[FQM_STOP] = stop;
	a.meta[FQM_DONE] = full ? 1 : 0;
}

// GetComplementaryBase, src/tools.cpp:3-17
__device__ __forceinline__ uint8_t comp_char(uint8_t c)
{
	const uint8_t u = c & 0xDFu;
	return u == 'A' ? 'T' : u == 'C' ? 'G' : u == 'G' ? 'C' : u == 'T' ? 'A' : 'N';
}

// ---- one line by ONE lane, 16 bytes per load and store (round 5).  Rounds 3-4 moved a line with all 64 lanes, a byte per lane and instruction: 31 bytes per
// store instruction, the largest kernel of a run; assembling lines in the LDS first (the verdict's suggestion) was slower still -- 41 ms against 31 ms per 20 M
// reads at 4 KB, 61 ms at 16 KB: fewer waves per CU for the same byte-wide loads (profiles/r05t_ab_sam_buf.log).  Here every lane copies its OWN read's line
// piece by piece with unaligned 16-byte accesses: a wave instruction moves 1 KB, and the 64 lines of a wave are consecutive in the output.  A piece's last
// partial word is copied whole where the line still has room behind it (the next piece overwrites the surplus), byte by byte at the line's end.

__device__ __forceinline__ uint64_t comp8(uint64_t x)          // GetComplementaryBase (src/tools.cpp:3-17) on eight characters
{
	const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull;
	auto is = [&](uint64_t c) { const uint64_t t = u ^ (c * 0x0101010101010101ull); return ((~(((t & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | t | 0x7F7F7F7F7F7F7F7Full)) >> 7) * 0xFFull; };
	const uint64_t mA = is(0x41), mC = is(0x43), mG = is(0x47), mT = is(0x54);
	return (mA & 0x5454545454545454ull) | (mC & 0x4747474747474747ull) | (mG & 0x4343434343434343ull) | (mT & 0x4141414141414141ull) | (~(mA | mC | mG | mT) & 0x4E4E4E4E4E4E4E4Eull);
}

// n bytes of src to p; `end` = the end of the line (writing up to 15 bytes past the piece is fine below it)
__device__ __forceinline__ uint8_t *lane_copy(uint8_t *p, const uint8_t *end, const uint8_t *src, int n)
{
	int k = 0;
	for (; k + 16 <= n; k += 16) *reinterpret_cast<SamU128 *>(p + k) = *reinterpret_cast<const SamU128 *>(src + k);
	if (k < n) {
		if (p + k + 16 <= end) *reinterpret_cast<SamU128 *>(p + k) = *reinterpret_cast<const SamU128 *>(src + k);
		else for (; k < n; ++k) p[k] = src[k];
	}
	return p + n;
}

// the same with the source read backwards (qualities reversed; kComp: the reverse complement of the bases, GetComplementarySeq src/tools.cpp:19-29)
template <bool kComp>
__device__ __forceinline__ uint8_t *lane_copy_reversed(uint8_t *p, const uint8_t *src, int n)
{
	int k = 0;
	for (; k + 16 <= n; k += 16) {
		const SamU128 v = *reinterpret_cast<const SamU128 *>(src + n - 16 - k);
		SamU128 o;
		o.lo = __builtin_bswap64(v.hi); o.hi = __builtin_bswap64(v.lo);
		if (kComp) { o.lo = comp8(o.lo); o.hi = comp8(o.hi); }
		*reinterpret_cast<SamU128 *>(p + k) = o;
	}
	for (; k < n; ++k) p[k] = kComp ? comp_char(src[n - 1 - k]) : src[n - 1 - k];
	return p + n;
}

// ---- the long pieces (bases, qualities) by EIGHT lanes per line: lane `sub` of the eight moves bytes [16 sub + 128 k, 16 sub + 128 k + 16) -- the eight cover 128
// consecutive bytes per step.  With a line per lane the 64 lanes of a load touched 64 different 128-byte lines for 16 bytes each, and the lines did not
// stay in the L2 until the lane came back for the next 16: the counters showed 242 GB per 100 M reads moved by sam_format_kernel for 76 GB of text, 181 GB
// by fq_materialise_kernel for 31 GB (profiles/r05zj_bench_pmc_summary.json).  Exact: nothing is written outside [p, p + n).
__device__ __forceinline__ void group_copy(uint8_t *p, const uint8_t *src, int n, int sub)
{
	for (int off = sub << 4; off < n; off += 128) {
		if (off + 16 <= n) *reinterpret_cast<SamU128 *>(p + off) = *reinterpret_cast<const SamU128 *>(src + off);
		else for (int k = off; k < n; ++k) p[k] = src[k];
	}
}

template <bool kComp>
__device__ __forceinline__ void group_copy_reversed(uint8_t *p, const uint8_t *src, int n, int sub)
{
	for (int off = sub << 4; off < n; off += 128) {
		if (off + 16 <= n) {
			const SamU128 v = *reinterpret_cast<const SamU128 *>(src + n - 16 - off);
			SamU128 o;
			o.lo = __builtin_bswap64(v.hi); o.hi = __builtin_bswap64(v.lo);
			if (kComp) { o.lo = comp8(o.lo); o.hi = comp8(o.hi); }
			*reinterpret_cast<SamU128 *>(p + off) = o;
		} else for (int k = off; k < n; ++k) p[k] = kComp ? comp_char(src[n - 1 - k]) : src[n - 1 - k];
	}
}

// EIGHT lanes per read (round 5; a wave per read and a byte per lane in rounds 3-4, then a lane per read): its characters into the batch's character
// array, reverse-complemented for the second read of a pair -- 16 bytes per lane, 128 consecutive bytes per group and step (group_copy above)
__global__ __launch_bounds__(256) void fq_materialise_kernel(FqArgs a)
{
	const int sub = threadIdx.x & 7;
	for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < a.n_reads; i += ((int64_t)gridDim.x * blockDim.x) >> 3) {
		const int f = a.two_files ? (int)(i & 1) : 0;
		const int64_t j = a.two_files ? i >> 1 : i;
		const FqWindow &w = a.w[f];
		const uint8_t *src = w.text + w.rec_seq[j];
		const int n = w.rec_rlen[j];
		uint8_t *dst = a.enc + a.read_off[i];
		if (a.paired && (i & 1)) group_copy_reversed<true>(dst, src, n, sub);
		else group_copy(dst, src, n, sub);
	}
}

// ---- SAM text ------------------------------------------------------------------------------------------------------------------
namespace {

// characters "%d" / "%lld" print.  Every number of a SAM line fits 32 bits in practice (positions are contig-relative, contig
// lengths are 32-bit in the index format); 32-bit division by the constant 10 is a multiply, a 64-bit one a subroutine.
__device__ __forceinline__ int u32_chars(uint32_t u)
{
	return u < 10u ? 1 : u < 100u ? 2 : u < 1000u ? 3 : u < 10000u ? 4 : u < 100000u ? 5 : u < 1000000u ? 6 : u < 10000000u ? 7 : u < 100000000u ? 8 : u < 1000000000u ? 9 : 10;
}
__device__ __forceinline__ int int_chars(long long v)
{
	if (v >= -2147483647ll && v <= 2147483647ll) {
		const int x = (int)v;
		return x < 0 ? 1 + u32_chars((uint32_t)(-x)) : u32_chars((uint32_t)x);
	}
	int n = v < 0 ? 2 : 1;
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	while (u >= 10) { u /= 10; n++; }
	return n;
}

__device__ __forceinline__ char *put_int(char *p, long long v)
{
	if (v >= -2147483647ll && v <= 2147483647ll) {
		const int x = (int)v;
		uint32_t u = x < 0 ? (uint32_t)(-x) : (uint32_t)x;
		if (x < 0) *p++ = '-';
		const int n = u32_chars(u);
		for (int i = n - 1; i >= 0; --i) { p[i] = (char)('0' + (int)(u % 10u)); u /= 10u; }
		return p + n;
	}
	const int n = int_chars(v);
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	if (v < 0) p[0] = '-';
	for (int i = n - 1; i >= (v < 0 ? 1 : 0); --i) { p[i] = (char)('0' + (int)(u % 10)); u /= 10; }
	return p + n;
}

__device__ __forceinline__ char *put_lit(char *p, const char *s, int n)
{
	for (int i = 0; i < n; ++i) p[i] = s[i];
	return p + n;
}

struct ReadText {                      // where the name and the qualities of read r lie, and its length
	const uint8_t *name, *qual;
	int name_len, qlen, rlen;
	bool held_reversed;                // the second read of a pair: held reverse-complemented, its qualities reversed
};

__device__ __forceinline__ ReadText read_text(const SamArgs &a, int64_t r)
{
	const int f = a.two_files ? (int)(r & 1) : 0;
	const int64_t j = a.two_files ? r >> 1 : r;
	const FqWindow &w = a.w[f];
	ReadText t;
	const uint32_t nm = w.rec_name[j];
	t.name = w.text + w.rec_hdr[j] + (nm & 0xFFFFu);
	t.name_len = (int)(nm >> 16);
	t.qual = w.text + w.rec_qual[j];
	t.qlen = w.rec_qlen[j];
	t.rlen = (int)(a.read_off[r + 1] - a.read_off[r]);
	t.held_reversed = a.paired && (r & 1);
	return t;
}

#define SAM_UNMAPPED_MID "\t*\t0\t0\t*\t*\t0\t0\t"
#define SAM_UNMAPPED_TAIL "\tAS:i:0\tXS:i:0\n"

// the bytes of one record's line (src/Mapping.cpp:181-186 / 225-230 unmapped, :199-223 / 246-262 / 297-302 mapped)
__device__ int record_size(const SamArgs &a, const ReadText &t, const kg_aln_record &rec)
{
	if (rec.kind == KG_ALN_UNMAPPED)
		return t.name_len + 1 + int_chars(rec.flag) + (int)(sizeof(SAM_UNMAPPED_MID) - 1) + t.rlen + 1 + t.qlen + (int)(sizeof(SAM_UNMAPPED_TAIL) - 1);
	if (rec.kind != KG_ALN_MAPPED) return 0;
	int n = t.name_len + 1 + int_chars(rec.flag) + 1 + (a.chr_name_off[rec.chr + 1] - a.chr_name_off[rec.chr]) + 1 + int_chars(rec.pos) + 1 + int_chars(rec.mapq) + 1 + rec.cigar_len;
	n += rec.has_mate ? 3 + int_chars(rec.mate_pos) + 1 + int_chars(rec.tlen) + 1 : 7;
	n += t.rlen + 1 + t.qlen;
	n += 6 + int_chars(t.rlen - rec.score) + 6 + int_chars(rec.score) + 6 + int_chars(rec.sub_score) + 1;
	return n;
}

}  // namespace

__global__ __launch_bounds__(256) void sam_size_kernel(SamArgs a)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (int64_t)gridDim.x * blockDim.x) {
		const kg_aln_record &first = a.records[r];
		int n = 0;
		if (first.kind == KG_ALN_HOST) {
			a.host_list[atomicAdd(&a.ctl[0], 1ull)] = (int32_t)r;
		} else {
			const ReadText t = read_text(a, r);
			for (int64_t at = r; at >= 0; at = a.records[at].next) n += record_size(a, t, a.records[at]);
		}
		a.sam_len[r] = n;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) a.sam_len[a.n_reads] = 0;
}

// One wave per 64 consecutive reads, two phases.
//   1. lane l prints the numeric fields of read base + l ("\tFLAG\t", "\tPOS\tMAPQ\tCIGAR\t=\tPNEXT\tTLEN\t", the tags) into its own slot
//      of the wave's LDS block and leaves there what the copy phase needs (where the name, the qualities, the characters and the
//      contig name lie, where the line goes): 64 records are fetched and printed side by side, not one after the other.
//   2. for each of the 64 reads all lanes move the pieces -- name, contig name, sequence and qualities straight from the FASTQ
//      text / the read characters -- to the output; everything this phase needs per read comes from the LDS.
// A read with further records chained behind it (-m) takes the record-by-record path (format_chain).
namespace {

constexpr int kFmtA = 16, kFmtB = 96, kFmtT = 64, kFmtSlot = kFmtA + kFmtB + kFmtT;     // bytes of a lane's strings (every string 16-byte aligned: they are read back 16 bytes at a time)

struct FmtDesc {                       // what phase 2 needs for one read
	const uint8_t *name, *qual, *seq, *chr;
	uint8_t *out;
	int32_t room_end_lo;               // (low bits of the end offset: checked against the bytes written)
	int16_t name_len, n_chr, rlen, qlen;
	uint8_t nA, nB, nT, flags;         // flags: 1 print, 2 reverse-complement, 4 qualities reversed, 8 chained records follow
};

// the fields of one record as text: A = "\tFLAG[\t]", B = the middle, T = the tags
__device__ __forceinline__ void print_fields(const ReadText &t, const kg_aln_record &rec, char *A, char *B, char *T, int &nA, int &nB, int &nT)
{
	const bool mapped = rec.kind == KG_ALN_MAPPED;
	char *q = A;
	*q++ = '\t'; q = put_int(q, rec.flag);
	if (mapped) *q++ = '\t';
	nA = (int)(q - A);
	q = B;
	if (mapped) {
		*q++ = '\t'; q = put_int(q, rec.pos); *q++ = '\t'; q = put_int(q, rec.mapq); *q++ = '\t';
		for (int i = 0; i < rec.cigar_len; ++i) *q++ = rec.cigar[i];
		if (rec.has_mate) { q = put_lit(q, "\t=\t", 3); q = put_int(q, rec.mate_pos); *q++ = '\t'; q = put_int(q, rec.tlen); *q++ = '\t'; }
		else q = put_lit(q, "\t*\t0\t0\t", 7);
	} else q = put_lit(q, SAM_UNMAPPED_MID, (int)(sizeof(SAM_UNMAPPED_MID) - 1));
	nB = (int)(q - B);
	q = T;
	if (mapped) {
		q = put_lit(q, "\tNM:i:", 6); q = put_int(q, t.rlen - rec.score);
		q = put_lit(q, "\tAS:i:", 6); q = put_int(q, rec.score);
		q = put_lit(q, "\tXS:i:", 6); q = put_int(q, rec.sub_score);
		*q++ = '\n';
	} else q = put_lit(q, SAM_UNMAPPED_TAIL, (int)(sizeof(SAM_UNMAPPED_TAIL) - 1));
	nT = (int)(q - T);
}

// all lanes: one line from its pieces; returns the end
__device__ __forceinline__ uint8_t *copy_line(uint8_t *__restrict__ p, int lane, const uint8_t *__restrict__ name, int name_len, const char *A, int nA,
                                              const uint8_t *__restrict__ chr, int n_chr, const char *B, int nB, const uint8_t *__restrict__ seq, int rlen, bool flip,
                                              const uint8_t *__restrict__ qual, int qlen, bool qrev, const char *T, int nT)
{
	// (every piece is at most a few wave-widths long: the loads of all pieces are issued before the first store is needed)
	for (int k = lane; k < name_len; k += 64) p[k] = name[k];
	p += name_len;
	if (lane < nA) p[lane] = (uint8_t)A[lane];
	p += nA;
	for (int k = lane; k < n_chr; k += 64) p[k] = chr[k];
	p += n_chr;
	for (int k = lane; k < nB; k += 64) p[k] = (uint8_t)B[k];
	p += nB;
	// the read as the record shows it: as held, or its reverse complement (GetComplementarySeq) with the qualities reversed
	if (flip) { for (int k = lane; k < rlen; k += 64) p[k] = comp_char(seq[rlen - 1 - k]); }
	else { for (int k = lane; k < rlen; k += 64) p[k] = seq[k]; }
	p += rlen;
	if (lane == 0) *p = '\t';
	p += 1;
	if (qrev) { for (int k = lane; k < qlen; k += 64) p[k] = qual[qlen - 1 - k]; }
	else { for (int k = lane; k < qlen; k += 64) p[k] = qual[k]; }
	p += qlen;
	if (lane < nT) p[lane] = (uint8_t)T[lane];
	return p + nT;
}

}  // namespace

__global__ __launch_bounds__(64) void sam_format_kernel(SamArgs a)
{
	__shared__ __attribute__((aligned(16))) char str[64 * kFmtSlot];
	__shared__ FmtDesc desc[64];
	__shared__ int chain_len[3];
	__shared__ int chunk_pre[65];          // phase 2b: chunks of the lines before line i
	const int lane = threadIdx.x;
	const int64_t n_groups = (a.n_reads + 63) >> 6;
	for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
		// ---- phase 1: lane = read -------------------------------------------------------------------------------------------
		{
			const int64_t r = (g << 6) + lane;
			FmtDesc d;
			d.flags = 0;
			if (r < a.n_reads) {
				const kg_aln_record &rec = a.records[r];
				const int kind = rec.kind;
				if (kind == KG_ALN_UNMAPPED || kind == KG_ALN_MAPPED || (kind != KG_ALN_HOST && rec.next >= 0)) {
					const ReadText t = read_text(a, r);
					const int64_t o0 = a.sam_off[r], o1 = a.sam_off[r + 1];
					d.name = t.name; d.qual = t.qual; d.seq = a.enc + a.read_off[r];
					d.out = a.sam + o0;
					d.room_end_lo = (int32_t)(o1 - o0);
					d.name_len = (int16_t)t.name_len; d.rlen = (int16_t)t.rlen; d.qlen = (int16_t)t.qlen;
					d.chr = a.chr_names; d.n_chr = 0;
					const bool mapped = kind == KG_ALN_MAPPED;
					if (mapped) { d.chr = a.chr_names + a.chr_name_off[rec.chr]; d.n_chr = (int16_t)(a.chr_name_off[rec.chr + 1] - a.chr_name_off[rec.chr]); }
					const bool flip = mapped && rec.flip;
					int nA = 0, nB = 0, nT = 0;
					// (reads beyond what the 16-bit fields and the output buffer hold take the record-by-record path, which checks)
					const bool wide = t.rlen > 32000 || t.name_len > 32000 || o1 > a.sam_capacity;
					const bool chained = rec.next >= 0 || wide || (kind != KG_ALN_UNMAPPED && kind != KG_ALN_MAPPED);
					if (!chained) print_fields(t, rec, str + lane * kFmtSlot, str + lane * kFmtSlot + kFmtA, str + lane * kFmtSlot + kFmtA + kFmtB, nA, nB, nT);
					d.nA = (uint8_t)nA; d.nB = (uint8_t)nB; d.nT = (uint8_t)nT;
					d.flags = (uint8_t)(1 | (flip ? 2 : 0) | (flip != t.held_reversed ? 4 : 0) | (chained ? 8 : 0));
				}
			}
			desc[lane] = d;
		}
		__syncthreads();
		// ---- phase 2: every lane its own line (the strings it printed in phase 1 lie in its LDS slot) ------------------------------
		{
			const FmtDesc d = desc[lane];
			if ((d.flags & 1) && !(d.flags & 8)) {
				const uint8_t *S = reinterpret_cast<const uint8_t *>(str) + lane * kFmtSlot;
				uint8_t *p = d.out;
				const uint8_t *const end = d.out + d.room_end_lo;
				uint8_t *const bases = d.out + d.name_len + d.nA + d.n_chr + d.nB;      // (from here on other lanes write, phase 2b: no surplus past it)
				p = lane_copy(p, bases, d.name, d.name_len);
				p = lane_copy(p, bases, S, d.nA);
				p = lane_copy(p, bases, d.chr, d.n_chr);
				p = lane_copy(p, bases, S + kFmtA, d.nB);
				p += d.rlen;
				*p++ = '\t';
				p += d.qlen;
				p = lane_copy(p, end, S + kFmtA + kFmtB, d.nT);
				if (p != end) atomicAdd(&a.ctl[1], 1ull);
			}
		}
		// ---- phase 2b: the bases and the qualities of the 64 lines as ONE list of 16-byte chunks (a line's bases first, then its qualities; 20 chunks for a
		//      150-base read), lane l taking chunks l, l + 64, ...: consecutive lanes move consecutive chunks of a line -- whole 128-byte lines of memory per
		//      instruction on either side, every lane busy in every step.  The read as the record shows it: as held, or its reverse complement
		//      (GetComplementarySeq) with the qualities reversed.  (A line per lane moved the same bytes with the same number of instructions but touched
		//      64 different memory lines per instruction; eight or sixteen lanes per line left lanes idle: 83 ms against 77 per 100 M reads.)
		{
			const FmtDesc &dm = desc[lane];
			const bool on = (dm.flags & 1) && !(dm.flags & 8);
			const int mine = on ? ((dm.rlen + 15) >> 4) + ((dm.qlen + 15) >> 4) : 0;
			const int incl = wave_inclusive_scan(mine);
			if (lane == 0) chunk_pre[0] = 0;
			chunk_pre[lane + 1] = incl;
		}
		__syncthreads();
		{
			const int total = chunk_pre[64];
			int line = 0;
			for (int idx = lane; idx < total; idx += 64) {
				while (idx >= chunk_pre[line + 1]) ++line;
				const FmtDesc &d = desc[line];
				const int c = idx - chunk_pre[line];
				const int n_seq = (d.rlen + 15) >> 4;
				const bool is_seq = c < n_seq;
				const int off = (is_seq ? c : c - n_seq) << 4, n = is_seq ? d.rlen : d.qlen;
				const uint8_t *src = is_seq ? d.seq : d.qual;
				uint8_t *dst = d.out + d.name_len + d.nA + d.n_chr + d.nB + (is_seq ? 0 : d.rlen + 1) + off;
				const bool rev = is_seq ? (d.flags & 2) != 0 : (d.flags & 4) != 0;
				if (off + 16 <= n) {
					SamU128 v = *reinterpret_cast<const SamU128 *>(rev ? src + n - 16 - off : src + off);
					if (rev) {
						const uint64_t lo = __builtin_bswap64(v.hi), hi = __builtin_bswap64(v.lo);
						v.lo = lo; v.hi = hi;
						if (is_seq) { v.lo = comp8(v.lo); v.hi = comp8(v.hi); }
					}
					*reinterpret_cast<SamU128 *>(dst) = v;
				} else {
					for (int k = off; k < n; ++k) {
						const uint8_t b = rev ? src[n - 1 - k] : src[k];
						dst[k - off] = rev && is_seq ? comp_char(b) : b;
					}
				}
			}
		}
		__syncthreads();
		// ... and, all lanes per read, the rare ones with chained records (-m) or beyond the 16-bit fields
		for (int i = 0; i < 64;) {
			if (!(desc[i].flags & 1) || !(desc[i].flags & 8)) { ++i; continue; }
			// -m: every record chained behind the read's own, one after the other (lane 0 prints the fields of each)
			const int64_t r = (g << 6) + i;
			const ReadText t = read_text(a, r);
			const uint8_t *const seq = a.enc + a.read_off[r];
			uint8_t *p = a.sam + a.sam_off[r];
			const int64_t room = a.sam_off[r + 1];
			if (room > a.sam_capacity) { if (lane == 0) atomicAdd(&a.ctl[1], 1ull); ++i; continue; }
			char *S = str + i * kFmtSlot;                    // (the read's own slot is free: nothing was printed into it)
			for (int64_t at = r; at >= 0; at = a.records[at].next) {
				const kg_aln_record &rec = a.records[at];
				if (rec.kind != KG_ALN_UNMAPPED && rec.kind != KG_ALN_MAPPED) continue;
				const bool mapped = rec.kind == KG_ALN_MAPPED;
				__syncthreads();
				if (lane == 0) print_fields(t, rec, S, S + kFmtA, S + kFmtA + kFmtB, chain_len[0], chain_len[1], chain_len[2]);
				__syncthreads();
				const uint8_t *cn = a.chr_names;
				int nc = 0;
				if (mapped) { cn = a.chr_names + a.chr_name_off[rec.chr]; nc = a.chr_name_off[rec.chr + 1] - a.chr_name_off[rec.chr]; }
				const bool flip = mapped && rec.flip;
				p = copy_line(p, lane, t.name, t.name_len, S, chain_len[0], cn, nc, S + kFmtA, chain_len[1], seq, t.rlen, flip, t.qual, t.qlen, flip != t.held_reversed,
				              S + kFmtA + kFmtB, chain_len[2]);
			}
			if (lane == 0 && (int64_t)(p - a.sam) != room) atomicAdd(&a.ctl[1], 1ull);
			++i;
		}
		__syncthreads();
	}
}

__global__ void sam_reset_kernel(SamArgs a)
{
	if (threadIdx.x < 4) a.ctl[threadIdx.x] = 0;
}

// Measurement aid (KG_STREAM_CHECKSUM, bench.py's gpu_pipeline leg): the batch's SAM text summed on the device -- ctl[2] += the sum of its bytes,
// ctl[3] += its line feeds -- so that a run whose text is never copied into file pages still shows WHICH text it made (both figures are
// independent of how the text is cut into batches).  16 bytes per lane and step, one pair of atomics per wave.
__global__ __launch_bounds__(256) void sam_checksum_kernel(SamArgs a)
{
	const int64_t bytes = a.sam_off[a.n_reads];
	const int64_t n16 = bytes >> 4;
	unsigned long long sum = 0, lines = 0;
	const uint4 *src = reinterpret_cast<const uint4 *>(a.sam);
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
		const uint4 v = src[i];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			sum += (w[k] & 255u) + ((w[k] >> 8) & 255u) + ((w[k] >> 16) & 255u) + (w[k] >> 24);
			const uint32_t x = w[k] ^ 0x0A0A0A0Au;                                   // a zero byte where the text holds '\n'
			lines += __popc(((x - 0x01010101u) & ~x) & 0x80808080u);
		}
	}
	if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) {
		const uint8_t c = a.sam[(n16 << 4) + threadIdx.x];
		sum += c; lines += c == 10 ? 1 : 0;
	}
	for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off); lines += __shfl_down(lines, off); }
	if ((threadIdx.x & 63) == 0 && (sum | lines)) { atomicAdd(&a.ctl[2], sum); atomicAdd(&a.ctl[3], lines); }
}

// ---- launches ------------------------------------------------------------------------------------------------------------------
namespace {

struct Widen32 {
	__host__ __device__ __forceinline__ int64_t operator()(const int32_t &x) const { return (int64_t)x; }
};
using Wide32Iter = hipcub::TransformInputIterator<int64_t, Widen32, const int32_t *>;

inline int grid_of(int64_t items, int block, int max_blocks)
{
	int64_t g = (items + block - 1) / block;
	if (g < 1) g = 1;
	if (g > max_blocks) g = max_blocks;
	return (int)g;
}

}  // namespace

size_t fq_scan_temp_bytes(int64_t max_items)
{
	size_t b1 = 0, b2 = 0;
	Wide32Iter it((const int32_t *)nullptr, Widen32());
	(void)hipcub::DeviceScan::ExclusiveSum(nullptr, b1, it, (int64_t *)nullptr, (int)max_items);
	(void)hipcub::DeviceScan::ExclusiveSum(nullptr, b2, (const int32_t *)nullptr, (int32_t *)nullptr, (int)max_items);
	return b1 > b2 ? b1 : b2;
}

hipError_t launch_fq_parse(const FqArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream)
{
	hipError_t e;
	kt_begin(KT_FQ_PARSE, stream);
	hipLaunchKernelGGL(fq_reset_kernel, dim3(grid_of(a.max_reads + 1, 256, n_cu * 8)), dim3(256), 0, stream, a);
	for (int f = 0; f < (a.two_files ? 2 : 1); ++f) {
		const FqWindow &w = a.w[f];
		const int64_t ta = w.begin & ~(int64_t)15;
		const int64_t n_tiles = w.end > ta ? (w.end - ta + kFqTile - 1) / kFqTile : 0;
		hipLaunchKernelGGL(fq_count_kernel, dim3(grid_of(n_tiles, 1, n_cu * 64)), dim3(256), 0, stream, a, f, n_tiles);
		size_t tb = scan_temp_bytes;
		if ((e = hipcub::DeviceScan::ExclusiveSum(scan_temp, tb, (const int32_t *)w.tile_lines, w.tile_lines, (int)(n_tiles + 1), stream)) != hipSuccess) return e;
		hipLaunchKernelGGL(fq_index_kernel, dim3(grid_of(n_tiles, 1, n_cu * 64)), dim3(256), 0, stream, a, f, n_tiles);
		hipLaunchKernelGGL(fq_record_kernel, dim3(grid_of(w.line_capacity / 4, 256, n_cu * 16)), dim3(256), 0, stream, a, f);
	}
	size_t tb = scan_temp_bytes;
	Wide32Iter it(a.read_len, Widen32());
	if ((e = hipcub::DeviceScan::ExclusiveSum(scan_temp, tb, it, a.read_off, (int)(a.max_reads + 1), stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(fq_plan_kernel, dim3(1), dim3(64), 0, stream, a);
	kt_end(KT_FQ_PARSE, stream);
	return hipGetLastError();
}

hipError_t launch_fq_materialise(const FqArgs &a, int n_cu, hipStream_t stream)
{
	if (a.n_reads <= 0) return hipSuccess;
	kt_begin(KT_FQ_MATERIALISE, stream);
	hipLaunchKernelGGL(fq_materialise_kernel, dim3(grid_of(a.n_reads * 8, 256, n_cu * 16)), dim3(256), 0, stream, a);
	kt_end(KT_FQ_MATERIALISE, stream);
	return hipGetLastError();
}

// ---- grouped seeding (abi_stream.hip): a lane's batch as one segment of its group's batch ------------------------------------------
// slot i of the segment: offset of the read in the GROUP's character array (the lane's part starts at enc_base) and its length;
// the slots behind the n reads are empty
__global__ __launch_bounds__(256) void group_publish_kernel(const int64_t *local_off, int64_t n, int64_t slots, int64_t enc_base, int64_t *g_off, int32_t *g_len)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += stride) {
		const int64_t at = local_off[i < n ? i : n];
		g_off[i] = enc_base + at;
		g_len[i] = i < n ? (int32_t)(local_off[i + 1] - at) : 0;
	}
}

// the lane's seed offsets out of the group's: seed_off[i] = g_seed_off[i] - first, i = 0 .. n (the slot behind a segment's last read
// holds the segment's end: empty slots have no seeds)
__global__ __launch_bounds__(256) void group_rebase_kernel(const int64_t *g_seed_off, int64_t n, int64_t first, int64_t *seed_off)
{
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) seed_off[i] = g_seed_off[i] - first;
}

hipError_t launch_group_publish(const int64_t *local_off, int64_t n, int64_t slots, int64_t enc_base, int64_t *g_off, int32_t *g_len, int n_cu, hipStream_t stream)
{
	hipLaunchKernelGGL(group_publish_kernel, dim3(grid_of(slots, 256, n_cu * 8)), dim3(256), 0, stream, local_off, n, slots, enc_base, g_off, g_len);
	return hipGetLastError();
}

hipError_t launch_group_rebase(const int64_t *g_seed_off, int64_t n, int64_t first, int64_t *seed_off, int n_cu, hipStream_t stream)
{
	hipLaunchKernelGGL(group_rebase_kernel, dim3(grid_of(n + 1, 256, n_cu * 8)), dim3(256), 0, stream, g_seed_off, n, first, seed_off);
	return hipGetLastError();
}

hipError_t launch_sam_size(const SamArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream)
{
	kt_begin(KT_SAM_SIZE, stream);
	hipLaunchKernelGGL(sam_reset_kernel, dim3(1), dim3(64), 0, stream, a);
	hipLaunchKernelGGL(sam_size_kernel, dim3(grid_of(a.n_reads, 256, n_cu * 16)), dim3(256), 0, stream, a);
	size_t tb = scan_temp_bytes;
	Wide32Iter it(a.sam_len, Widen32());
	hipError_t e = hipcub::DeviceScan::ExclusiveSum(scan_temp, tb, it, a.sam_off, (int)(a.n_reads + 1), stream);
	if (e != hipSuccess) return e;
	kt_end(KT_SAM_SIZE, stream);
	return hipGetLastError();
}

hipError_t launch_sam_format(const SamArgs &a, int n_cu, hipStream_t stream)
{
	if (a.n_reads <= 0) return hipSuccess;
	kt_begin(KT_SAM_FORMAT, stream);
	hipLaunchKernelGGL(sam_format_kernel, dim3(grid_of((a.n_reads + 63) / 64, 1, n_cu * 64)), dim3(64), 0, stream, a);
	kt_end(KT_SAM_FORMAT, stream);
	return hipGetLastError();
}

hipError_t launch_sam_checksum(const SamArgs &a, int n_cu, hipStream_t stream)
{
	if (a.n_reads <= 0) return hipSuccess;
	hipLaunchKernelGGL(sam_checksum_kernel, dim3(n_cu * 8), dim3(256), 0, stream, a);
	return hipGetLastError();
}

}  // namespace kg
