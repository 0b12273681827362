// fm_device.hpp -- device-side view of the FM-index and the rank / LF primitives (gfx950).
//
// Data layout in HBM (see DESIGN.md "Index layout"):
//   occ   : the reference's interleaved Occ/BWT blocks, one 64-byte block per 128 BWT symbols:
//           bytes 0..31  = four u64 running counts (A,C,G,T before the block),
//           bytes 32..63 = eight u32 words of 16 symbols each, first symbol in bits 31:30
//           (reference src/BWT_Index/bwtindex.c:53-75, macros src/bwt_search.cpp:31-33).
//           One block = one 64-byte HBM/L2 access; a lane fetches it with a dwordx2 (its base's
//           count) and two dwordx4 (the symbols).
//   planes: DEVICE-PRIVATE re-layout built once at load (build_planes_kernel): per 64 BWT symbols one
//           64-byte block of four 16-byte segments, segment c = { u64 bit-plane: bit j set iff symbol
//           j of the block == c, u64 count of base c before the block } (plane first, so a 32-bit
//           index needs one dwordx3 per rank).  A rank query is ONE
//           16-byte load + mask + popcount (the 2-bit layout costs ~150 VALU ops per LF step, which
//           made the search kernel VALU-bound even with the index in L2).  1 byte/symbol: 9.3 MB
//           for E. coli, 6.2 GB for hg38 -- cheap against 288 GB of HBM; every rank still touches
//           exactly one 64-byte line.
//   sa    : u64 sample per 32 ranks, sa[0] = (u64)-1 (reference src/bwt_index.cpp:16-36).
//   fsa   : optional full suffix array (u32 when 2L < 2^32, else u64), expanded on the device at
//           load time; turns the ~31-step LF walk of bwt_sa() into one 4/8-byte gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kg {

struct FmView {
	const uint4 *planes;   // device-private rank structure, see below
	const uint32_t *occ;
	const uint64_t *sa;
	const uint32_t *fsa32;
	const uint64_t *fsa64;
	// q-mer interval table (device-private, built at load): for the first kQmer bases of a search,
	// the interval after kQmer-1 extension steps.  Entry = { k, n | lf2 << 28 } (u32 index) or
	// { k lo, k hi, n, lf2 } (u64 index); n == 0 means "no such q-mer / not representable": the
	// search then starts step by step, so results never depend on the table.
	const uint2 *qtab32;
	const uint4 *qtab64;
	uint64_t primary;
	uint64_t seq_len;
	uint64_t L2[5];
};

constexpr int kQmer = 12;

// bit 2i set for every 2-bit field of w (16 fields, MSB first) equal to the base whose
// replicated pattern is `pat` (= base * 0x55555555)
__device__ __forceinline__ uint32_t match_fields(uint32_t w, uint32_t pat)
{
	uint32_t x = ~(w ^ pat);
	return x & (x >> 1) & 0x55555555u;
}

// mask selecting the first t fields (t clamped to [0,16]) of a 16-field word
__device__ __forceinline__ uint32_t head_fields(int t)
{
	t = t < 0 ? 0 : t;
	return t >= 16 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (2 * t));
}

struct OccBlock {
	uint64_t cnt;   // running count of the wanted base before the block
	uint4 lo, hi;   // 128 symbols
};

__device__ __forceinline__ OccBlock load_block(const FmView &ix, uint64_t blk, int c)
{
	const uint32_t *p = ix.occ + (blk << 4);
	OccBlock b;
	b.cnt = *reinterpret_cast<const uint64_t *>(p + 2 * c);
	b.lo = *reinterpret_cast<const uint4 *>(p + 8);
	b.hi = *reinterpret_cast<const uint4 *>(p + 12);
	return b;
}

// occurrences of the base with pattern `pat` among the first m symbols (0..128) of the block
__device__ __forceinline__ uint32_t count_head(const OccBlock &b, uint32_t pat, int m)
{
	uint32_t n = 0;
	n += __popc(match_fields(b.lo.x, pat) & head_fields(m));
	n += __popc(match_fields(b.lo.y, pat) & head_fields(m - 16));
	n += __popc(match_fields(b.lo.z, pat) & head_fields(m - 32));
	n += __popc(match_fields(b.lo.w, pat) & head_fields(m - 48));
	n += __popc(match_fields(b.hi.x, pat) & head_fields(m - 64));
	n += __popc(match_fields(b.hi.y, pat) & head_fields(m - 80));
	n += __popc(match_fields(b.hi.z, pat) & head_fields(m - 96));
	n += __popc(match_fields(b.hi.w, pat) & head_fields(m - 112));
	return n;
}

// both ranks of one block (the common case of bwt_2occ4, reference src/bwt_search.cpp:87-118)
__device__ __forceinline__ void count_head2(const OccBlock &b, uint32_t pat, int m1, int m2, uint32_t &n1, uint32_t &n2)
{
	uint32_t w[8] = {b.lo.x, b.lo.y, b.lo.z, b.lo.w, b.hi.x, b.hi.y, b.hi.z, b.hi.w};
	n1 = n2 = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		uint32_t f = match_fields(w[i], pat);
		n1 += __popc(f & head_fields(m1 - 16 * i));
		n2 += __popc(f & head_fields(m2 - 16 * i));
	}
}

__device__ __forceinline__ int symbol_of(const OccBlock &b, int idx)  // idx in 0..127
{
	uint32_t w[8] = {b.lo.x, b.lo.y, b.lo.z, b.lo.w, b.hi.x, b.hi.y, b.hi.z, b.hi.w};
	uint32_t word = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) word = (idx >> 4) == i ? w[i] : word;
	return (word >> ((~idx & 15) << 1)) & 3;
}

// One bwt_invPsi step (reference src/bwt_search.cpp:120-126): k -> rank of the suffix one text
// position to the left.  A single block fetch gives both BWT[k] and its rank.
__device__ __forceinline__ uint64_t lf_step(const FmView &ix, uint64_t k)
{
	if (k == ix.primary) return 0;
	uint64_t kk = k - (k > ix.primary);
	const uint4 *p = reinterpret_cast<const uint4 *>(ix.occ + ((kk >> 7) << 4));
	// whole 64-byte line in one go: BWT[k] decides which of the four counts is used, and a second,
	// dependent fetch of that count would double the latency of every step of the walk
	uint4 c01 = p[0], c23 = p[1];
	OccBlock b;
	b.lo = p[2];
	b.hi = p[3];
	int idx = (int)(kk & 127);
	int c = symbol_of(b, idx);
	uint32_t lo = c == 0 ? c01.x : c == 1 ? c01.z : c == 2 ? c23.x : c23.z;
	uint32_t hi = c == 0 ? c01.y : c == 1 ? c01.w : c == 2 ? c23.y : c23.w;
	uint32_t n = count_head(b, (uint32_t)c * 0x55555555u, idx + 1);
	return ix.L2[c] + (((uint64_t)hi << 32) | lo) + n;
}

// ---- bit-plane rank structure ---------------------------------------------------------------

// occurrences of base c in BWT[0..kk] ($-less coordinate), one 16-byte gather
__device__ __forceinline__ uint64_t rank_plane(const FmView &ix, uint64_t kk, int c)
{
	uint4 v = ix.planes[((kk >> 6) << 2) + (uint64_t)c];
	uint64_t bits = ((uint64_t)v.y << 32) | v.x;
	uint64_t cnt = ((uint64_t)v.w << 32) | v.z;
	uint64_t m = (2ull << (kk & 63)) - 1;          // bits 0..pos (pos = 63 wraps to all ones)
	return cnt + (uint64_t)__popcll(bits & m);
}

// bwt_invPsi on the plane layout: the whole 64-byte block (four segments) in one go, the plane
// whose bit is set at the position names BWT[k] and also gives its rank
__device__ __forceinline__ uint64_t lf_step_plane(const FmView &ix, uint64_t k)
{
	if (k == ix.primary) return 0;
	uint64_t kk = k - (k > ix.primary);
	const uint4 *p = ix.planes + ((kk >> 6) << 2);
	uint4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
	uint32_t pos = (uint32_t)(kk & 63);
	uint64_t b1 = ((uint64_t)s1.y << 32) | s1.x, b2 = ((uint64_t)s2.y << 32) | s2.x, b3 = ((uint64_t)s3.y << 32) | s3.x;
	int c = ((b1 >> pos) & 1) ? 1 : ((b2 >> pos) & 1) ? 2 : ((b3 >> pos) & 1) ? 3 : 0;
	uint4 v = c == 0 ? s0 : c == 1 ? s1 : c == 2 ? s2 : s3;
	uint64_t bits = ((uint64_t)v.y << 32) | v.x;
	uint64_t cnt = ((uint64_t)v.w << 32) | v.z;
	uint64_t m = (2ull << pos) - 1;
	uint64_t l2 = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
	return l2 + cnt + (uint64_t)__popcll(bits & m);
}

}  // namespace kg
