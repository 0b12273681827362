// fm_device.hpp -- device-side view of the FM-index and the rank / LF primitives (gfx950).
//
// Data layout in HBM (see DESIGN.md "Index layout"):
//   occ   : the reference's interleaved Occ/BWT blocks, one 64-byte block per 128 BWT symbols:
//           bytes 0..31  = four u64 running counts (A,C,G,T before the block),
//           bytes 32..63 = eight u32 words of 16 symbols each, first symbol in bits 31:30
//           (reference src/BWT_Index/bwtindex.c:53-75, macros src/bwt_search.cpp:31-33).
//           One block = one 64-byte HBM/L2 access; a lane fetches it with a dwordx2 (its base's
//           count) and two dwordx4 (the symbols).
//   sa    : u64 sample per 32 ranks, sa[0] = (u64)-1 (reference src/bwt_index.cpp:16-36).
//   fsa   : optional full suffix array (u32 when 2L < 2^32, else u64), expanded on the device at
//           load time; turns the ~31-step LF walk of bwt_sa() into one 4/8-byte gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kg {

struct FmView {
	const uint32_t *occ;
	const uint64_t *sa;
	const uint32_t *fsa32;
	const uint64_t *fsa64;
	uint64_t primary;
	uint64_t seq_len;
	uint64_t L2[5];
};

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

}  // namespace kg
