// fm_device.hpp -- device-side view of the FM-index and the rank / LF primitives (gfx950).
//
// Data layout in HBM (see DESIGN.md "Index layout"):
//   occ   : the reference's interleaved Occ/BWT blocks exactly as in the .bwt file, one 64-byte block per
//           128 BWT symbols: bytes 0..31 = four u64 running counts (A,C,G,T before the block), bytes
//           32..63 = eight u32 words of 16 symbols each, first symbol in bits 31:30 (reference
//           src/BWT_Index/bwtindex.c:53-75, macros src/bwt_search.cpp:31-33).  Only the load-time
//           converter (build_planes_kernel) reads it.
//   planes: DEVICE-PRIVATE re-layout built once at load (build_planes_kernel): per 64 BWT symbols one
//           64-byte block of four 16-byte segments, segment c = { u64 bit-plane: bit j set iff symbol
//           j of the block == c, u64 count of base c before the block } (plane first, so a 32-bit
//           index needs one dwordx3 per rank).  A rank query is ONE
//           16-byte load + mask + popcount (the 2-bit layout costs ~150 VALU ops per LF step, which
//           made the search kernel VALU-bound even with the index in L2).  1 byte/symbol: 9.3 MB
//           for E. coli, 6.2 GB for hg38 -- cheap against 288 GB of HBM; every rank still touches
//           exactly one 64-byte line.
//   planes2: DEVICE-PRIVATE two-step rank structure built once at load (index_build.inc): for every ordered pair of
//           bases (c1, c2) a bit-plane over the BWT rows: bit p set iff BWT[p] == c1 and BWT[LF(p)] == c2 (the two text
//           characters in front of the row's suffix).  One 128-byte LINE = 896 rows of ONE pair: a 16-byte header
//           { u64 count of the pair before the line's first row, 6 x 10-bit counts of the line's rows before its 2nd ..
//           7th segment } and seven 16-byte segments of 128 rows; line index = (p / 896) * 16 + c1 * 4 + c2.  A rank =
//           header + one segment (two 16-byte gathers from ONE line).  Two backward-search steps then cost one rank pair
//           in one plane: both ends of an interval narrower than a line come out of the same line, wider ones out of two
//           -- against (1 + 2 x 0.63) x 2 lines for the same two steps on `planes`.  2.29 bytes/symbol: 14.2 GB for hg38.
//   sa    : u64 sample per 32 ranks, sa[0] = (u64)-1 (reference src/bwt_index.cpp:16-36).
//   fsa   : optional full suffix array (u32 when 2L < 2^32, else u64), expanded on the device at
//           load time; turns the ~31-step LF walk of bwt_sa() into one 4/8-byte gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kg {

struct FmView {
	const uint4 *planes;   // device-private rank structure, see below
	const uint4 *planes2;  // two-step rank structure (null: single steps only), see below
	const uint4 *planes3;  // three-step rank structure (null: at most double steps): the same lines for the 64 triples of bases
	const uint32_t *occ;
	const uint64_t *sa;
	const uint32_t *fsa32;
	const uint64_t *fsa64;
	const uint8_t *fsa40;  // KG_SA_FULL40: the expanded suffix array in 5-byte little-endian entries (entry k at byte 5 k)
	// KG_SA_DENSE4 / 8: entry i = SA[i << dsa_shift] (null otherwise); SA[k] = dsa[k' >> shift] + steps after `steps` LF steps k -> k'
	const uint32_t *dsa32;
	const uint64_t *dsa64;
	int dsa_shift;
	// q-mer interval table (device-private, built at load): for the first `qmer` bases of a search, the
	// interval after qmer-1 extension steps.  q grows with the text (4^q ~ 2L: 12 for E. coli, 16 for hg38)
	// so that the jump lands on intervals of a few suffixes.  Entry = { k, n | sa << 27 | lf2 << 28 } (u32
	// index) or k | n << 34 | sa << 59 | lf2 << 60 (u64 index); n == 0 means "no such q-mer / not
	// representable": the search then starts step by step, so results never depend on the table.  sa = 1
	// (n == 1 and the full SA resident): k is SA[k] already, the search continues against the text.  sa = 1 with n == 0: the q-mer
	// does not occur in the text at all (SensitiveMode then knows the search has no hit without making it).
	const uint2 *qtab32;
	const uint64_t *qtab64;
	int qmer;
	// the indexed text itself (forward strand then reverse complement, 2L bases), 2 bits per base, base i in
	// bits 2*(i&3) of byte i>>2, built at load from the .pac bases when the full SA is: once a search's interval
	// has shrunk to ONE suffix the rest of the match is a plain comparison against the text at that suffix
	const uint8_t *text;
	uint64_t primary;
	uint64_t seq_len;
	uint64_t L2[5];
	// two steps at once (first c1, then c2): k'' = t2[c1 * 4 + c2] + rank2(k - 1), where t2 = L2[c2] + 1 + occ(L2[c1], c2)
	uint64_t t2[16];
	uint64_t t3[64];       // three steps: k3 = t3[c1 * 16 + c2 * 4 + c3] + rank3(k - 1), t3 = L2[c3] + 1 + occ(t2[c1 c2] - 1, c3)
};

struct __attribute__((packed, aligned(1))) FsaU64u { uint64_t v; };

// the expanded suffix array, whichever entry width is resident
__device__ __forceinline__ bool fsa_resident(const FmView &ix) { return ix.fsa32 != nullptr || ix.fsa64 != nullptr || ix.fsa40 != nullptr; }
__device__ __forceinline__ uint64_t fsa_entry40(const uint8_t *fsa40, uint64_t k)
{
	return reinterpret_cast<const FsaU64u *>(fsa40 + 5 * k)->v & 0xFFFFFFFFFFull;       // (the array ends with 8 spare bytes)
}
__device__ __forceinline__ uint64_t fsa_entry(const FmView &ix, uint64_t k)
{
	return ix.fsa32 ? (uint64_t)ix.fsa32[k] : ix.fsa40 ? fsa_entry40(ix.fsa40, k) : ix.fsa64[k];
}

constexpr uint32_t kPlane2Rows = 896;   // rows per 128-byte line of planes2 (7 segments of 128)

constexpr int kQmerMin = 8, kQmerMax = 16;

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

// rank of row `pos` (0..895, inclusive) within one planes2 line, from its header and the segment holding the row
__device__ __forceinline__ uint64_t rank2_line(uint4 hdr, uint4 sg, uint32_t pos)
{
	uint64_t cnt = ((uint64_t)hdr.y << 32) | hdr.x, subs = ((uint64_t)hdr.w << 32) | hdr.z;
	uint32_t s = pos >> 7, w = pos & 127;
	uint32_t sub = s ? (uint32_t)(subs >> (10 * (s - 1))) & 1023u : 0u;
	uint64_t lo = ((uint64_t)sg.y << 32) | sg.x, hi = ((uint64_t)sg.w << 32) | sg.z;
	uint64_t mlo = w >= 64 ? ~0ull : (2ull << w) - 1;                 // (w = 63: 2 << 63 wraps to 0, minus 1 = all ones)
	uint64_t mhi = w < 64 ? 0ull : (2ull << (w - 64)) - 1;
	return cnt + sub + (uint64_t)(__popcll(lo & mlo) + __popcll(hi & mhi));
}

// occurrences of the pair (c1, c2) -- or, on planes3 with sh = 6, of the triple (c1, c2, c3) -- in rows [0..kk] ($-less coordinate)
__device__ __forceinline__ uint64_t rank2_plane(const uint4 *planes_k, int sh, uint64_t kk, int code)
{
	uint64_t blk = kk / kPlane2Rows;
	uint32_t pos = (uint32_t)(kk - blk * kPlane2Rows);
	const uint4 *line = planes_k + (((blk << sh) + (uint64_t)code) << 3);
	return rank2_line(line[0], line[1 + (pos >> 7)], pos);
}

}  // namespace kg
