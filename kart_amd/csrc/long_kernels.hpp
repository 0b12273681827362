// long_kernels.hpp -- argument block of the long-read (-pacbio) report kernels (long_kernels.hip) shared with the C ABI (abi_long.hip).
//
// What runs on the device after chaining for the reference's bPacBioData branch of ReadMapping() (src/Mapping.cpp:513-530):
// RemoveRedundantCandidates (:317-346), GenMappingReport (src/AlignmentCandidates.cpp:624-745) with IdentifyNormalPairs of the
// whole read (:420-490), CheckCoordinateValidity (:582-610), Process{Head,Normal,Tail}SequencePair (src/tools.cpp:225-397) with
// GenerateNormalPairAlignment on the fragment kernels (frag_kernels.hip), GenCoordinateInfo / GenerateCIGAR (:492-562),
// SetSingleAlignmentFlag / EvaluateMAPQ (src/Mapping.cpp:49-70, 160-175) -- one kg_aln_record per read whose CIGAR lies in a text pool.
#pragma once
#include "seed_kernels.hpp"

namespace kg {

enum { LP_NONE = 0, LP_SIMPLE = 1, LP_IMM = 2, LP_REQ = 3 };     // LrPair::kind
enum { LS_INVALID = 0, LS_PLANNED = 1, LS_HOST = 2 };            // LrCand::state
// ctl words
enum { LC_LIVE = 0, LC_POOL = 1, LC_REQ = 2, LC_COLS = 3, LC_MAXLEN = 4, LC_ELEMS = 5, LC_HOST = 6, LC_TEXT = 7,
       LC_R_DASH = 8, LC_R_FRAG = 9, LC_R_ORDER = 10, LC_R_ELEMS = 11, LC_R_OVERLAP = 12, LC_WORDS = 16 };

// one pair of a candidate after IdentifyNormalPairs (SeedPair_t, src/structure.h:106-114, plus what pass 1 decided for it)
struct LrPair {
	int64_t gPos;
	int32_t rPos, rLen, gLen;
	int32_t v;            // LP_IMM: the pair's score (-1: the > 3000 soft clip); LP_REQ: index of its fragment request
	uint8_t kind;
	uint8_t op;           // LP_IMM: CIGAR op character
	uint8_t pad[2];
};

// a candidate that takes part in GenMappingReport's loop (Score != 0 after RemoveRedundantCandidates)
struct LrCand {
	int64_t pair_off;     // its slice of the pair pool (2 * count + 3 entries)
	int64_t elem_off;     // its merged CIGAR elements (len << 2 | op; op: 0 M, 1 I, 2 D, 3 S)
	int64_t pos;          // 1-based position on the contig (Coordinate_t::gPos)
	int32_t cand, read;
	int32_t n_pairs;
	int32_t state;
	int32_t score;        // AlnScore
	int32_t chr;
	int32_t n_elems;
	int32_t text_bytes;   // length of the CIGAR string
	uint8_t fwd;
	uint8_t pad[7];
};

struct LrArgs {
	// the batch as seeding + chaining left it
	const uint8_t *enc;             // read characters
	const int64_t *read_off;
	int64_t n_reads;
	const int64_t *cand_off;        // [n_reads + 1]
	const kg_candidate *cands;      // dense, read order
	kg_seed *seeds;                 // their seeds (CheckOverlappingSeeds works on them in place)
	int64_t n_cands;
	// text and contigs
	const uint8_t *text;            // 2 bits per base, forward + reverse complement
	int64_t genome_size, two_genome_size;
	const int64_t *contig_end;      // ChrLocMap keys, ascending
	const int32_t *end_chr;
	int n_ends;
	const int64_t *chr_fwd_start;
	int n_chr;
	// work lists
	int32_t *cand_lc;               // [n_cands] index into lcs, or -1
	LrCand *lcs;                    // [n_cands]
	LrPair *pool;
	int64_t pool_capacity;
	int64_t *req_f1;                // [req_capacity] the read fragment's offset in enc
	int64_t *req_g;                 // its text coordinate
	int32_t *req_rl, *req_gl;
	int64_t *req_oo;                // where its op string goes
	int64_t req_capacity;
	unsigned long long *ctl;        // [LC_WORDS]
	// after the fragment kernels
	const uint8_t *ops;
	const int32_t *aln_len;
	const int32_t *runs;            // runs of equal ops per request (frag_stitch_kernel): what its CIGAR takes at most
	const uint8_t *status;
	uint32_t *elems;
	int64_t elem_capacity;
	// per read
	uint8_t *r_host;                // [n_reads] the host maps this read
	int64_t *cig_bytes;             // [n_reads + 1] bytes of CIGAR text per read, then its exclusive scan
	int32_t *r_best;                // [n_reads] the lc whose report is printed, or -1
	kg_aln_record *records;         // [n_reads]
	char *cigar;                    // the text pool
};

hipError_t launch_long_plan(const LrArgs &a, int n_cu, hipStream_t stream);       // select + IdentifyNormalPairs + pass 1 (requests)
hipError_t launch_long_finish(const LrArgs &a, int n_cu, hipStream_t stream);     // pass 2 (elements, scores, coordinates) + per-read records
size_t long_scan_temp_bytes(int64_t max_items);
hipError_t launch_long_scan(const LrArgs &a, void *temp, size_t temp_bytes, hipStream_t stream);             // cig_bytes -> offsets, in place
hipError_t launch_long_text(const LrArgs &a, int n_cu, hipStream_t stream);       // GenerateCIGAR into the text pool (after the scan of cig_bytes)

}  // namespace kg
