// frag_kernels.hpp -- argument block of the fragment-pair alignment kernels (frag_kernels.hip) shared with the C ABI.
#pragma once
#include "seed_kernels.hpp"

namespace kg {

constexpr int kFragMaxLen = 8192;        // longest side of a fragment the partition kernel takes (2-bit codes of both sides in the LDS; 4096 until round 5:
                                         // 1.2 % of 7 kb reads on the hg38-sized genome have a stretch of 4-7 kb without seeds -- inside a repeat copy every seed has more than 50 hits)
constexpr int kFragMaxRuns = 383;        // exact matches of >= 8 bases per fragment pair (255 until round 5; a 7 kb stretch at 15 % error has ~290)
constexpr int kFragSmallLen = 1024, kFragSmallRuns = 127;   // the partition kernel's small instantiation (frag_kernels.hip)
constexpr int kFragMaxPairs = 2 * kFragMaxRuns + 2;   // ... and the normal pairs IdentifyNormalPairs makes of them
constexpr int kFragMaxDepth = 6;         // levels of the -pacbio recursion (src/tools.cpp:197)

// the partition kernel's waves take their pieces / NW jobs / op bytes from stretches of the lists they reserve at level 0 (one atomic per ~50 tasks);
// the lists need room for what the waves leave unused: frag_pool_slack()
constexpr int kFragPieceChunk = 1536, kFragJobChunk = 512, kFragOpsChunk = 24576;
constexpr int kFragWavesPerCu = 10, kFragSmallWavesPerCu = 32;      // the full-size instantiation: 15 KB of LDS per wave; the small one (KG_FRAG_TWO_TIERS): the CU's wave slots
int64_t frag_pool_waves(int64_t n_requests, int n_cu);              // waves of a level-0 launch that may leave a stretch of each list behind

enum { FP_JOB = 3, FP_TASK = 4 };        // FragPiece::kind beyond the literal runs KG_OP_DIAG / KG_OP_GAP1 / KG_OP_GAP2
enum { FC_TASKS = 0, FC_PIECES = 1, FC_JOBS = 2, FC_OPS = 3, FC_LEVEL0 = 4, FC_PROF = FC_LEVEL0 + kFragMaxDepth + 2, FC_WHY = FC_PROF + 12, FC_WORDS = FC_WHY + 6 };   // FC_PROF: wave cycles per phase (KG_FRAG_PROF)
// FC_WHY: tasks that sent their request back, by reason: [0] a side above kFragMaxLen, [1] a read character other than A/C/G/T, [2] more than kFragMaxRuns matches,
// [3] more normal pairs than the LDS arrays hold, [4] a work list was full, [5] recursion deeper than kFragMaxDepth

struct FragTask {
	int64_t f1_off;       // the read fragment in the characters the caller uploaded
	int64_t g;            // the genome fragment's text coordinate
	int32_t rL, gL;
	int32_t first, count; // its pieces
	int32_t status;       // 1: outside the envelope
	int32_t root;         // the request it belongs to
};
struct FragPiece {
	int32_t kind;         // KG_OP_*: a literal run of v columns; FP_JOB: NW job v; FP_TASK: task v (a sub-fragment partitioned again)
	int32_t v;
};

struct FragArgs {
	// requests
	const char *f1;                 // read fragments, concatenated
	const int64_t *off1;            // [n + 1]; with rlen: [n] offsets of fragments that lie anywhere in f1
	const int32_t *rlen = nullptr;  // [n] or null (the fragments are consecutive: rLen = off1[r + 1] - off1[r])
	const int64_t *gpos;            // [n] text coordinate of the genome fragment
	const int32_t *glen;            // [n]
	int64_t n;
	const uint8_t *text;            // 2-bit text of the index
	int64_t two_genome_size;
	int pacbio, max_gaps;
	int no_fast_pairs;              // KG_FRAG_NO_FAST_PAIRS (A/B aid): IdentifyNormalPairs always by lane 0
	int prof;                       // KG_FRAG_PROF: wave cycles per phase of the partition kernel into ctl[FC_PROF ..]
	int one_tier = 0;               // KG_FRAG_ONE_TIER (A/B aid): the partition kernel's full-size instantiation takes every task
	// work lists
	FragTask *tasks;
	int64_t task_capacity;
	FragPiece *pieces;
	int64_t piece_capacity;
	NwJobDesc *jobs;
	int64_t job_capacity, ops_capacity;
	uint8_t *job_ops;
	int32_t *job_len;
	unsigned long long *ctl;        // [FC_WORDS]
	// results
	uint8_t *status;                // [n] 1: the caller plans this request itself
	uint8_t *ops;                   // the op strings, request r at ops_off[r] (room for rLen + gLen columns)
	const int64_t *ops_off;         // [n]
	int32_t *aln_len;               // [n]
	int32_t *runs = nullptr;        // [n] or null: the number of runs of equal ops in every request's string (what its CIGAR takes at most)
};

hipError_t launch_frag_partition(const FragArgs &a, int n_cu, hipStream_t stream);   // levels 0 .. kFragMaxDepth - 1: pieces, NW jobs
hipError_t launch_frag_stitch(const FragArgs &a, int n_cu, hipStream_t stream);      // after the NW kernels: the requests' op strings

}  // namespace kg
