// align_kernels.hpp -- argument block of the device-side alignment stage (align_kernels.hip) shared with the C-ABI layer.
//
// What runs on the device after chaining, for the common case of short-read candidates (reference file:line in
// align_kernels.hip): candidate pairing + filters, IdentifyNormalPairs, the per-pair decisions of
// Process{Head,Normal,Tail}SequencePair, gap closing through the NW kernels, CIGAR / AlnScore / coordinates
// (GenMappingReport), best / second-best, CheckPairedFinalAlignments, SAM flags, MAPQ -- one kg_aln_record per read.
// Whatever the device path does not take (mate rescue, the 8-mer partition of long gap fragments, candidates with many
// seeds, very long CIGARs) is handed back per read pair (record kind KG_ALN_HOST) to the host's implementation of the same
// reference code.
#pragma once
#include "seed_kernels.hpp"

namespace kg {

constexpr int kAlnMaxSeeds = 12;                       // seeds of one candidate the device path takes
constexpr int kAlnMaxGaps = 8;                         // gap pairs inserted between them (the host merges more than 8 differently)
constexpr int kAlnMaxPairs = kAlnMaxSeeds + kAlnMaxGaps + 2;   // + head + tail
constexpr int kAlnMaxCigar = 2 * kAlnMaxPairs + 8;     // (length, op) elements before merging
constexpr int kAlnMaxFrag = 255;                       // longest side of a gap fragment sent to the NW kernels from here
constexpr int kAlnMaxScore = 2047;                     // rows of the MAPQ table
constexpr int kAlnPairProduct = 4096;                  // n1 * n2 above which a pair's candidate pairing is left to the host

// one normal pair as the finish pass needs it
struct AlnSpillPair {
	int64_t gPos;
	int32_t rPos;
	int16_t rLen, gLen;
	int32_t val;          // SIMPLE: unused; IMMEDIATE: score; JOB: job index
	uint8_t kind;         // 0 none, 1 simple, 2 immediate, 3 NW job
	uint8_t op;           // IMMEDIATE: CIGAR op character (0 = no element)
	int16_t op_len;
};

// A normal pair that took the 8-mer partition route of GenerateNormalPairAlignment (src/tools.cpp:146-212): its alignment is the
// concatenation of pieces -- literal stretches and the op strings of sub-fragment jobs -- assembled into one op string by the
// finish pass.
struct AlnPiece {
	int32_t v;            // literal: number of columns; job: job index
	uint8_t kind;         // KG_OP_DIAG / KG_OP_GAP1 / KG_OP_GAP2: a literal run of that op; 3: an NW job
};
struct AlnPlan {
	int64_t ops;          // where the assembled op string goes (room for rLen + gLen columns)
	int32_t first, count; // pieces
};

// A fragment pair of a parked candidate that waits for its 8-mer partition (aln_partition_kernel: dense, one task per lane)
struct PartTask {
	int64_t enc_off;      // the read fragment's offset in the batch's characters
	int64_t g;            // the genome fragment's text position
	int32_t spill, j;     // whose pair it is
	int32_t read;
	int16_t rL, gL;
};

struct AlnSpill {
	int32_t cand;         // dense candidate index
	int32_t num;
	AlnSpillPair p[kAlnMaxPairs];
};

// One window of RescueUnpairedAlignment (src/AlignmentRescue.cpp:127-165): mate 1 of pair `read` is searched in the reference
// window [left, left + slen) next to candidate `j` of mate 2.  The rescued candidate, if any, takes candidate slot
// n_cands + (index of the task).
struct RescueTask {
	int64_t left;
	int32_t read;         // mate 1 (the read whose 8-mers are looked up)
	int32_t j;            // candidate of mate 2 the window belongs to
	int32_t slen;
	int32_t score1;       // the rescued candidate must beat this (best score of mate 1 before the rescue)
	int32_t ordinal;      // index of the task among the tasks of its pair
};

constexpr int kRescueMaxRead = 256;                    // longest mate the rescue kernel takes
constexpr int kRescueMaxWindow = 2048;                 // longest window (EstDistance <= MaxInsertSize 1500 + read length)
constexpr int kRescueMaxRuns = 256;                    // exact-match runs of one window the kernel keeps

struct AlnArgs {
	FmView ix;                      // text
	// reads
	const uint8_t *enc;             // read characters
	const int64_t *read_off;
	int64_t n_reads;
	// chunks of the batch (reference: 4000-read chunks; a chunk is paired when the run is and its count is even)
	const int64_t *chunk_off;       // [n_chunks + 1]
	const uint8_t *chunk_paired;    // [n_chunks]
	int n_chunks;
	int all_paired;                 // every chunk is paired (the usual case): aln_pair_kernel then takes pair u = reads 2u, 2u + 1 per lane
	// candidates (dense, read order)
	const int64_t *cand_off;        // [n_reads + 1]
	const kg_candidate *cands;
	const kg_seed *cand_seeds;
	int64_t n_cands;
	// contigs
	const int64_t *contig_end;      // ChrLocMap keys, ascending
	const int32_t *end_chr;         // contig index of every key
	int n_ends;
	const int64_t *chr_fwd_start, *chr_rev_start, *chr_len;
	int n_chr;
	int64_t genome_size, two_genome_size;
	// parameters
	int est_distance, max_insert, max_gaps;
	int multi_hit;                  // -m: every candidate the reference's output loops print (src/Mapping.cpp:194-223, 242-263, 293)
	int unset_flag;                 // what a never-assigned SamFlag prints as
	int dbg_rescue_scan;            // A/B aid (KG_RESCUE_SCAN): the rescue kernel walks every diagonal of the window (rounds 2-3) instead of looking 10-mers up
	int dbg_no_inline;              // 1 unless KG_ALN_INLINE: every alignment goes through the NW kernels
	int dbg_finish_lanes;           // A/B aids: 1 (KG_ALN_FINISH_LANES) a candidate per lane in the finish pass (rounds 2-5), 2 (KG_ALN_FINISH_WAVE) a candidate per wave, 3 (KG_ALN_FINISH_G16) per sixteen lanes; 0: per group of eight lanes
	int dbg_plan_group;             // KG_ALN_PLAN_GROUP: pass 1 by groups of eight lanes (aln_plan_group_kernel) instead of a candidate per lane
	int pair_heavy;                 // pairs with more candidate pairs than this are the whole wave's (kPairHeavy; KG_ALN_PAIR_HEAVY)
	int dbg_no_heavy;               // A/B aid (KG_ALN_NO_HEAVY): pairs with many candidates stay in their lane's loop (rounds 2-5)
	int dbg_no_partition;           // diagnostics (KG_DBG_NO_PARTITION): fragments that need the 8-mer partition go back to the host
	int64_t extra_capacity;         // records beyond one per read: records[n_reads .. n_reads + extra_capacity), handed out through ctl[7]
	const uint8_t *mapq_tab;        // [(kAlnMaxScore + 1) * 6]: EvaluateMAPQ's libm branch, tabulated by the host
	// per-candidate state
	int32_t *c_score, *c_mate, *c_read;
	int32_t *rep_score, *rep_chr;
	int64_t *rep_pos;
	uint8_t *rep_fwd, *rep_cigar_len;
	char *rep_cigar;                // [n_cands * KG_ALN_CIGAR_MAX]
	// mate rescue
	RescueTask *tasks;
	int64_t task_capacity;
	int64_t *resc_posdiff;          // [task_capacity] PosDiff of the rescued candidate
	int32_t *resc_count;            // [task_capacity] its simple pairs ...
	kg_seed *resc_seeds;            // [task_capacity * kAlnMaxSeeds] ... sorted by (gPos, rPos)
	int32_t *resc_off;              // [n_reads] first task of the pair whose mate 1 this read is
	uint8_t *resc_n;                // [n_reads] number of tasks (= rescue candidate slots appended to this read's list)
	uint8_t *r_pending;             // [n_reads] the pair waits for its rescue windows
	// per-read state
	uint8_t *r_host;                // the pair of this read goes back to the host
	// spill + jobs
	AlnSpill *spill;
	int64_t spill_capacity;
	NwJobDesc *jobs;
	int64_t job_capacity, ops_capacity;
	AlnPlan *plans;                 // [job_capacity]
	AlnPiece *pieces;               // [4 * job_capacity]
	PartTask *part_tasks;           // [job_capacity], handed out through ctl[3]
	int32_t *plan_order;            // [n_cands + task_capacity] or null: the order in which aln_plan_kernel takes the candidates (binned by seed count, ctl[24..31])
	int32_t *slow_pairs;            // [n_reads / 2] or null: the pairs aln_trivial_kernel did not decide (ctl[35] of them) -- the only ones the kernels behind it look at
	int32_t *slow_cands;            // [n_cands]: the candidates of those pairs (ctl[34])
	int32_t *plan_slow;             // [n_cands + task_capacity] or null: the candidates aln_plan_fast_kernel left to aln_plan_kernel (ctl[32] of them)
	unsigned long long *ctl;        // [0] spill count, [1] job count, [2] ops bytes, [3] partition tasks, [4] rescue tasks, [5] plans, [6] pieces, [7] extra records (-m)
	uint8_t *nw_ops;
	int32_t *nw_len;
	// outputs
	kg_aln_record *records;         // [n_reads]
	kg_chunk_stats *chunk_stats;    // [n_chunks]
};

hipError_t launch_align_front(const AlnArgs &a, int n_cu, hipStream_t stream);     // pairing, mate rescue, plan (+ finish of job-free candidates)
hipError_t launch_align_back(const AlnArgs &a, int n_cu, hipStream_t stream);      // finish of the spilled candidates + per-read records

}  // namespace kg
