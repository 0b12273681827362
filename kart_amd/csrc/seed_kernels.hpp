// seed_kernels.hpp -- argument blocks shared by seed_kernels.hip and the C-ABI layer.
#pragma once
#include "../../include/kart_amd.h"
#include "fm_device.hpp"

namespace kg {

// one successful BWT_Search (len >= MinSeedLength, freq <= OCC_Thr): the interval on the
// reverse-complement side plus where its seeds go inside the read's output segment
struct Hit {
	uint64_t k;          // first rank of the interval (x[1] of the reference's bi-interval)
	int32_t read;
	int32_t rpos;
	int32_t len;
	int32_t n;           // interval size = number of seeds
	int32_t seed_start;  // running seed count of the read before this hit
	int32_t direct;      // 1: k already holds the text position of the pattern (search finished against the text)
};

// device control block: zeroed by one hipMemsetAsync per batch
//   [0] read queue head  [1] hit count  [2] locate queue head  [3] unused
//   [4..11] counters: searches, lf1, lf2, inv, sa, seeds, bases, overflow(needed seeds or 0)
//   [12..16] reads on the sort work lists: 9..16, 17..32, 33..64 seeds (4 / 2 / 1 lists per wave), 65..256 (small LDS),
//            > 512 (large LDS)
//   [17..21] what the search kernel itself fetched: q-mer table lookups, rank steps executed, those touching two 128-byte
//            lines, text-comparison rounds, packed window words, two-line rank steps on intervals below 960, double steps, those with two lines, bytes their ranks read (kg_workspace_traffic)
//   [27] reads on the sixth sort work list: 257..512 seeds (medium LDS)
constexpr int kCtlWords = 28;
constexpr int kMaxSeedSegments = 8;

struct SeedArgs {
	FmView ix;
	const uint8_t *enc;
	const int64_t *read_off;
	int64_t n_reads;
	int64_t n_bases;
	int mode, min_seed_len, occ_thr;
	int ascii;           // the read bytes are characters, not codes (KG_INPUT_ASCII)
	// A GROUPED batch (abi_stream.hip: ONE seeding launch over the parsed batches of several stream lanes -- the MI355X form of the
	// reference's N workers on N chunks, src/Mapping.cpp:716-717): n_seg segments of seg_stride read slots each, segment s holding
	// seg_prefix[s+1] - seg_prefix[s] reads in its first slots; the rest of a segment's slots are empty (read_len 0).  read_off
	// then jumps from one segment's part of `enc` to the next, so the lengths come from read_len, and the search kernel's lanes
	// draw tickets 0 .. seg_prefix[n_seg]-1 that skip the empty slots.  n_seg == 0: an ordinary batch (read_len null).
	const int32_t *read_len;
	int n_seg;
	int64_t seg_stride;
	int64_t seg_prefix[kMaxSeedSegments + 1];
	// scratch
	uint64_t *packed;    // 4-bit read codes, 16 per word, read r at word (read_off[r] >> 4) + 3 r
	// EXPERIMENT (KG_SORT_READS): the order in which the search kernel's lanes draw the reads -- sorted by the first 16 bases, so
	// that lanes of a wave start in neighbouring q-mer table entries / rank lines; null = input order
	int single_steps;         // 1: never take two steps at once (the reference's per-step block accounting, counters lf1 / lf2, is then exact)
	int32_t *read_order;
	uint32_t *sort_keys;      // [4 * max_reads]: keys in, keys out, ids in, ids out
	void *sort_temp;
	size_t sort_temp_bytes;
	// SensitiveMode within a read in parallel (long reads): the search of IdentifySeedPairs_SensitiveMode at read position p depends on the read
	// alone (it sees at most the 30 bases behind p, src/AlignmentCandidates.cpp:132-169), and where the loop goes next depends only on its result --
	// the loop is a walk p -> next(p).  A read is cut into segments of seg_len bases; a lane starts a walk at every segment start (a "virtual read":
	// vr_read / vr_pos).  Before a lane searches at p it CLAIMS p (one bit per read position, atomicOr): a position somebody claimed already is
	// searched exactly once by that somebody, who walks on from it -- the lane's walk has merged and ends.  The walk from position 0 is therefore
	// covered link by link whichever lane ran which part; every search writes its advance to step[p]; seg_path_kernel follows the links from 0 and
	// marks them; seg_filter_kernel keeps the hits of marked positions and gives them their places in the read's seed list.  The walks from the later
	// starts that were not on the path (they merge after ~160 bases at 15 % error) are the price: ~15 % more searches for a chain of ~45 dependent
	// searches instead of ~470 per 7 kb read.  null vr_read: one lane per read, as the reference's loop.
	const int32_t *vr_read = nullptr, *vr_pos = nullptr;
	const unsigned long long *vr_total = nullptr;      // number of virtual reads (device)
	uint32_t *claim = nullptr;                        // read r's bits start at word (read_off[r] >> 5) + r
	uint8_t *step = nullptr;                          // [n_bases]: advance of the search (or skip) at every position visited; bit 7: on the walk from 0
	int seg_len = 0;
	Hit *hits;
	int64_t max_hits;
	int32_t *seeds_per_read;
	unsigned long long *read_queue, *hit_count, *locate_queue, *counters, *traffic;
	// outputs
	int64_t *seed_off;
	kg_seed *seeds;
	int64_t seed_capacity;
};

size_t scan_temp_bytes(int64_t max_reads);
size_t sort_temp_bytes(int64_t max_reads);
// ev: optional array of 5 events recorded before/after the four phases (search | scan | locate | sort)
hipError_t launch_seed_batch(const SeedArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream, hipEvent_t *ev);
// the virtual reads of a batch (segment starts): vr_n[r] = segments of read r, vr_off = their exclusive scan (vr_off[n_reads] = total), then the two lists
hipError_t launch_seed_segments(const SeedArgs &a, int32_t *vr_n, int64_t *vr_off, int32_t *vr_read, int32_t *vr_pos, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream);
hipError_t launch_build_planes(const uint32_t *occ, uint64_t n_blocks64, uint4 *planes, hipStream_t stream);
hipError_t launch_build_planes_k(FmView ix, int k_steps, uint4 *planes_k, uint64_t n_lines, uint64_t *t2_dev, uint64_t *t3_dev, hipStream_t stream);
hipError_t launch_planes2_check(const FmView &ix, uint64_t samples, uint64_t seed, unsigned long long *bad_dev, hipStream_t stream);
hipError_t launch_build_text(const uint8_t *pac, uint64_t l_pac, uint8_t *text, uint64_t n_bytes, hipStream_t stream);
hipError_t launch_build_qtab(const FmView &ix, int q, uint2 *t32, uint64_t *t64, hipStream_t stream);
hipError_t launch_rank_sa(const FmView &ix, const uint64_t *ks, int64_t n, uint64_t *occ4, uint64_t *sa_walk, uint64_t *sa_full, hipStream_t stream);
hipError_t launch_expand_sa(const FmView &ix, uint64_t n_sa, uint32_t *fsa32, uint64_t *fsa64, uint8_t *fsa40, hipStream_t stream);
hipError_t launch_sample_sa(const uint32_t *fsa32, const uint64_t *fsa64, uint64_t n_out, int shift, uint32_t *d32, uint64_t *d64, hipStream_t stream);

// Optional per-kernel timing of one workspace's launches (bench.py's per-kernel roofline entries): the entry points set the calling
// thread's current timer, the launchers bracket their kernels with its events on the launch stream, and the entry point reads the
// durations once the stream has been synchronised.
enum { KT_CHAIN = 0, KT_ALN_PAIR, KT_ALN_RESCUE, KT_ALN_PLAN_FAST, KT_ALN_PLAN, KT_ALN_PARTITION, KT_NW, KT_ALN_FINISH, KT_ALN_FINAL,
       KT_SAM_SIZE, KT_SAM_FORMAT, KT_FQ_PARSE, KT_FQ_MATERIALISE, KT_LOCATE_SORT, KT_ALN_TRIVIAL, KT_SLOTS = 16 };
// (a slot may be bracketed more than once per batch -- the NW kernels run for the batch and again for what comes back from the partition --: every
//  bracket takes the slot's next pair of events, up to kKtRing of them between two reads of the timer; the reader sums the pairs used)
constexpr int kKtRing = 4;
struct KernelTimer {
	hipEvent_t b[KT_SLOTS][kKtRing] = {}, e[KT_SLOTS][kKtRing] = {};
	int used[KT_SLOTS] = {};          // brackets closed since the last read (those beyond kKtRing re-use the last pair: counted, not timed twice)
	int open_at[KT_SLOTS] = {};
};
extern thread_local KernelTimer *kt_current;
inline void kt_begin(int slot, hipStream_t st)
{
	KernelTimer *k = kt_current;
	if (!k || !k->b[slot][0]) return;
	const int i = k->used[slot] < kKtRing ? k->used[slot] : kKtRing - 1;
	k->open_at[slot] = i;
	(void)hipEventRecord(k->b[slot][i], st);
}
inline void kt_end(int slot, hipStream_t st)
{
	KernelTimer *k = kt_current;
	if (!k || !k->e[slot][0]) return;
	(void)hipEventRecord(k->e[slot][k->open_at[slot]], st);
	if (k->used[slot] < kKtRing) k->used[slot]++;
}
struct KtUse {                     // the calling thread's launches are timed by `k` (null: not at all) while this object lives
	KernelTimer *prev;
	explicit KtUse(KernelTimer *k) : prev(kt_current) { kt_current = k; }
	~KtUse() { kt_current = prev; }
};

struct ChainArgs {
	const int64_t *read_off;     // for rlen
	int64_t n_reads;
	const int64_t *seed_off;
	const kg_seed *seeds;        // per read sorted by the seeding mode's comparator
	const int64_t *contig_end;   // ChrLocMap keys (last coordinate of every strand copy), ascending
	int n_ends;
	int pacbio, max_gaps;
	int32_t *n_cands;            // [n_reads + 1], the last entry stays 0 (scan tail)
	int32_t *used;               // [n_reads + 1] candidate seeds written per read
	kg_candidate *cands;         // sparse: candidate c of read r at seed_off[r] + c
	kg_seed *cand_seeds;
	uint8_t *taken;              // PacBio: one flag per seed
	// dense outputs (read order), filled by the compaction pass
	int64_t *cand_off, *cseed_off;   // [n_reads + 1] exclusive scans of n_cands / used
	kg_candidate *dense_cands;
	kg_seed *dense_seeds;
};
hipError_t launch_chain_batch(const ChainArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream);

// a gap-closing job written on the device by the alignment stage (align_kernels.hip)
struct NwJobDesc {
	int64_t o1;           // offset of the read-side fragment in the read characters
	int64_t o2;           // text coordinate of the genome-side fragment
	int64_t ops;          // offset of the job's op string
	int32_t m, n;
};

constexpr int kNwQueueWords = 8;   // the NW kernels' queue block: every allocation of it and nw_reset_kernel use this one number
struct NwArgs {
	// descriptor mode (jobs written on the device by the alignment stage): desc[p] instead of the offset arrays, sequence 2
	// from the 2-bit text, the job count from device memory (n = capacity of the lists)
	const NwJobDesc *desc = nullptr;
	const uint8_t *text2 = nullptr;
	const unsigned long long *n_dev = nullptr;
	const char *f1;
	const int64_t *off1;
	const char *f2;
	const int64_t *off2;
	int64_t n;
	uint8_t *ops;
	int32_t *aln_len;
	// scratch: three class lists of n entries each, queue words, and for the wave-per-pair kernel
	// one direction-word slab per resident wave
	int32_t *big_list;
	unsigned long long *queue;   // [0..2] class list sizes, [3] class-2 work queue head
	uint32_t *dir_scratch;
	int64_t dir_words_per_wave;
	int big_waves;               // number of slabs = grid of the wave-per-pair kernel
	int big_lds_bytes;           // boundary column + sequence-1 codes for the longest pair
	int64_t gb_offset_words;     // > 0: fragments beyond kNwMaxLen -- boundary column and codes at this word offset of the wave's slab instead of the LDS
	// two tiers of the wave-per-pair kernel (tier_len > 0): pairs up to tier_len take a launch of their own whose LDS and slabs are sized for
	// tier_len -- many waves per CU -- instead of sharing the few waves a launch sized for the batch's longest pair can hold
	int tier_len = 0;
	int t1_waves = 0, t1_lds_bytes = 0;
	int64_t t1_dir_words_per_wave = 0;
	uint32_t *t1_dir_scratch = nullptr;
};

constexpr int kNwMaxLen = 7000;  // longest fragment the wave-per-pair kernel's 64 KB LDS holds
inline int nw_big_lds_bytes(int max_len) { return 8 * (max_len + 1) + ((max_len + 15) & ~15) + 16; }
inline int64_t nw_dir_words(int max_len) { return max_len > 128 ? 16ll * (((int64_t)max_len + 255) / 256) * ((int64_t)max_len + 64) : 8ll * ((int64_t)max_len + 64); }   // per stripe (64 K columns, K = 2 up to 128 columns, else 4) and anti-diagonal step 2 K 64-bit lane masks

hipError_t launch_nw_batch(const NwArgs &a, int n_cu, hipStream_t stream);

}  // namespace kg
