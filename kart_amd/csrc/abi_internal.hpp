// abi_internal.hpp -- handles behind the C ABI (include/kart_amd.h), shared by abi.hip and abi_stream.hip.
#pragma once
#include "seed_kernels.hpp"
#include "align_kernels.hpp"

#include <mutex>
#include <string>
#include <vector>

#define KG_INTERNAL __attribute__((visibility("hidden")))
KG_INTERNAL int kg_fail(int code, const char *fmt, ...);      // records the thread's kg_last_error() message, returns `code`

#define HIP_TRY(expr)                                                                                     \
	do {                                                                                                  \
		hipError_t e_ = (expr);                                                                           \
		if (e_ != hipSuccess) return kg_fail(e_ == hipErrorOutOfMemory ? KG_ERR_NOMEM : KG_ERR_NO_DEVICE, \
		                                     "%s: %s", #expr, hipGetErrorString(e_));                     \
	} while (0)

using namespace kg;

struct ContigRec {
	std::string name;
	int64_t fwd_start, rev_start, len;
};

// Scratch of one in-flight kg_nw_batch* call.  Cached in the index handle and recycled once the event
// recorded behind the call's last kernel has completed: steady state allocates nothing, and the
// device-pointer entry stays asynchronous.  (hipMallocAsync/hipFreeAsync were used first; with calls of
// varying size on the null stream the recycled pool blocks produced intermittently empty work lists.)
struct NwScratch {
	int32_t *lists = nullptr;
	size_t list_words = 0;
	unsigned long long *queue = nullptr;
	uint32_t *dir = nullptr;
	size_t dir_words = 0;
	hipEvent_t done = nullptr;
	bool busy = false;
	bool pending = false;     // acquired, but the event behind its kernels is not recorded yet: `done` still reports the PREVIOUS use
	hipStream_t last_stream = nullptr;   // where its last kernels were enqueued: the next use on the SAME stream needs no wait (stream order)
	// staging of the host-buffer entry (kg_nw_batch): inputs, offsets and outputs, grown on demand
	char *io = nullptr;
	size_t io_bytes = 0;
	bool io_busy = false;
	hipStream_t io_stream = nullptr;   // kg_nw_batch's copies and kernels (created with the staging block)
};

struct kg_index {
	std::mutex nw_mu;
	std::vector<NwScratch *> nw_pool;
	int device = 0;
	int n_cu = 256;
	int sa_mode = KG_SA_SAMPLED;
	FmView view{};
	int64_t l_pac = 0;
	uint64_t n_sa = 0;
	std::vector<ContigRec> contigs;
	std::vector<uint8_t> pac;   // forward strand, 2 bits/base (host copy)
	// device allocations
	uint32_t *d_occ = nullptr;
	uint4 *d_planes = nullptr;
	uint4 *d_planes2 = nullptr;        // two-step rank structure (fm_device.hpp)
	uint4 *d_planes3 = nullptr;        // three-step rank structure
	void *d_qtab = nullptr;
	uint64_t *d_sa = nullptr;
	void *d_fsa = nullptr;
	void *d_dsa = nullptr;          // KG_SA_DENSE4 / 8: every 4th / 8th entry of the expansion
	uint8_t *d_text = nullptr;
	uint8_t *d_pac = nullptr;
	int64_t *d_contig_end = nullptr;   // ChrLocMap keys, ascending
	int n_ends = 0;
	int32_t *d_end_chr = nullptr;      // contig of every key
	int64_t *d_chr_tab = nullptr;      // [3 * n_contigs]: FowardLocation, ReverseLocation, len
	uint8_t *d_mapq_tab = nullptr;     // EvaluateMAPQ's libm branch, tabulated (kg_align_batch)
	uint64_t device_bytes = 0;
};

struct kg_workspace {
	kg_index *ix = nullptr;
	int64_t max_reads = 0, max_bases = 0, max_hits = 0;
	// device scratch
	Hit *d_hits = nullptr;
	uint64_t *d_packed = nullptr;
	int32_t *d_seeds_per_read = nullptr;
	unsigned long long *d_ctl = nullptr;
	void *d_scan_temp = nullptr;
	size_t scan_bytes = 0;
	uint32_t *d_sort_keys = nullptr;    // EXPERIMENT (KG_SORT_READS)
	bool single_steps = false;          // kg_workspace_set_single_steps
	void *d_sort_temp = nullptr;
	size_t sort_bytes = 0;
	// staging for the host-buffer entry point
	uint8_t *d_enc = nullptr;
	int64_t *d_read_off = nullptr;
	int64_t *d_seed_off = nullptr;
	kg_seed *d_seeds = nullptr;
	int64_t seed_capacity = 0;
	int64_t last_reads = 0, last_seeds = 0;   // batch the staging buffers currently hold (kg_seed_batch)
	// a group's workspace (kgi_seed_group): the batch is n segments of `group_stride` read slots, group_prefix = reads before each
	int group_segments = 0;
	// SensitiveMode's segment walks (abi.hip, kg_seed_batch_device): whether the last launch used them; batches left for which they stay off
	// after a hit-list overflow (kgi_seed_resident re-runs such a batch with one lane per read); how often that happened
	bool last_segmented = false;
	int segments_off_batches = 0;
	int64_t segment_fallbacks = 0;
	int64_t group_stride = 0, group_prefix[kMaxSeedSegments + 1] = {0};
	int32_t *group_read_len = nullptr;        // [max_reads] length of the read in every slot (0: empty)
	bool enc_borrowed = false;                // d_enc is a part of a group's array (a stream lane): not this workspace's to free
	kg_candidate *d_cands = nullptr;
	kg_seed *d_cand_seeds = nullptr;
	int32_t *d_n_cands = nullptr;
	uint8_t *d_taken = nullptr;
	int32_t *d_used = nullptr;
	int64_t *d_cand_off = nullptr, *d_cseed_off = nullptr;
	kg_candidate *d_dense_cands = nullptr, *h_cands = nullptr;
	kg_seed *d_dense_seeds = nullptr, *h_cand_seeds = nullptr;
	int64_t h_cand_capacity = 0;
	// The pinned arrays kg_candidates_batch / kg_align_batch hand out rotate through kRing sets, so that a result stays valid
	// while the next kRing - 1 batches go through the workspace (a pipelined caller keeps several batches in flight and need
	// not copy anything out)
	static constexpr int kRing = 4;
	kg_candidate *ring_cands[kRing] = {nullptr, nullptr, nullptr, nullptr};
	kg_seed *ring_seeds[kRing] = {nullptr, nullptr, nullptr, nullptr};
	int64_t ring_cand_capacity[kRing] = {0, 0, 0, 0};
	kg_aln_record *ring_records[kRing] = {nullptr, nullptr, nullptr, nullptr};
	int64_t ring_record_capacity[kRing] = {0, 0, 0, 0};
	int ring_at = 0, ring_rec_at = 0;
	int64_t cand_capacity = 0, ncand_capacity = 0;
	kg_seed *h_seeds = nullptr;     // pinned
	int64_t h_seed_capacity = 0;
	// alignment stage (kg_align_batch)
	int64_t last_cands = -1;            // candidates the last kg_candidates_batch left on the device
	int64_t last_cand_seeds = 0;        // ... and their seeds
	bool last_pacbio = false;           // ... chained for long reads (GenerateAlignmentCandidateForPacBioSeq)
	void *lr = nullptr;                 // scratch of the long-read report (abi_long.hip: kg_longread_batch)
	// SensitiveMode within a read in parallel (SeedArgs::vr_read): allocated with the first long-read batch
	int32_t *d_vr_n = nullptr, *d_vr_read = nullptr, *d_vr_pos = nullptr;
	int64_t *d_vr_off = nullptr;
	uint32_t *d_claim = nullptr;
	uint8_t *d_step = nullptr;
	int64_t vr_capacity = 0;
	bool last_ascii = false;            // the resident reads are characters (KG_INPUT_ASCII)
	void *d_aln_cand = nullptr;         // per-candidate state, one block
	int64_t aln_cand_capacity = 0;
	void *d_aln_read = nullptr;         // per-read state (host flags, records), one block
	int64_t aln_read_capacity = 0;
	kg_aln_record *h_records = nullptr; // pinned
	AlnSpill *d_spill = nullptr;
	NwJobDesc *d_jobs = nullptr;
	uint8_t *d_job_ops = nullptr;
	int32_t *d_job_len = nullptr;
	int64_t spill_capacity = 0, job_capacity = 0, ops_capacity = 0;
	void *d_plans = nullptr;            // partition plans + pieces, one block
	void *d_tasks = nullptr;            // rescue windows and the candidates they yield, one block
	int64_t task_capacity = 0;
	int64_t *d_chunk_off = nullptr;
	uint8_t *d_chunk_paired = nullptr;
	kg_chunk_stats *d_chunk_stats = nullptr;
	int chunk_capacity = 0;
	unsigned long long *d_aln_ctl = nullptr;
	hipStream_t stream = nullptr;
	hipEvent_t sync_ev = nullptr;       // kgi_sync: blocking wait on the stream
	unsigned long long *h_small = nullptr;   // pinned, 16 words: totals and flags read back between stages
	bool profiling = false;
	hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	KernelTimer kt;                     // per-kernel events of the launches made for this workspace (created with profiling; seed_kernels.hpp)
};


// ---- stage bodies shared by the batch entry points (abi.hip) and the FASTQ -> SAM stream (abi_stream.hip) ----------------------
extern "C" {
KG_INTERNAL hipError_t kgi_sync(kg_workspace *ws);
KG_INTERNAL int kgi_seed_resident(kg_workspace *ws, int mode, int min_seed_len, int occ_thr, int64_t n_reads, int64_t n_bases, int64_t *total_out);
KG_INTERNAL int kgi_seed_group(kg_workspace *ws, int mode, int min_seed_len, int occ_thr, int n_seg, int64_t stride, const int64_t *counts, int64_t *seed_base);
KG_INTERNAL int kgi_chain_resident(kg_workspace *ws, int pacbio, int max_gaps, int64_t totals[2]);
KG_INTERNAL int kgi_nw_launch(kg_index *ix, NwArgs &a, int64_t max_len, hipStream_t st);
KG_INTERNAL void kgi_long_release(kg_workspace *ws);  // abi_long.hip: the long-read report's scratch of this workspace
KG_INTERNAL void kgi_frag_release(kg_index *ix);      // abi_frag.hip: the fragment service's scratches of this index
KG_INTERNAL int kgi_align_resident(kg_workspace *ws, const int64_t *chunk_off, const uint8_t *chunk_paired, int n_chunks, int est_distance, int max_insert,
                       int max_gaps, int multi_hit, int unset_flag, int64_t host_record_capacity, AlnArgs &a);
}
