// mapper.hpp -- host side of the MI355X-native Kart pipeline (FASTQ -> SAM).
//
// Mirrors the reference's driver (reference src/Mapping.cpp, src/GetData.cpp, src/main.cpp) above
// the C ABI of include/kart_amd.h.  The three hot-path stages are batched:
//     seeding + chaining -> KernelBackend::seed_and_chain  (kg_seed_batch + kg_candidates_batch, HIP)
//     gap closing  -> KernelBackend::nw_batch    (kg_nw_batch,     HIP)
// everything else here is the reference's per-read control flow restated on the host (chaining,
// pairing with the EstDistance feedback, mate rescue, report, flags, MAPQ, SAM text), written to
// reproduce `kart -t 1` byte for byte.  The only KernelBackend in the product is the HIP one
// (hip_backend.cpp); a CPU backend exists solely under tests/ to exercise this host logic where
// no GPU is present.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/kart_amd.h"

namespace kart {

struct Options {
	std::string index_prefix;
	std::vector<std::string> files1, files2;
	std::string out_name = "output.sam";
	int threads = 4;          // accepted for CLI compatibility; output is always the -t 1 order
	int max_gaps = 5;         // -g   (reference src/main.cpp:92)
	int max_insert = 1500;    // MaxInsertSize (src/main.cpp:96)
	bool paired = false;      // -p / -f2
	bool pacbio = false;      // -pacbio
	bool multi_hit = false;   // -m
	bool silent = false;      // -silent
	bool bam = false;         // -bo: BAM instead of SAM (src/main.cpp:155-158)
	int device = 0;
	std::vector<int> devices;       // -gpu a,b,c: one process per listed device, the input sharded between them
	int shard_rank = 0, shard_count = 1;   // this process maps shard_rank of shard_count contiguous chunk ranges of the library ...
	std::string rendezvous;         // ... coordinating with the other processes through this shared-memory file
	bool parts = false;             // -parts: a sharded run writes one file per shard, <out>.<r>; `cat <out>.0 <out>.1 ...` is the single-process file.
	                                // One file takes ~10-13 GB/s of text on the 2-socket box however many processes write it: with several
	                                // GPUs that is the run's ceiling; parts have no such common ceiling.
	int sa_mode = KG_SA_AUTO;
	int64_t batch_reads = default_batch_reads();   // reads seeded per GPU call (a whole number of 4000-read chunks; KART_AMD_BATCH_READS overrides)
	int64_t stream_reads = default_stream_reads();   // reads per batch of the device's FASTQ-in / SAM-out stream (KART_AMD_STREAM_READS overrides)
	static int64_t default_stream_reads() { const char *e = getenv("KART_AMD_STREAM_READS"); long long v = e ? atoll(e) : 0; return v >= 4000 ? (int64_t)v : 1120000; }   // (280 chunks: four lanes' batches = 4.3 M reads per seeding launch, roofline.frac 0.157-0.159 in product; 1.0 M: 0.149-0.151, 1.2 M: 0.164, the FASTQ -> SAM rate alike within its noise, profiles/r04zf_ab_stream_reads.log)
	static int64_t default_batch_reads() { const char *e = getenv("KART_AMD_BATCH_READS"); long long v = e ? atoll(e) : 0; return v >= 4000 ? (int64_t)v : 400000; }
};

// The gap-closing jobs of one chunk, flat: fragments concatenated (read side f1, genome side f2) with
// offsets, results as op codes (KG_OP_*) at ops[o1[j] + o2[j] ...) with len[j] columns -- the same
// layout kg_nw_batch uses, so batching chunks is a handful of large copies.
struct NwJobs {
	std::string f1, f2;
	std::vector<int64_t> o1{0}, o2{0};
	std::vector<uint8_t> ops;
	std::vector<int32_t> len;
	int add(const char *a, int m, const char *b, int n)
	{
		f1.append(a, (size_t)m);
		f2.append(b, (size_t)n);
		o1.push_back((int64_t)f1.size());
		o2.push_back((int64_t)f2.size());
		return (int)o1.size() - 2;
	}
	size_t size() const { return o1.size() - 1; }
	void clear()
	{
		f1.clear(); f2.clear(); o1.assign(1, 0); o2.assign(1, 0); ops.clear(); len.clear();
	}
};

// FASTQ text in, SAM text out (kg_stream_*, include/kart_amd.h): `lanes` batches in flight on the device, the host only moves
// bytes.  The calls of one lane must not overlap; different lanes are driven from different threads.
struct StreamBackend {
	virtual ~StreamBackend() {}
	virtual int lanes() const = 0;
	virtual int64_t max_reads() const = 0;
	virtual int64_t max_window() const = 0;
	virtual char *staging(int lane, int file) = 0;                                   // page-locked, max_window() bytes
	virtual void upload(int lane, int file, int64_t from, int64_t to) = 0;           // asynchronous
	virtual bool parse(int lane, const kg_stream_window &w, kg_stream_parsed &out) = 0;   // false: the window does not fit the lane (the caller's own reader takes over)
	virtual void map(int lane, const kg_stream_params &p, kg_stream_result &out) = 0;
	// the records and candidates of reads [first, first + count) of the lane's mapped batch into the result's arrays (kg_stream_fetch): a result made
	// with p.fetch_all == 0 holds them only for what was fetched
	virtual void fetch(int lane, int64_t first, int64_t count) { (void)lane; (void)first; (void)count; }
	virtual bool timing(kg_stream_timing_t &t, bool reset) { (void)t; (void)reset; return false; }   // device time per stage since the last reset
	// seeding groups (kg_stream_group_absent): lanes [g * seed_group(), (g + 1) * seed_group()) seed their batches in ONE launch per round;
	// a lane without a batch for `rounds` rounds says so (< 0: until further notice, 0: it takes part again)
	virtual int seed_group() const { return 0; }
	virtual void group_absent(int lane, int rounds) { (void)lane; (void)rounds; }
};

// The fragment pairs of one chunk whose alignment GenerateNormalPairAlignment (src/tools.cpp:142-223) is to produce -- 8-mer partition,
// the -pacbio recursion, nw_alignment of the pieces -- as ONE batched call (kg_fragments_batch): read fragments concatenated (f1
// with offsets o1), genome fragments as text coordinates.  Results: op codes at ops[oo[j] ...), len[j] columns; status[j] != 0:
// outside the kernels' envelope, the host plans that pair itself.
struct FragJobs {
	std::string f1;
	std::vector<int64_t> o1{0}, g, oo;
	std::vector<int32_t> gl;
	std::vector<void *> owner;          // whoever waits for job j (the host's PairWork)
	// results: they stay in the backend's page-locked arrays (several sets in rotation: valid for the next few calls -- the stage that
	// reads them runs during the next call at the latest, pipeline.inc); ops of job j at ops[oo[j] ...]
	const uint8_t *ops = nullptr, *status = nullptr;
	const int32_t *len = nullptr;
	int64_t cols = 0;
	int add(const char *a, int m, int64_t gpos, int n, void *who)
	{
		f1.append(a, (size_t)m);
		o1.push_back((int64_t)f1.size());
		g.push_back(gpos); gl.push_back(n); oo.push_back(cols); owner.push_back(who);
		cols += m + n;
		return (int)g.size() - 1;
	}
	size_t size() const { return g.size(); }
	void clear() { f1.clear(); o1.assign(1, 0); g.clear(); oo.clear(); gl.clear(); owner.clear(); ops = status = nullptr; len = nullptr; cols = 0; }
};

struct KernelBackend {
	virtual ~KernelBackend() {}
	// GenerateNormalPairAlignment for the fragment pairs of several chunks in one call (fills ops / len / status of every part);
	// false: this backend has no such stage -- the host plans every pair itself
	virtual bool fragments_batch(std::vector<FragJobs *> &parts, bool pacbio, int max_gaps) { (void)parts; (void)pacbio; (void)max_gaps; return false; }
	virtual bool has_fragments() const { return false; }
	// a stream with at least this capacity (kept by the backend across runs), or null: the backend has no such path
	// seed_group > 1: one seeding launch over the batches of that many lanes (kg_stream_config::seed_group)
	virtual bool has_stream() const { return false; }          // stream() can succeed (asked before a gz library starts inflating towards it)
	virtual StreamBackend *stream(int64_t max_reads, int64_t max_window, int lanes, int seed_group = 0) { (void)max_reads; (void)max_window; (void)lanes; (void)seed_group; return nullptr; }
	// index constants the host needs
	virtual int min_seed_len() const = 0;
	// IdentifySeedPairs_{Fast,Sensitive}Mode + GenerateAlignmentCandidateFor{Illumina,PacBio}Seq for a batch: enc = the concatenated
	// read CHARACTERS (the backend applies EnCodeReadSeq), off[n+1].  Out: n_cands[r] candidates per read, stored densely in read order (those of read r start at
	// cand_off[r]), their seeds at cand_seeds[cands[].first ...].  The seeds themselves never leave the device.
	// `cands` / `cand_seeds` may point at the vectors (which the backend then fills) or at storage of the backend's own that
	// stays valid while the next three batches go through it.
	virtual void seed_and_chain(int mode, bool pacbio, int max_gaps, const uint8_t *enc, const std::vector<int64_t> &off,
	                            std::vector<int32_t> &n_cands, std::vector<int64_t> &cand_off, std::vector<kg_candidate> &own_cands,
	                            std::vector<kg_seed> &own_seeds, const kg_candidate *&cands, const kg_seed *&cand_seeds) = 0;
	// memory for the read characters of a batch (page-locked where the backend copies it to a device)
	virtual void *host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
	virtual void host_free(void *p) { free(p); }
	// nw_alignment for the jobs of several chunks in one call (fills ops/len of every part)
	virtual void nw_batch(std::vector<NwJobs *> &parts) = 0;
	// The per-read report of the batch the last seed_and_chain() call left on the device (kg_align_batch): one record per read (+ the chained ones of -m)
	// (kind KG_ALN_HOST = this pair is the host's) and the chunks' pairing statistics under `est`.  false: this backend has no
	// such stage -- the host maps every read itself.
	// (`records`: storage of the backend, valid while the next three batches go through it)
	virtual bool align(const std::vector<int64_t> &chunk_off, const std::vector<uint8_t> &chunk_paired, int est, int max_insert, int max_gaps,
	                   bool multi_hit, int unset_flag, const kg_aln_record *&records, std::vector<kg_chunk_stats> &chunk_stats)
	{
		(void)chunk_off; (void)chunk_paired; (void)est; (void)max_insert; (void)max_gaps; (void)multi_hit; (void)unset_flag; (void)records; (void)chunk_stats;
		return false;
	}
	// The same for long reads (-pacbio, kg_longread_batch; src/Mapping.cpp:513-530): one record per read of the batch the last
	// seed_and_chain(pacbio) call left on the device, the CIGAR strings in `cigar_pool` (records with cigar_len == KG_ALN_CIGAR_POOLED
	// hold {int64 offset, int32 bytes} in their first 12 CIGAR bytes); chunk_stats[c].unmapped / unique / host_pairs are counted from the
	// records.  false: the host maps every read itself.
	// `slot`: long_slot() right after the batch's seed_and_chain() call -- long-read batches alternate between two workspaces, and this call
	// may run on a thread of its own while the next batch is being seeded in the other one.  long_enabled(): the backend has the stage.
	virtual bool align_long(int slot, const std::vector<int64_t> &chunk_off, const kg_aln_record *&records, const char *&cigar_pool, std::vector<kg_chunk_stats> &chunk_stats)
	{
		(void)slot; (void)chunk_off; (void)records; (void)cigar_pool; (void)chunk_stats;
		return false;
	}
	virtual int long_slot() const { return 0; }
	virtual bool long_enabled() const { return false; }
	virtual bool long_overlap() const { return false; }       // two workspaces: align_long(slot) may run beside the next batch's seed_and_chain()
	// diagnostics of the stage above (why pairs came back for the host), empty when there is none
	virtual std::string align_diagnostics() { return std::string(); }
};

struct Contig {
	std::string name;
	int64_t fwd_start, rev_start, len;
};

// RestoreReferenceInfo (reference src/bwt_index.cpp:230-259): contigs, ChrLocMap, RefSequence
struct RefData {
	int64_t genome_size = 0, two_genome_size = 0;
	std::vector<Contig> contigs;
	std::map<int64_t, int> chr_end;     // last coordinate of each strand copy -> contig index
	std::unique_ptr<char[], void (*)(void *)> seq{nullptr, free};   // 2L + 1, forward then reverse complement (2 MB-aligned, huge pages)
	bool load(const std::string &prefix, std::string &err, int threads = 16);
};

struct Stats {
	int64_t total_reads = 0, unmapped = 0, unique = 0, paired = 0, distance = 0;
	double map_seconds = 0;     // first read in -> last SAM byte handed to the writer (index load excluded)
	int64_t respeculated = 0;   // chunks re-mapped because their speculated EstDistance did not hold
	int64_t rewritten_chunks = 0;   // (-parts, a later shard) chunks whose text was written a second time because settling changed a chunk in front of them
	int64_t stream_reads = 0;   // reads that went through the device's FASTQ-in / SAM-out stream
	kg_stream_timing_t device{};   // ... and what their batches cost on the device (HIP events on the lanes' streams, summed)
	double lane_seconds[6] = {0, 0, 0, 0, 0, 0};   // ... and what the lanes' host threads waited for / worked on (kh_stats_t::lane_seconds), summed over `lanes` threads
	int lanes = 0;
	bool sharded = false;       // the totals above are this process's shard only (kart::shard_totals() gives the run's)
};

// the run-wide totals of a sharded run, summed from the rendezvous block once every shard has finished
bool shard_totals(const std::string &rendezvous, int shard_count, Stats &sum);
void shard_mark_failed(const std::string &rendezvous, int rank = -1);   // rank >= 0: gives up that process's turn at the output file; -1: whoever holds it
std::string run_error_message();       // why the last run_mapping() returned non-zero (empty: no recorded reason)

// Mapping() of the reference: maps every input library and writes SAM to `out`.
// Returns 0 on success; `summary` receives the reference's end-of-run statistics.
int run_mapping(const Options &opt, const RefData &ref, KernelBackend &kern, FILE *out, Stats &stats);

// the output file: read-write on a regular file (the writer places chunks through shared mappings), write-only otherwise (pipes)
FILE *open_output(const std::string &path);

int parse_cli(int argc, char **argv, Options &opt);   // reference src/main.cpp:106-190; <0 = exit(code)

}  // namespace kart
