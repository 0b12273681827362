// mapper.cpp -- host pipeline: reads in, SAM out, kernels through KernelBackend (see mapper.hpp).  The code lives in topic
// fragments under detail/ that are included below into ONE translation unit (everything in them is file-local).
//
// Every block names the reference code whose observable behaviour it reproduces (paths relative
// to the reference tree).  The aim is byte-identical output to `kart -t 1`; where the reference
// has undefined behaviour the choice made here is stated next to the code.
#include "mapper.hpp"

#include <fcntl.h>
#include <immintrin.h>
#include <sched.h>
#include <signal.h>
#include <pthread.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <sys/vfs.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <atomic>
#include <climits>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <malloc.h>
#include <memory>
#include <mutex>
#include <string_view>
#include <thread>

namespace kart {

namespace {

// the pieces of the pipeline, in dependency order (one translation unit: everything below is file-local)
#include "detail/types.inc"
#include "detail/normal_pairs.inc"
#include "detail/kmer.inc"
#include "detail/gap_closing.inc"
#include "detail/report.inc"
#include "detail/pairing.inc"
#include "detail/sam.inc"
#include "detail/bam.inc"
#include "detail/reader.inc"
#include "detail/shard.inc"
#include "detail/chunk_state.inc"
#include "detail/writer.inc"
#include "detail/chunk_stages.inc"
#include "detail/pgzip.inc"
#include "detail/batch_reader.inc"
#include "detail/deferred.inc"
#include "detail/pipeline.inc"
#include "detail/stream.inc"

}  // namespace

// ----------------------------------------------------------------------------------------------
// public entry points
// ----------------------------------------------------------------------------------------------
bool RefData::load(const std::string &prefix, std::string &err, int threads)
{
	FILE *fp = fopen((prefix + ".ann").c_str(), "r");
	if (!fp) { err = "cannot read " + prefix + ".ann"; return false; }
	long long l_pac;
	int n_seqs;
	unsigned seed;
	if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(fp); err = "bad .ann header"; return false; }
	genome_size = l_pac;
	two_genome_size = 2 * genome_size;
	int64_t total = 0;
	for (int i = 0; i < n_seqs; ++i) {
		unsigned gi;
		char name[1024];
		long long offv;
		int len, n_ambs, ch;
		if (fscanf(fp, "%u%1023s", &gi, name) != 2) { fclose(fp); err = "bad .ann record"; return false; }
		while ((ch = fgetc(fp)) != '\n' && ch != EOF) {}
		if (fscanf(fp, "%lld%d%d", &offv, &len, &n_ambs) != 3) { fclose(fp); err = "bad .ann record"; return false; }
		Contig c;
		c.name = name;
		c.len = len;
		c.fwd_start = total;
		total += len;
		c.rev_start = two_genome_size - total;
		chr_end[c.fwd_start + c.len - 1] = i;
		chr_end[c.rev_start + c.len - 1] = i;
		contigs.push_back(c);
	}
	fclose(fp);
	std::vector<unsigned char> pac;
	if (!slurp(prefix + ".pac", pac) || (int64_t)pac.size() < genome_size / 4 + 1) { err = "cannot read " + prefix + ".pac"; return false; }
	// both strands as characters (src/bwt_index.cpp:242-258), one .pac byte = four bases at a time; the 2L bytes
	// are first touched by the decoding threads themselves (6.2 GB for hg38)
	// random access all over 6.2 GB (hg38): transparent huge pages keep the report stage out of the page walker
	{
		void *mem = nullptr;
		size_t bytes = (((size_t)two_genome_size + 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
		if (posix_memalign(&mem, (size_t)2 << 20, bytes) != 0) { err = "out of memory for the reference sequence"; return false; }
		madvise(mem, bytes, MADV_HUGEPAGE);
		seq.reset((char *)mem);
	}
	seq[(size_t)two_genome_size] = '\0';
	std::vector<uint32_t> fw4(256), rc4(256);
	for (int v = 0; v < 256; ++v) {
		char f[4], r[4];
		for (int j = 0; j < 4; ++j) {
			int b = (v >> ((3 - j) << 1)) & 3;       // base j of the byte (first base in the top bits)
			f[j] = "ACGT"[b];
			r[3 - j] = "TGCA"[b];                    // the reverse strand runs the other way
		}
		memcpy(&fw4[(size_t)v], f, 4);
		memcpy(&rc4[(size_t)v], r, 4);
	}
	int64_t whole = genome_size >> 2;                // bytes whose four bases all exist
	int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, whole >> 20));
	std::vector<std::thread> pool;
	char *out = seq.get();
	for (int t = 0; t < nt; ++t)
		pool.emplace_back([&, t]() {
			for (int64_t i = whole * t / nt, e = whole * (t + 1) / nt; i < e; ++i) {
				memcpy(out + (i << 2), &fw4[pac[(size_t)i]], 4);
				memcpy(out + (two_genome_size - (i << 2) - 4), &rc4[pac[(size_t)i]], 4);
			}
		});
	for (std::thread &th : pool) th.join();
	static const char fw[4] = {'A', 'C', 'G', 'T'}, rc[4] = {'T', 'G', 'C', 'A'};
	for (int64_t f = whole << 2; f < genome_size; ++f) {
		int b = pac[(size_t)(f >> 2)] >> ((~f & 3) << 1) & 3;
		seq[(size_t)f] = fw[b];
		seq[(size_t)(two_genome_size - 1 - f)] = rc[b];
	}
	return true;
}

int run_mapping(const Options &opt, const RefData &ref, KernelBackend &kern, FILE *out, Stats &stats)
{
	// keep freed memory inside the arenas: the per-read strings/vectors of one batch are reused by the next,
	// and handing pages back to the kernel serialises every worker on the address-space lock
	mallopt(M_MMAP_THRESHOLD, 1 << 30);
	mallopt(M_TRIM_THRESHOLD, -1);
	mallopt(M_TOP_PAD, 64 << 20);
	double t_begin = now_s();
	struct rusage ru0;
	getrusage(RUSAGE_SELF, &ru0);
	g_sections = getenv("KART_AMD_VERBOSE") != nullptr;
	{ const char *uf = getenv("KART_AMD_UNSET_FLAG"); g_unset_flag = uf ? atoi(uf) : 0; }      // (per run: a session maps several)
	run_error_reset();
	g_check_align = getenv("KART_AMD_CHECK_ALIGN") != nullptr;
	// (a sharded run: every process takes L3 domains of its own)
	g_io_cpus = detect_io_cpus(opt.shard_count > 1 ? opt.shard_rank : -1);
	// The stream's lane threads (they pread the input into the staging buffers and drive the device) take the NEXT L3 domain, the
	// writers keep theirs: 34.2 / 34.4 -> 41.3 / 41.7 M mapped reads/s at 100 M reads (profiles/r04zi_ab_lane_cpus.log).  Round 3 saw
	// no gain from it -- four lanes then, reading the input through its mapping: their page faults met the writers' at the address space's
	// lock wherever they ran; with pread and eight lanes what is left to share is the cores.  A sharded run keeps a process on one domain
	// (the next one is the next process's).  KART_AMD_LANE_CPUS=same: with the writers.
	g_lane_cpus = IoCpus();
	{
		const char *e = getenv("KART_AMD_LANE_CPUS");
		if ((!e || !strcmp(e, "next")) && g_io_cpus.valid && opt.shard_count <= 1) {
			g_lane_cpus = detect_io_cpus(-2);
			// (one domain only: "next" is the same one -- nothing gained, nothing lost)
		} else if ((!e || !strcmp(e, "next")) && g_io_cpus.valid && opt.shard_count > 1 && count_l3_domains() >= 2 * opt.shard_count) {
			// a sharded run on a machine with domains to spare: process r's writers on domain r, its lanes on domain r + N (by analogy
			// with the measurement above; not measured itself -- the pool has no multi-GPU node)
			g_lane_cpus = detect_io_cpus(opt.shard_rank + opt.shard_count);
		}
	}
	Ctx cx{opt, ref, kern, kern.min_seed_len()};
	cx.frag_service = opt.pacbio && kern.has_fragments();
	Options &o = const_cast<Options &>(opt);
	RunTotals tot;
	// one process per GPU (shard.inc): this process maps shard_rank of shard_count chunk ranges of the library
	Shard shard;
	shard.rank = opt.shard_rank; shard.count = std::max(1, opt.shard_count);
	if (shard.active()) {
		if (shard.rank < 0 || shard.rank >= shard.count || shard.count > Rendezvous::kMaxRanks) { fprintf(stderr, "Error! bad shard %d/%d\n", shard.rank, shard.count); return 1; }
		shard.rdv = rendezvous_open(opt.rendezvous);
		if (!shard.rdv) { fprintf(stderr, "Error! cannot map the rendezvous file [%s]\n", opt.rendezvous.c_str()); return 1; }
		stats.sharded = true;
	}
	FILE *shard_out = nullptr;                           // the output file as a later shard opens it
	// header: @PG first, then @SQ, no @HD (src/Mapping.cpp:664-675)
	if (opt.bam)
		for (size_t i = 0; i < ref.contigs.size(); ++i) cx.bam_ref_id[ref.contigs[i].name] = (int)i;
	if (shard.rank == 0) {
		std::string header = "@PG\tID:kart\tPN:Kart\tVN:2.5.6\n";
		for (size_t i = 0; i < ref.contigs.size(); ++i) header += "@SQ\tSN:" + ref.contigs[i].name + "\tLN:" + std::to_string((long long)ref.contigs[i].len) + "\n";
		if (opt.bam) {                                  // src/Mapping.cpp:676-680: the same text, as the BAM header
			std::string blocks;
			bgzf_append(bam_header(ref, header), blocks);
			fwrite(blocks.data(), 1, blocks.size(), out);
		} else fwrite(header.data(), 1, header.size(), out);
	}
	for (size_t lib = 0; lib < opt.files1.size(); ++lib) {
		const std::string &f1 = opt.files1[lib];
		bool gz = f1.size() >= 2 && f1.substr(f1.find_last_of('.') + 1) == "gz";   // src/Mapping.cpp:688
		cx.fastq = is_fastq(f1);
		Source src;
		Input &in1 = src.in1, &in2 = src.in2;
		bool want_fast = !gz && cx.fastq && !getenv("KART_AMD_NO_MMAP");
		if (gz) in1.gz = gzopen(f1.c_str(), "rb"); else in1.fp = fopen(f1.c_str(), "r");
		bool sep = false;
		if (opt.files1.size() == opt.files2.size()) {
			sep = true;
			o.paired = true;
			const std::string &f2 = opt.files2[lib];
			if (cx.fastq != is_fastq(f2)) {
				fprintf(stdout, "Error! %s and %s are with different format...\n", f1.c_str(), f2.c_str());
				continue;
			}
			if (gz) in2.gz = gzopen(f2.c_str(), "rb"); else in2.fp = fopen(f2.c_str(), "r");
		}
		if (!in1.fp && !in1.gz) continue;
		if (sep && !in2.fp && !in2.gz) continue;
		src.sep = sep;
		src.fast = want_fast && src.m1.open(f1) && (!sep || src.m2.open(opt.files2[lib]));
		src.gzfast = gz && cx.fastq && !getenv("KART_AMD_NO_MMAP");
		if (src.gzfast) {
			src.g1.f = in1.gz; src.g2.f = in2.gz; gzbuffer(in1.gz, 1 << 20); if (in2.gz) gzbuffer(in2.gz, 1 << 20);
			src.g1.path = f1; if (sep) src.g2.path = opt.files2[lib];
			// bgzip-ped files are inflated member by member on several threads (both mate files are filled side by side: half each)
			const int per_file = std::max(1, opt.threads / (sep ? 2 : 1));
			if (!src.g1.try_bgzf(f1.c_str(), per_file)) src.g1.try_pgz(f1.c_str(), per_file);
			if (sep && !src.g2.try_bgzf(opt.files2[lib].c_str(), per_file)) src.g2.try_pgz(opt.files2[lib].c_str(), per_file);
		}
		if (shard.active()) {
			// only a single library of plain 4-line FASTQ is split; anything else is mapped by shard 0 alone
			bool splittable = src.fast && opt.files1.size() == 1;
			plan_shard(cx, src, shard, splittable);
			if (shard.solo && shard.rank > 0) {          // nothing to map here: publish an empty shard once its predecessor is done
				Rendezvous *rv = shard.rdv;
				shard.wait([&]() { return rv->done[shard.rank - 1].load() != 0; }, "waiting for the previous shard");
				rv->paired_end[shard.rank].store(rv->paired_end[shard.rank - 1].load());
				rv->distance_end[shard.rank].store(rv->distance_end[shard.rank - 1].load());
				rv->done[shard.rank].store(1);
				rv->written[shard.rank].store(1);
				continue;
			}
		}
		double tl = now_s();
		map_library(cx, src, out, stats, tot, shard, shard_out);
		tot.t_lib += now_s() - tl;
	}
	const bool last_writer = !shard.active() || (shard.solo ? shard.rank == 0 : shard.rank == shard.count - 1);
	const size_t bam_eof_bytes = 28;                     // (an empty BGZF block)
	if (opt.bam && last_writer) {                       // the empty BGZF block that marks the end of the file (SAMv1 4.1.2)
		std::string eof_block;
		bgzf_append_block((const unsigned char *)"", 0, eof_block);
		FILE *to = shard_out ? shard_out : out;          // (a later shard's stream stands right behind its own text)
		if (to) fwrite(eof_block.data(), 1, eof_block.size(), to);
	}
	if (shard_out) fclose(shard_out);
	if (shard.active()) {
		// the run is complete when every shard's text is in the file
		Rendezvous *rv = shard.rdv;
		if (shard.rank == 0 && opt.parts)
			shard.wait([&]() { for (int q = 0; q < shard.count; ++q) if (!rv->written[q].load()) return false; return true; }, "waiting for the other shards");
		else if (shard.rank == 0) {
			shard.wait([&]() { for (int q = 0; q < shard.count; ++q) if (!rv->written[q].load()) return false; return true; }, "waiting for the other shards");
			// every shard's text is in place and nobody extends the file any more (the writers grow it in large steps): its exact size
			int64_t total = rv->header_bytes.load() + (opt.bam ? (int64_t)bam_eof_bytes : 0);
			for (int q = 0; q < shard.count; ++q) total += rv->out_bytes[q].load();
			struct stat sb;
			if (out && fstat(fileno(out), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > total) {
				fflush(out);
				if (ftruncate(fileno(out), (off_t)total) != 0) perror("ftruncate");
			}
		}
	}
	stats.paired = tot.iPaired;
	stats.distance = tot.iDistance;
	stats.map_seconds = now_s() - t_begin;
	if (getenv("KART_AMD_VERBOSE"))
		fprintf(stdout, "stage seconds: unhidden read+encode+seed %.2f (seed calls %.2f) | finish+format(k-1) with chain+pair+plan(k) %.2f | nw %.2f | commit %.2f (of it formatting into the output %.2f) | writer drain %.2f | waiting for the long-read report %.2f | libraries %.2f of %.2f\n",
		        tot.t_read, tot.t_seed, tot.t_a, tot.t_nw, tot.t_commit, tot.t_format, tot.t_drain, tot.t_long_wait, tot.t_lib, stats.map_seconds);
	if (getenv("KART_AMD_VERBOSE")) {
		struct rusage ru1;
		getrusage(RUSAGE_SELF, &ru1);
		auto secs = [](const timeval &a, const timeval &b) { return (double)(a.tv_sec - b.tv_sec) + 1e-6 * (double)(a.tv_usec - b.tv_usec); };
		fprintf(stdout, "cpu seconds of the mapping phase: user %.2f, system %.2f (wall %.2f)\n", secs(ru1.ru_utime, ru0.ru_utime), secs(ru1.ru_stime, ru0.ru_stime), stats.map_seconds);
	}
	if (getenv("KART_AMD_VERBOSE") && shard.active())
		fprintf(stdout, "shard %d/%d: %lld reads in %.3f s | waited %.3f s for the totals of the shard before | settled in %.3f s (%lld chunks mapped again, %lld chunks of text written again) | writer drain %.3f s\n", shard.rank, shard.count,
		        (long long)stats.total_reads, stats.map_seconds, tot.t_shard_wait, tot.t_shard_settle, (long long)stats.respeculated, (long long)stats.rewritten_chunks, tot.t_drain);
	if (getenv("KART_AMD_VERBOSE") && g_frag_total.load() > 0)
		fprintf(stdout, "fragment pairs (GenerateNormalPairAlignment) aligned by the device: %lld, of them handed back and planned here: %lld\n", (long long)g_frag_total.load(), (long long)g_frag_back.load());
	if (getenv("KART_AMD_VERBOSE")) fprintf(stdout, "i/o threads: %s (%d CPUs)\n", g_io_cpus.valid ? "kept on the CPUs of one last-level cache" : "not pinned", g_io_cpus.count);
	if (getenv("KART_AMD_VERBOSE")) fprintf(stdout, "device report: %lld reads decided on the device, %lld mapped by the host stages\n", (long long)tot.dev_reads, (long long)tot.host_reads);
	if (getenv("KART_AMD_VERBOSE")) fprintf(stdout, "device report: read_batch total %.3f s (line index %.3f, views %.3f, views + chunk assembly %.3f, materialise + characters %.3f) | %s\n", 1e-9 * (double)g_read_ns.load(),
		        1e-9 * (double)g_read_part_ns[0].load(), 1e-9 * (double)g_read_part_ns[1].load(), 1e-9 * (double)g_read_part_ns[2].load(), 1e-9 * (double)g_read_part_ns[3].load(), kern.align_diagnostics().c_str());
	if (g_check_align) fprintf(stdout, "CHECK_ALIGN: %lld device records compared with the host's text, %lld differ\n", (long long)g_check_n.load(), (long long)g_check_bad.load());
	if (g_sections) {
		fprintf(stdout, "worker thread-seconds:");
		for (int i = 0; i < 6; ++i) fprintf(stdout, " %s %.2f%s", g_sec_name[i], 1e-9 * (double)g_sec_ns[i].load(), i < 5 ? " |" : "\n");
	}
	return run_failed() ? 1 : 0;
}

std::string run_error_message()
{
	std::lock_guard<std::mutex> lk(g_run_err_mu);
	return g_run_err;
}

FILE *open_output(const std::string &path)
{
	// measurement aid (bench.py's gpu_pipeline leg): the run's text goes nowhere -- the device pipeline and the copies back to the host run as
	// usual, what is left out is the host's copy into the output file's fresh pages
	if (getenv("KART_AMD_OUTPUT_NULL")) return fopen("/dev/null", "w");
	struct stat sb;
	bool regular = stat(path.c_str(), &sb) != 0 || S_ISREG(sb.st_mode);
	return fopen(path.c_str(), regular ? "w+" : "w");
}

bool shard_totals(const std::string &rendezvous, int shard_count, Stats &sum)
{
	Rendezvous *rv = rendezvous_open(rendezvous);
	if (!rv) return false;
	sum = Stats();
	bool ok = rv->failed.load() == 0;
	for (int q = 0; q < shard_count && q < Rendezvous::kMaxRanks; ++q) {
		if (!rv->done[q].load()) ok = false;
		sum.total_reads += rv->total_reads[q].load(); sum.unmapped += rv->unmapped[q].load(); sum.unique += rv->unique[q].load();
		sum.respeculated += rv->respec[q].load();
	}
	// iPaired / iDistance are running totals: the last shard's end values are the run's
	sum.paired = rv->paired_end[shard_count - 1].load();
	sum.distance = rv->distance_end[shard_count - 1].load();
	munmap(rv, sizeof(Rendezvous));
	return ok;
}

void shard_mark_failed(const std::string &rendezvous, int rank)
{
	if (Rendezvous *rv = rendezvous_open(rendezvous)) {
		rv->failed.store(1);
		// (a process that fails while it holds the output file's turn must not keep the others out of it)
		int32_t mine = rank + 1;
		if (rank >= 0) rv->file_turn.compare_exchange_strong(mine, 0);
		else rv->file_turn.store(0);
		munmap(rv, sizeof(Rendezvous));
	}
}

}  // namespace kart
