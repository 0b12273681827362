// cli.cpp -- command line of the reference (src/main.cpp:16-31, 106-190): same flags, same messages.
#include <sys/stat.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "mapper.hpp"

#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <ctime>
#include <thread>

namespace kart {

#include "knobs.inc"

static void usage(const char *prog)
{
	fprintf(stdout, "kart v2.5.6 (MI355X-native hot path; CLI of Hsin-Nan Lin & Wen-Lian Hsu's kart)\n\n");
	fprintf(stdout, "Usage: %s -i Index_Prefix -f <ReadFile_A1 ReadFile_B1 ...> [-f2 <ReadFile_A2 ReadFile_B2 ...>] -o Output\n\n", prog);
	fprintf(stdout, "Options: -t INT        number of threads [4]\n");
	fprintf(stdout, "         -f            files with #1 mates reads (format:fa, fq, fq.gz)\n");
	fprintf(stdout, "         -f2           files with #2 mates reads (format:fa, fq, fq.gz)\n");
	fprintf(stdout, "         -o            alignment filename in SAM format [output.sam]\n");
	fprintf(stdout, "         -bo           alignment filename in BAM format\n");
	fprintf(stdout, "         -m            output multiple alignments\n");
	fprintf(stdout, "         -g INT        max gaps (indels) [5]\n");
	fprintf(stdout, "         -p            paired-end reads are interlaced in the same file\n");
	fprintf(stdout, "         -pacbio       pacbio data\n");
	fprintf(stdout, "         -gpu INT[,INT..] HIP device [0]; a list runs one process per device on contiguous parts of the input\n");
	fprintf(stdout, "         -parts        with a device list: one output file per device, Output.0 Output.1 ... (their concatenation is the alignment file)\n");
	fprintf(stdout, "         -v            version\n\n");
}

// returns 0 to proceed, otherwise -(exit code + 1)
int parse_cli(int argc, char **argv, Options &opt)
{
	if (argc == 1 || strcmp(argv[1], "-h") == 0) { usage(argv[0]); return -1; }
	for (int i = 1; i < argc; i++) {
		std::string p = argv[i];
		if (p == "-i" && i + 1 < argc) opt.index_prefix = argv[++i];
		else if (p == "-f") {
			while (++i < argc && argv[i][0] != '-') opt.files1.push_back(argv[i]);
			i--;
		} else if (p == "-f2") {
			while (++i < argc && argv[i][0] != '-') opt.files2.push_back(argv[i]);
			i--;
		} else if (p == "-t" && i + 1 < argc) {
			if ((opt.threads = atoi(argv[++i])) <= 0) {
				fprintf(stdout, "Warning! Thread number should be a positive number!\n");
				opt.threads = 4;
			}
		} else if (p == "-g" && i + 1 < argc) {
			if ((opt.max_gaps = atoi(argv[++i])) < 0) opt.max_gaps = 0;
		} else if (p == "-o" && i + 1 < argc) { opt.bam = false; opt.out_name = argv[++i]; }
		else if (p == "-bo" && i + 1 < argc) { opt.bam = true; opt.out_name = argv[++i]; }
		else if (p == "-gpu" && i + 1 < argc) {          // one device, or a list a,b,c: one process per device, the input sharded
			opt.devices.clear();
			for (const char *q = argv[++i]; *q;) {
				opt.devices.push_back(atoi(q));
				while (*q && *q != ',') q++;
				if (*q == ',') q++;
			}
			if (opt.devices.empty()) opt.devices.push_back(0);
			opt.device = opt.devices[0];
		} else if (p == "-shard" && i + 1 < argc) {      // r/N: this process maps chunk range r of N (set by the launcher / bench.py)
			if (sscanf(argv[++i], "%d/%d", &opt.shard_rank, &opt.shard_count) != 2 || opt.shard_count < 1 || opt.shard_rank < 0 || opt.shard_rank >= opt.shard_count) {
				fprintf(stdout, "Error! -shard expects r/N with 0 <= r < N\n");
				return -2;
			}
		} else if (p == "-rendezvous" && i + 1 < argc) opt.rendezvous = argv[++i];
		else if (p == "-parts") opt.parts = true;
		else if (p == "-silent") opt.silent = true;
		else if (p == "-pacbio") opt.pacbio = true;
		else if (p == "-m") opt.multi_hit = true;
		else if (p == "-pair" || p == "-p") opt.paired = true;
		else if (p == "-v" || p == "--version") { fprintf(stdout, "kart v2.5.6\n\n"); return -1; }
		else if (p == "-knobs") {          // the environment variables the product reads (host/knobs.inc): none is needed, none changes a result
			fprintf(stdout, "%-28s %-4s %-7s %s\n", "variable", "kind", "where", "what (T tuning, A A/B aid, D diagnostics / test aid, M measurement aid)");
			for (const Knob &k : kKnobs) fprintf(stdout, "%-28s %-4s %-7s %s%s\n", k.name, k.kind, k.where, k.what, getenv(k.name) ? "   [set]" : "");
			return -1;
		}
		else {
			fprintf(stdout, "Error! Unknown parameter: %s\n", argv[i]);
			usage(argv[0]);
			return -2;
		}
	}
	if (opt.files1.empty()) {
		fprintf(stdout, "Error! Please specify a valid read input!\n");
		usage(argv[0]);
		return -2;
	}
	if (!opt.files2.empty() && opt.files1.size() != opt.files2.size()) {
		fprintf(stdout, "Error! Paired-end reads input numbers do not match!\n");   // reference src/main.cpp:184-190: lists both sets
		fprintf(stdout, "Read1:\n");
		for (const std::string &f : opt.files1) fprintf(stdout, "\t%s\n", f.c_str());
		fprintf(stdout, "Read2:\n");
		for (const std::string &f : opt.files2) fprintf(stdout, "\t%s\n", f.c_str());
		return -2;
	}
	struct stat s;
	bool ok = true;
	for (const std::string &f : opt.files1)
		if (stat(f.c_str(), &s) == -1) { ok = false; fprintf(stdout, "Cannot access file:[%s]\n", f.c_str()); }
	for (const std::string &f : opt.files2)
		if (stat(f.c_str(), &s) == -1) { ok = false; fprintf(stdout, "Cannot access file:[%s]\n", f.c_str()); }
	if (opt.out_name != "output.sam") {   // CheckOutputFileName, src/main.cpp:33-61
		if (stat(opt.out_name.c_str(), &s) == 0 && (s.st_mode & S_IFDIR)) { ok = false; fprintf(stdout, "Warning: %s is a directory!\n", opt.out_name.c_str()); }
		for (char ch : opt.out_name)
			if (!(isalnum((unsigned char)ch) || ch == '/' || ch == '.' || ch == '-' || ch == '_')) {
				ok = false;
				fprintf(stdout, "Warning: [%s] is not a valid filename!\n", opt.out_name.c_str());
				break;
			}
	}
	if (!ok) return -1;
	bool have_idx = !opt.index_prefix.empty();
	for (const char *ext : {".ann", ".amb", ".pac"})   // CheckBWAIndexFiles, src/GetData.cpp:222-238
		if (have_idx && !std::ifstream(opt.index_prefix + ext).good()) have_idx = false;
	if (!have_idx) {
		fprintf(stdout, "Error! Please specify a valid reference index!\n");
		usage(argv[0]);
		return -2;
	}
	return 0;
}

// one process: index + reference in, one shard (or all) of the input mapped, SAM out
static int run_one(const Options &opt, KernelBackend *(*make_backend)(const Options &, std::string &), Stats &st, bool quiet)
{
	if (!quiet) fprintf(stdout, "Load the genome index files...\n");
	RefData ref;
	std::string err;
	// the host copy of the reference (both strands as characters) is decoded while the device index is uploaded and its
	// rank / q-mer / suffix-array structures are built
	std::string ref_err;
	bool ref_ok = false;
	std::thread ref_loader([&]() { ref_ok = ref.load(opt.index_prefix, ref_err, std::max(1, opt.threads)); });
	KernelBackend *kern = make_backend(opt, err);
	ref_loader.join();
	if (!ref_ok) { fprintf(stdout, "\n\nError! Index files are corrupt! (%s)\n", ref_err.c_str()); delete kern; return 1; }
	if (!kern) { fprintf(stderr, "Error! %s\n", err.c_str()); return 1; }
	FILE *out = nullptr;
	if (opt.shard_rank == 0 || opt.parts) {                  // later shards open the file once shard 0 has created it (-parts: each its own)
		const std::string name = opt.parts && opt.shard_count > 1 ? opt.out_name + "." + std::to_string(opt.shard_rank) : opt.out_name;
		out = kart::open_output(name);
		if (!out) { fprintf(stderr, "Error! Cannot open file [%s]\n", name.c_str()); delete kern; return 1; }
	}
	if (opt.silent && !quiet) fprintf(stdout, "Start read mapping...\n");
	int rc = run_mapping(opt, ref, *kern, out, st);
	if (out) fclose(out);
	delete kern;
	return rc;
}

// -gpu a,b,c: one child process per device (forked before anything touches HIP), each mapping its shard; see detail/shard.inc
static int run_sharded(const Options &opt, KernelBackend *(*make_backend)(const Options &, std::string &), Stats &st)
{
	const int n = (int)opt.devices.size();
	char path[128];
	snprintf(path, sizeof(path), "/dev/shm/kart-amd-rdv-%d-%lld", (int)getpid(), (long long)time(NULL));
	unlink(path);
	fprintf(stdout, "Load the genome index files...\n");
	fflush(stdout);
	std::vector<pid_t> kids;
	for (int r = 0; r < n; ++r) {
		pid_t pid = fork();
		if (pid < 0) { perror("fork"); shard_mark_failed(path); break; }
		if (pid == 0) {
			Options o = opt;
			o.device = opt.devices[(size_t)r];
			o.devices.assign(1, o.device);
			o.shard_rank = r; o.shard_count = n; o.rendezvous = path;
			o.threads = std::max(1, opt.threads / n);        // -t is the budget of the whole run
			Stats s;
			int rc = run_one(o, make_backend, s, true);
			fflush(stdout);
			_exit(rc);
		}
		kids.push_back(pid);
	}
	if (opt.silent) fprintf(stdout, "Start read mapping...\n");
	int rc = (int)kids.size() == n ? 0 : 1;
	for (size_t left = kids.size(); left > 0; --left) {
		int status = 0;
		pid_t pid = wait(&status);
		if (pid < 0) break;
		if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { rc = 1; shard_mark_failed(path); }   // the others stop waiting for it
	}
	if (rc == 0 && !shard_totals(path, n, st)) rc = 1;
	unlink(path);
	return rc;
}

int cli_main(int argc, char **argv, KernelBackend *(*make_backend)(const Options &, std::string &))
{
	Options opt;
	int rc = parse_cli(argc, argv, opt);
	if (rc < 0) return -rc - 1;
	if (opt.shard_count > 1 && opt.rendezvous.empty()) { fprintf(stdout, "Error! -shard needs -rendezvous FILE\n"); return 1; }
	time_t t0 = time(NULL);
	Stats st;
	const bool launcher = opt.devices.size() > 1 && opt.shard_count == 1;
	rc = launcher ? run_sharded(opt, make_backend, st) : run_one(opt, make_backend, st, false);
	if (rc != 0) return rc;
	bool paired = opt.paired || !opt.files2.empty();
	if (opt.shard_count > 1) {     // one shard of a run someone else coordinates: that process reports the totals
		if (getenv("KART_AMD_VERBOSE")) fprintf(stdout, "shard %d/%d: %lld reads, mapping seconds %.3f, re-mapped chunks %lld\n", opt.shard_rank, opt.shard_count, (long long)st.total_reads, st.map_seconds, (long long)st.respeculated);
		return 0;
	}
	fprintf(stdout, "\rAll the %lld %s reads have been processed in %lld seconds.\n", (long long)st.total_reads, paired ? "paired-end" : "single-end", (long long)(time(NULL) - t0));
	if (st.total_reads > 0) {   // src/Mapping.cpp:736-741
		long long mapped = st.total_reads - st.unmapped;
		if (paired)
			fprintf(stdout, "\t# of total mapped sequences = %lld (sensitivity = %.2f%%)\n\t# of paired sequences = %lld (%.2f%%), average insert size = %d\n", mapped,
			        (int)(10000 * (1.0 * mapped / st.total_reads) + 0.5) / 100.0, (long long)st.paired, (int)(10000 * (1.0 * st.paired / st.total_reads) + 0.5) / 100.0,
			        (st.paired > 1 ? (int)(st.distance / (st.paired >> 1)) : 0));
		else
			fprintf(stdout, "\t# of total mapped sequences = %lld (sensitivity = %.2f%%)\n", mapped, (int)(10000 * (1.0 * mapped / st.total_reads) + 0.5) / 100.0);
		fprintf(stdout, "Alignment output: %s\n", opt.out_name.c_str());
		if (getenv("KART_AMD_VERBOSE") && !launcher) fprintf(stdout, "mapping seconds (index load excluded): %.3f\n", st.map_seconds);
		if (getenv("KART_AMD_VERBOSE")) fprintf(stdout, "chunks re-mapped after EstDistance speculation: %lld\n", (long long)st.respeculated);
	}
	return 0;
}

}  // namespace kart
