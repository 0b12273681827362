// cli.cpp -- command line of the reference (src/main.cpp:16-31, 106-190): same flags, same messages.
#include <sys/stat.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "mapper.hpp"

#include <algorithm>
#include <thread>

namespace kart {

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
	fprintf(stdout, "         -gpu INT      HIP device [0]\n");
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
		else if (p == "-gpu" && i + 1 < argc) opt.device = atoi(argv[++i]);
		else if (p == "-silent") opt.silent = true;
		else if (p == "-pacbio") opt.pacbio = true;
		else if (p == "-m") opt.multi_hit = true;
		else if (p == "-pair" || p == "-p") opt.paired = true;
		else if (p == "-v" || p == "--version") { fprintf(stdout, "kart v2.5.6\n\n"); return -1; }
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

int cli_main(int argc, char **argv, KernelBackend *(*make_backend)(const Options &, std::string &))
{
	Options opt;
	int rc = parse_cli(argc, argv, opt);
	if (rc < 0) return -rc - 1;
	fprintf(stdout, "Load the genome index files...\n");
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
	FILE *out = fopen(opt.out_name.c_str(), "w");
	if (!out) { fprintf(stderr, "Error! Cannot open file [%s]\n", opt.out_name.c_str()); delete kern; return 1; }
	if (opt.silent) fprintf(stdout, "Start read mapping...\n");
	time_t t0 = time(NULL);
	Stats st;
	run_mapping(opt, ref, *kern, out, st);
	fclose(out);
	bool paired = opt.paired || !opt.files2.empty();
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
		if (getenv("KART_AMD_VERBOSE")) fprintf(stdout, "mapping seconds (index load excluded): %.3f\n", st.map_seconds);
		if (getenv("KART_AMD_VERBOSE")) fprintf(stdout, "chunks re-mapped after EstDistance speculation: %lld\n", (long long)st.respeculated);
	}
	delete kern;
	return 0;
}

}  // namespace kart
