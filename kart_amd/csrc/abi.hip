// abi.hip -- the extern "C" boundary of libkart_amd.so (declared in include/kart_amd.h).
//
// Host side of the boundary: index files -> HBM, per-batch workspaces, kernel launches.  No
// CPU implementation of the kernels lives here: without a usable HIP device every entry point
// returns KG_ERR_NO_DEVICE.
#include "abi_internal.hpp"
#include <cmath>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace kg { KG_INTERNAL thread_local KernelTimer *kt_current = nullptr; }      // seed_kernels.hpp: the calling thread's kernel timer

namespace {

thread_local char g_err[512] = "";

}  // namespace

int kg_fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}
#define fail kg_fail

namespace {

// a whole file in memory, without the zero fill a std::vector would do first (the .bwt of hg38 is 3.1 GB)
struct FileBuf {
	std::unique_ptr<unsigned char[]> mem;
	size_t n = 0;
	const unsigned char *data() const { return mem.get(); }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
};

bool read_file(const std::string &path, FileBuf &buf)
{
	FILE *fp = fopen(path.c_str(), "rb");
	if (!fp) return false;
	fseek(fp, 0, SEEK_END);
	long sz = ftell(fp);
	fseek(fp, 0, SEEK_SET);
	buf.mem.reset(new unsigned char[(size_t)sz + 1]);
	buf.n = (size_t)sz;
	size_t got = 0;
	while (got < (size_t)sz) {
		size_t x = fread(buf.mem.get() + got, 1, (size_t)sz - got, fp);
		if (x == 0) break;
		got += x;
	}
	fclose(fp);
	return got == (size_t)sz;
}

}  // namespace

extern "C" {

const char *kg_last_error(void) { return g_err; }

int kg_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// Replaces bwa_idx_load + RestoreReferenceInfo (reference src/bwt_index.cpp:16-36, 47-71, 103-122,
// 148-160, 230-259): same five files, same derived quantities, but the result lives in HBM.
int kg_index_load(const char *prefix, int device, int sa_mode, kg_index **out)
{
	if (!prefix || !out) return fail(KG_ERR_ARG, "kg_index_load: null argument");
	*out = nullptr;
	int ndev = kg_device_count();
	if (ndev <= 0) return fail(KG_ERR_NO_DEVICE, "kg_index_load: no HIP device available (this library has no CPU path)");
	if (device < 0 || device >= ndev) return fail(KG_ERR_ARG, "kg_index_load: device %d out of range (have %d)", device, ndev);
	HIP_TRY(hipSetDevice(device));

	std::string pre(prefix);
	FileBuf bwt, sa, pac;
	{
		// the three files are read side by side (page-cache copies of 3.1 + 1.6 + 0.8 GB for hg38)
		bool ok_sa = false, ok_pac = false;
		std::thread t_sa([&]() { ok_sa = read_file(pre + ".sa", sa); }), t_pac([&]() { ok_pac = read_file(pre + ".pac", pac); });
		bool ok_bwt = read_file(pre + ".bwt", bwt);
		t_sa.join(); t_pac.join();
		if (!ok_bwt || bwt.size() < 40 + 64) return fail(KG_ERR_IO, "cannot read %s.bwt", prefix);
		if (!ok_sa || sa.size() < 56) return fail(KG_ERR_IO, "cannot read %s.sa", prefix);
		if (!ok_pac || pac.empty()) return fail(KG_ERR_IO, "cannot read %s.pac", prefix);
	}

	// every early return below releases what was uploaded so far
	std::unique_ptr<kg_index, void (*)(kg_index *)> ix(new kg_index(), kg_index_destroy);
	ix->device = device;
	ix->sa_mode = sa_mode;
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

	FmView &v = ix->view;
	if (bwt.size() < 40) return fail(KG_ERR_IO, "%s.bwt is truncated", prefix);
	if (sa.size() < 56) return fail(KG_ERR_IO, "%s.sa is truncated", prefix);
	memcpy(&v.primary, bwt.data(), 8);
	v.L2[0] = 0;
	memcpy(&v.L2[1], bwt.data() + 8, 32);
	v.seq_len = v.L2[4];
	if (sa_mode == KG_SA_AUTO) {
		// below 2^32 text symbols the whole index is a few hundred MB: everything.  Above: the 5-byte suffix array with the full q-mer table and the
		// triple planes (KG_SA_FULL40_WIDE, ~9.5 bytes per text symbol more than the compact index: 150 GB for a human genome) where the
		// device keeps 64 GB for the workspaces behind it -- a 288 GB device running one process does --, else the compact index
		sa_mode = KG_SA_FULL;
		if (v.seq_len >= 0xFFFFFFFFull) {
			size_t free_b = 0, total_b = 0;
			const double wide_bytes = 24.5 * (double)v.seq_len;          // planes 1 + planes2 2.29 + planes3 9.14 + SA 5 + text 0.25 + table ~5.5 + the .bwt / .sa images in passing
			sa_mode = KG_SA_FULL40;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b > wide_bytes + (double)((size_t)64 << 30) && !getenv("KG_AUTO_COMPACT")) sa_mode = KG_SA_FULL40_WIDE;
		}
		ix->sa_mode = sa_mode;
	}
	size_t n_words = (bwt.size() - 40) / 4;
	{
		// 16 symbols per word, 8 count words in front of every 128-symbol block, one trailing count record
		// (reference src/BWT_Index/bwtindex.c:51-75)
		uint64_t need = (v.seq_len + 15) / 16 + 8 * ((v.seq_len + 127) / 128) + 8;
		if (v.seq_len == 0 || v.primary > v.seq_len || (uint64_t)n_words < need)
			return fail(KG_ERR_IO, "%s.bwt is truncated or malformed (%llu words for %llu symbols, need %llu)", prefix, (unsigned long long)n_words,
			            (unsigned long long)v.seq_len, (unsigned long long)need);
	}
	uint64_t sa_intv = 0, sa_len = 0;
	memcpy(&sa_intv, sa.data() + 40, 8);
	memcpy(&sa_len, sa.data() + 48, 8);
	if (sa_intv != 32) return fail(KG_ERR_IO, "%s.sa: unsupported SA interval %llu (expected 32)", prefix, (unsigned long long)sa_intv);
	if (sa_len != v.seq_len) return fail(KG_ERR_IO, "%s: .sa and .bwt disagree on the sequence length", prefix);
	ix->n_sa = (v.seq_len + 32) / 32;
	if ((sa.size() - 56) / 8 < ix->n_sa - 1) return fail(KG_ERR_IO, "%s.sa is truncated", prefix);

	// .ann: "l_pac n_seqs seed" then per contig "gi name [comment]" / "offset len n_ambs"
	{
		FILE *fp = fopen((pre + ".ann").c_str(), "r");
		if (!fp) return fail(KG_ERR_IO, "cannot read %s.ann", prefix);
		long long l_pac = 0;
		int n_seqs = 0;
		unsigned seed = 0;
		if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(fp); return fail(KG_ERR_IO, "%s.ann: bad header", prefix); }
		ix->l_pac = l_pac;
		int64_t total = 0;
		for (int i = 0; i < n_seqs; ++i) {
			unsigned gi;
			char name[1024];
			long long off;
			int len, n_ambs, ch;
			if (fscanf(fp, "%u%1023s", &gi, name) != 2) { fclose(fp); return fail(KG_ERR_IO, "%s.ann: bad contig record %d", prefix, i); }
			while ((ch = fgetc(fp)) != '\n' && ch != EOF) {}
			if (fscanf(fp, "%lld%d%d", &off, &len, &n_ambs) != 3) { fclose(fp); return fail(KG_ERR_IO, "%s.ann: bad contig record %d", prefix, i); }
			ContigRec c;
			c.name = name;
			c.len = len;
			c.fwd_start = total;
			total += len;
			c.rev_start = 2 * ix->l_pac - total;
			ix->contigs.push_back(c);
		}
		fclose(fp);
		if ((uint64_t)(2 * ix->l_pac) != v.seq_len) return fail(KG_ERR_IO, "%s: .ann l_pac does not match the BWT length", prefix);
	}
	size_t pac_bytes = (size_t)(ix->l_pac / 4 + 1);
	if (pac.size() < pac_bytes) return fail(KG_ERR_IO, "%s.pac is truncated", prefix);
	ix->pac.assign(pac.data(), pac.data() + pac_bytes);

	// upload: Occ/BWT blocks (+ one zero block of padding so a 64-byte fetch of the last,
	// partial block stays inside the allocation), SA samples, 2-bit reference
	size_t occ_bytes = n_words * 4 + 128;
	HIP_TRY(hipMalloc((void **)&ix->d_occ, occ_bytes));
	HIP_TRY(hipMemset(ix->d_occ, 0, occ_bytes));
	HIP_TRY(hipMemcpy(ix->d_occ, bwt.data() + 40, n_words * 4, hipMemcpyHostToDevice));
	const uint64_t sa0 = (uint64_t)-1;                    // sa[0] = -1, the samples follow straight from the file image
	HIP_TRY(hipMalloc((void **)&ix->d_sa, ix->n_sa * 8));
	HIP_TRY(hipMemcpy(ix->d_sa, &sa0, 8, hipMemcpyHostToDevice));
	if (ix->n_sa > 1) HIP_TRY(hipMemcpy(ix->d_sa + 1, sa.data() + 56, (ix->n_sa - 1) * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMalloc((void **)&ix->d_pac, pac_bytes));
	HIP_TRY(hipMemcpy(ix->d_pac, ix->pac.data(), pac_bytes, hipMemcpyHostToDevice));
	{
		std::vector<std::pair<int64_t, int32_t>> ec;
		for (size_t i = 0; i < ix->contigs.size(); ++i) {
			const ContigRec &c = ix->contigs[i];
			ec.emplace_back(c.fwd_start + c.len - 1, (int32_t)i);
			ec.emplace_back(c.rev_start + c.len - 1, (int32_t)i);
		}
		std::sort(ec.begin(), ec.end());
		std::vector<int64_t> ends;
		std::vector<int32_t> echr;
		for (const std::pair<int64_t, int32_t> &e : ec) { ends.push_back(e.first); echr.push_back(e.second); }
		ix->n_ends = (int)ends.size();
		HIP_TRY(hipMalloc((void **)&ix->d_contig_end, 8 * ends.size() + 8));
		HIP_TRY(hipMemcpy(ix->d_contig_end, ends.data(), 8 * ends.size(), hipMemcpyHostToDevice));
		HIP_TRY(hipMalloc((void **)&ix->d_end_chr, 4 * echr.size() + 8));
		HIP_TRY(hipMemcpy(ix->d_end_chr, echr.data(), 4 * echr.size(), hipMemcpyHostToDevice));
		std::vector<int64_t> tab;
		for (const ContigRec &c : ix->contigs) tab.push_back(c.fwd_start);
		for (const ContigRec &c : ix->contigs) tab.push_back(c.rev_start);
		for (const ContigRec &c : ix->contigs) tab.push_back(c.len);
		HIP_TRY(hipMalloc((void **)&ix->d_chr_tab, 8 * tab.size() + 8));
		HIP_TRY(hipMemcpy(ix->d_chr_tab, tab.data(), 8 * tab.size(), hipMemcpyHostToDevice));
		// EvaluateMAPQ (src/Mapping.cpp:172): (int)(30 * (1 - (float)(score - sub_score) / score) * log(score) + 0.4999) for the
		// only arguments that reach it (score - sub_score in 1..5), evaluated here with the host's libm -- float / double mix as written
		// ... and for score < sub_score with score < 8 (larger scores always give more than 60 there)
		std::vector<uint8_t> mq((size_t)(kAlnMaxScore + 1) * 6 + (size_t)(kAlnMaxScore + 1) * 8, 0);
		for (int sc = 1; sc <= kAlnMaxScore; ++sc)
			for (int d = 1; d <= 5 && d < sc; ++d) {
				int sub = sc - d;
				int q = (int)(30 * (1 - (float)(sc - sub) / sc) * log(sc) + 0.4999);
				mq[(size_t)sc * 6 + (size_t)d] = (uint8_t)(q > 60 ? 60 : q < 0 ? 0 : q);
			}
		for (int sc = 1; sc < 8; ++sc)
			for (int nd = 1; nd <= kAlnMaxScore; ++nd) {
				int sub = sc + nd;
				int q = (int)(30 * (1 - (float)(sc - sub) / sc) * log(sc) + 0.4999);
				mq[(size_t)(kAlnMaxScore + 1) * 6 + (size_t)sc * (kAlnMaxScore + 1) + (size_t)nd] = (uint8_t)(q > 60 ? 60 : q < 0 ? 0 : q);
			}
		HIP_TRY(hipMalloc((void **)&ix->d_mapq_tab, mq.size()));
		HIP_TRY(hipMemcpy(ix->d_mapq_tab, mq.data(), mq.size(), hipMemcpyHostToDevice));
	}
	ix->device_bytes = occ_bytes + ix->n_sa * 8 + pac_bytes;
	v.occ = ix->d_occ;
	v.sa = ix->d_sa;
	{
		// device-private rank structure (bit-planes, 1 byte/symbol), built from the uploaded blocks;
		// the source blocks are padded to a whole number of 128-symbol blocks by the zero fill above
		uint64_t n_blocks64 = ((v.seq_len + 127) / 128) * 2;
		size_t plane_bytes = (size_t)n_blocks64 * 64 + 64;
		HIP_TRY(hipMalloc((void **)&ix->d_planes, plane_bytes));
		HIP_TRY(launch_build_planes(ix->d_occ, n_blocks64, ix->d_planes, nullptr));
		HIP_TRY(hipDeviceSynchronize());
		v.planes = ix->d_planes;
		ix->device_bytes += plane_bytes;
	}
	v.planes2 = nullptr;
	v.planes3 = nullptr;
	for (int i = 0; i < 16; ++i) v.t2[i] = 0;
	for (int i = 0; i < 64; ++i) v.t3[i] = 0;
	if (!getenv("KG_NO_PLANES2")) {
		// two-step rank structure: one 128-byte line per 896 rows and pair of bases (2.29 bytes/symbol); an accelerator like the
		// q-mer table -- without it the search takes single steps
		uint64_t n_lines = (v.seq_len + kPlane2Rows - 1) / kPlane2Rows;
		size_t bytes = (size_t)n_lines * 16 * 128;
		uint64_t *t_dev = nullptr;
		if (hipMalloc((void **)&ix->d_planes2, bytes) == hipSuccess) {
			HIP_TRY(hipMalloc((void **)&t_dev, (16 + 64) * 8));
			HIP_TRY(launch_build_planes_k(v, 2, ix->d_planes2, n_lines, t_dev, t_dev + 16, nullptr));
			HIP_TRY(hipMemcpy(v.t2, t_dev, 16 * 8, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(v.t3, t_dev + 16, 64 * 8, hipMemcpyDeviceToHost));
			HIP_TRY(hipFree(t_dev));
			v.planes2 = ix->d_planes2;
			ix->device_bytes += bytes;
		} else (void)hipGetLastError();
	}
	v.fsa32 = nullptr;
	v.fsa64 = nullptr;
	v.fsa40 = nullptr;
	v.dsa32 = nullptr;
	v.dsa64 = nullptr;
	v.dsa_shift = 0;
	v.text = nullptr;
	{
		// the indexed text itself, 2 bits per base, forward + reverse complement: the alignment stage compares reads with it;
		// with the full suffix array resident the search kernel finishes single-suffix searches against it as well
		size_t text_bytes = (size_t)(v.seq_len / 4 + 1) + 16;
		HIP_TRY(hipMalloc((void **)&ix->d_text, text_bytes));
		HIP_TRY(launch_build_text(ix->d_pac, (uint64_t)ix->l_pac, ix->d_text, text_bytes, nullptr));
		HIP_TRY(hipDeviceSynchronize());
		ix->device_bytes += text_bytes;
	}
	v.qtab32 = nullptr;
	v.qtab64 = nullptr;
	v.qmer = 0;

	const bool dense = sa_mode == KG_SA_DENSE4 || sa_mode == KG_SA_DENSE8;
	const bool wide = sa_mode == KG_SA_FULL40_WIDE;        // 5-byte entries with the full q-mer table and the triple planes
	const bool compact = sa_mode == KG_SA_FULL40 || wide;  // 5-byte entries (KG_SA_FULL40 alone: a smaller q-mer table, no triple planes)
	if (sa_mode == KG_SA_FULL || dense || compact) {
		bool narrow = v.seq_len < 0xFFFFFFFFull && !getenv("KG_FORCE_U64");
		const bool packed = compact && !narrow;
		if (packed && v.seq_len >= (1ull << 40)) return fail(KG_ERR_ARG, "kg_index_load: KG_SA_FULL40 holds texts below 2^40 bases");
		size_t fsa_bytes = packed ? (size_t)(v.seq_len + 1) * 5 + 8 : (size_t)(v.seq_len + 1) * (narrow ? 4 : 8);
		HIP_TRY(hipMalloc(&ix->d_fsa, fsa_bytes));
		uint32_t *f32 = narrow ? (uint32_t *)ix->d_fsa : nullptr;
		uint64_t *f64 = narrow || packed ? nullptr : (uint64_t *)ix->d_fsa;
		uint8_t *f40 = packed ? (uint8_t *)ix->d_fsa : nullptr;
		HIP_TRY(launch_expand_sa(v, ix->n_sa, f32, f64, f40, nullptr));
		HIP_TRY(hipDeviceSynchronize());
		v.fsa32 = f32;
		v.fsa64 = f64;
		v.fsa40 = f40;
		if (!dense) ix->device_bytes += fsa_bytes;
		if (!getenv("KG_NO_DIRECT")) v.text = ix->d_text;   // finishing single-suffix searches by comparison against the text
	} else if (sa_mode != KG_SA_SAMPLED) {
		return fail(KG_ERR_ARG, "kg_index_load: unknown sa_mode %d", sa_mode);
	}
	// q-mer interval table: 4^q entries with 4^q ~ text length (12 for E. coli, 16 for hg38), 8 bytes each; built
	// last because entries of a single suffix hold that suffix (needs the full SA)
	if (!getenv("KG_NO_QTAB")) {
		bool narrow = v.seq_len < 0xFFFFFF00ull && !getenv("KG_FORCE_U64");
		int q = kQmerMin;
		while (q < kQmerMax && ((uint64_t)1 << (2 * q + 1)) <= v.seq_len) q++;      // round(log4(2L))
		if (compact && !wide && q > kQmerMin) q--;                                  // a quarter of the table: one more rank step per search
		if (const char *env = getenv("KG_QMER")) { int t = atoi(env); if (t >= kQmerMin && t <= kQmerMax) q = t; }   // tuning knob
		size_t tab_bytes = ((size_t)1 << (2 * q)) * 8;
		// the table is an accelerator, not a requirement: when the device cannot hold 4^q entries (34 GB at q = 16) take a smaller q
		while (hipMalloc(&ix->d_qtab, tab_bytes) != hipSuccess) {
			(void)hipGetLastError();
			ix->d_qtab = nullptr;
			if (q <= kQmerMin) return fail(KG_ERR_NOMEM, "kg_index_load: no device memory for the q-mer table");
			q--;
			tab_bytes >>= 2;
		}
		v.qmer = q;
		HIP_TRY(launch_build_qtab(v, q, narrow ? (uint2 *)ix->d_qtab : nullptr, narrow ? nullptr : (uint64_t *)ix->d_qtab, nullptr));
		HIP_TRY(hipDeviceSynchronize());
		if (narrow) v.qtab32 = (const uint2 *)ix->d_qtab; else v.qtab64 = (const uint64_t *)ix->d_qtab;
		ix->device_bytes += tab_bytes;
	}
	if (dense) {
		// the smaller index: keep every 4th / 8th entry of the expansion (the q-mer table above has taken the single suffixes it
		// needs from the full array), free the rest
		const int shift = sa_mode == KG_SA_DENSE4 ? 2 : 3;
		const bool narrow = v.fsa32 != nullptr;
		const uint64_t n_out = (v.seq_len >> shift) + 1;
		const size_t bytes = (size_t)n_out * (narrow ? 4 : 8);
		HIP_TRY(hipMalloc(&ix->d_dsa, bytes));
		HIP_TRY(launch_sample_sa(v.fsa32, v.fsa64, n_out, shift, narrow ? (uint32_t *)ix->d_dsa : nullptr, narrow ? nullptr : (uint64_t *)ix->d_dsa, nullptr));
		HIP_TRY(hipDeviceSynchronize());
		HIP_TRY(hipFree(ix->d_fsa));
		ix->d_fsa = nullptr;
		v.fsa32 = nullptr; v.fsa64 = nullptr;
		if (narrow) v.dsa32 = (const uint32_t *)ix->d_dsa; else v.dsa64 = (const uint64_t *)ix->d_dsa;
		v.dsa_shift = shift;
		ix->device_bytes += bytes;
	}
	// three-step rank structure (9.14 bytes/symbol: 56.7 GB for hg38), last and only where the device keeps room for the
	// workspaces after it (a 288 GB device does, with the 111 GB of everything else): searches then take three bases per rank pair
	if (v.planes2 && !dense && (!compact || wide) && !getenv("KG_NO_PLANES3")) {
		uint64_t n_lines = (v.seq_len + kPlane2Rows - 1) / kPlane2Rows;
		size_t bytes = (size_t)n_lines * 64 * 128, free_b = 0, total_b = 0;
		const size_t reserve = (size_t)48 << 30;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > bytes + reserve && hipMalloc((void **)&ix->d_planes3, bytes) == hipSuccess) {
			HIP_TRY(launch_build_planes_k(v, 3, ix->d_planes3, n_lines, nullptr, nullptr, nullptr));
			v.planes3 = ix->d_planes3;
			ix->device_bytes += bytes;
		} else (void)hipGetLastError();
	}
	*out = ix.release();
	return KG_OK;
}

void kg_index_destroy(kg_index *ix)
{
	if (!ix) return;
	(void)hipSetDevice(ix->device);
	kgi_frag_release(ix);
	if (ix->d_occ) (void)hipFree(ix->d_occ);
	if (ix->d_planes) (void)hipFree(ix->d_planes);
	if (ix->d_planes2) (void)hipFree(ix->d_planes2);
	if (ix->d_planes3) (void)hipFree(ix->d_planes3);
	if (ix->d_qtab) (void)hipFree(ix->d_qtab);
	for (NwScratch *sc : ix->nw_pool) {
		if (sc->done) { (void)hipEventSynchronize(sc->done); (void)hipEventDestroy(sc->done); }
		if (sc->lists) (void)hipFree(sc->lists);
		if (sc->queue) (void)hipFree(sc->queue);
		if (sc->dir) (void)hipFree(sc->dir);
		if (sc->io) (void)hipFree(sc->io);
		if (sc->io_stream) (void)hipStreamDestroy(sc->io_stream);
		delete sc;
	}
	if (ix->d_sa) (void)hipFree(ix->d_sa);
	if (ix->d_fsa) (void)hipFree(ix->d_fsa);
	if (ix->d_dsa) (void)hipFree(ix->d_dsa);
	if (ix->d_text) (void)hipFree(ix->d_text);
	if (ix->d_pac) (void)hipFree(ix->d_pac);
	if (ix->d_contig_end) (void)hipFree(ix->d_contig_end);
	if (ix->d_end_chr) (void)hipFree(ix->d_end_chr);
	if (ix->d_chr_tab) (void)hipFree(ix->d_chr_tab);
	if (ix->d_mapq_tab) (void)hipFree(ix->d_mapq_tab);
	delete ix;
}

int kg_index_info(const kg_index *ix, kg_index_info_t *info)
{
	if (!ix || !info) return fail(KG_ERR_ARG, "kg_index_info: null argument");
	info->genome_size = ix->l_pac;
	info->seq_len = ix->view.seq_len;
	info->primary = ix->view.primary;
	info->n_contigs = (int32_t)ix->contigs.size();
	// MinSeedLength: smallest k in 13..15 with 2L < 4^k, else 16 (reference src/Mapping.cpp:645)
	int k = 13;
	for (; k < 16; ++k)
		if ((double)(2 * ix->l_pac) < (double)(1ull << (2 * k))) break;
	info->min_seed_len = k;
	info->sa_mode = ix->sa_mode;
	info->device = ix->device;
	info->device_bytes = ix->device_bytes;
	return KG_OK;
}

int kg_index_contig(const kg_index *ix, int i, kg_contig_t *out)
{
	if (!ix || !out || i < 0 || i >= (int)ix->contigs.size()) return fail(KG_ERR_ARG, "kg_index_contig: bad argument");
	const ContigRec &c = ix->contigs[(size_t)i];
	out->name = c.name.c_str();
	out->fwd_start = c.fwd_start;
	out->rev_start = c.rev_start;
	out->len = c.len;
	return KG_OK;
}

int kg_rank_sa_batch(kg_index *ix, const uint64_t *k, int64_t n, uint64_t *occ4, uint64_t *sa_walk, uint64_t *sa_full)
{
	if (!ix || !k || n < 0) return fail(KG_ERR_ARG, "kg_rank_sa_batch: bad argument");
	if (n == 0) return KG_OK;
	HIP_TRY(hipSetDevice(ix->device));
	uint64_t *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 8 * (size_t)n * 7));
	uint64_t *d_k = d, *d_occ = d + n, *d_walk = d + 5 * n, *d_full = d + 6 * n;
	hipError_t e = hipMemcpy(d_k, k, 8 * (size_t)n, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = launch_rank_sa(ix->view, d_k, n, occ4 ? d_occ : nullptr, sa_walk ? d_walk : nullptr, sa_full ? d_full : nullptr, nullptr);
	if (e == hipSuccess) e = hipDeviceSynchronize();
	if (e == hipSuccess && occ4) e = hipMemcpy(occ4, d_occ, 32 * (size_t)n, hipMemcpyDeviceToHost);
	if (e == hipSuccess && sa_walk) e = hipMemcpy(sa_walk, d_walk, 8 * (size_t)n, hipMemcpyDeviceToHost);
	if (e == hipSuccess && sa_full) e = hipMemcpy(sa_full, d_full, 8 * (size_t)n, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(KG_ERR_NO_DEVICE, "kg_rank_sa_batch: %s", hipGetErrorString(e));
	return KG_OK;
}

void *kg_host_alloc(size_t bytes)
{
	void *p = nullptr;
	// (portable: page-locked for every device, whichever one the calling thread happens to have current -- the reader threads
	//  of a -gpu N run call this without ever selecting the index's device)
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return p;
}

void kg_host_free(void *p)
{
	if (p) (void)hipHostFree(p);
}

int kg_workspace_create(kg_index *ix, int64_t max_reads, int64_t max_bases, kg_workspace **out)
{
	if (!ix || !out || max_reads <= 0 || max_bases <= 0) return fail(KG_ERR_ARG, "kg_workspace_create: bad argument");
	*out = nullptr;
	HIP_TRY(hipSetDevice(ix->device));
	std::unique_ptr<kg_workspace, void (*)(kg_workspace *)> ws(new kg_workspace(), kg_workspace_destroy);
	ws->ix = ix;
	ws->max_reads = max_reads;
	ws->max_bases = max_bases;
	// every hit consumes at least 13 read bases (MinSeedLength >= 13), plus one per read of slack
	// ... plus the slack of the per-wave slot pools (one 256-slot chunk per resident wave)
	// (a hit per 13 bases is the most one walk over a read can leave; + a quarter for the walks from the segment starts of a long read that merge into it, search.inc)
	ws->max_hits = max_bases / 13 + max_bases / 52 + max_reads + (int64_t)ix->n_cu * 32 * 256 + 4096;
	HIP_TRY(hipMalloc((void **)&ws->d_hits, sizeof(Hit) * (size_t)ws->max_hits));
	HIP_TRY(hipMalloc((void **)&ws->d_packed, 8 * (size_t)(max_bases / 16 + 3 * max_reads + 64)));
	HIP_TRY(hipMalloc((void **)&ws->d_seeds_per_read, 4 * (size_t)max_reads));
	HIP_TRY(hipMalloc((void **)&ws->d_ctl, 8 * kCtlWords));
	ws->scan_bytes = scan_temp_bytes(max_reads + 1);
	HIP_TRY(hipMalloc(&ws->d_scan_temp, ws->scan_bytes ? ws->scan_bytes : 256));
	if (getenv("KG_SORT_READS")) {
		ws->sort_bytes = sort_temp_bytes(max_reads);
		HIP_TRY(hipMalloc((void **)&ws->d_sort_keys, 16 * (size_t)max_reads + 64));
		HIP_TRY(hipMalloc(&ws->d_sort_temp, ws->sort_bytes ? ws->sort_bytes : 256));
	}
	HIP_TRY(hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking));
	// (zeroed ON THE WORKSPACE'S STREAM: a hipMemset of device memory runs on the null stream, asynchronously, and nothing orders
	//  it against later work on a non-blocking stream -- it can land in the middle of the first batch)
	HIP_TRY(hipMemsetAsync(ws->d_ctl, 0, 8 * kCtlWords, ws->stream));
	HIP_TRY(hipHostMalloc((void **)&ws->h_small, 8 * 16, hipHostMallocDefault));
	*out = ws.release();
	return KG_OK;
}

void kg_workspace_destroy(kg_workspace *ws)
{
	if (!ws) return;
	(void)hipSetDevice(ws->ix->device);
	if (ws->stream) { (void)hipStreamSynchronize(ws->stream); (void)hipStreamDestroy(ws->stream); }
	kgi_long_release(ws);
	for (int i = 0; i < kg_workspace::kRing; ++i) {
		if (ws->ring_cands[i]) (void)hipHostFree(ws->ring_cands[i]);
		if (ws->ring_seeds[i]) (void)hipHostFree(ws->ring_seeds[i]);
		if (ws->ring_records[i]) (void)hipHostFree(ws->ring_records[i]);
	}
	void *ptrs[] = {ws->d_sort_keys, ws->d_sort_temp, ws->d_plans, ws->d_tasks, ws->d_aln_cand, ws->d_aln_read, ws->d_spill, ws->d_jobs, ws->d_job_ops, ws->d_job_len, ws->d_chunk_off, ws->d_chunk_paired, ws->d_chunk_stats, ws->d_aln_ctl, ws->d_used, ws->d_cand_off, ws->d_cseed_off, ws->d_dense_cands, ws->d_dense_seeds, ws->d_cands, ws->d_cand_seeds, ws->d_n_cands, ws->d_taken, ws->d_hits, ws->d_packed, ws->d_seeds_per_read, ws->d_ctl, ws->d_scan_temp, ws->d_enc, ws->d_read_off, ws->d_seed_off, ws->d_seeds, ws->d_vr_n, ws->d_vr_off, ws->d_vr_read, ws->d_vr_pos, ws->d_claim, ws->d_step};
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	if (ws->h_seeds) (void)hipHostFree(ws->h_seeds);
	if (ws->h_small) (void)hipHostFree(ws->h_small);
	if (ws->sync_ev) (void)hipEventDestroy(ws->sync_ev);
	for (hipEvent_t e : ws->ev)
		if (e) (void)hipEventDestroy(e);
	for (int i = 0; i < KT_SLOTS; ++i) {
		for (int j = 0; j < kKtRing; ++j) {
			if (ws->kt.b[i][j]) (void)hipEventDestroy(ws->kt.b[i][j]);
			if (ws->kt.e[i][j]) (void)hipEventDestroy(ws->kt.e[i][j]);
		}
	}
	delete ws;
}

int kg_workspace_counters(kg_workspace *ws, kg_counters_t *out)
{
	if (!ws || !out) return fail(KG_ERR_ARG, "kg_workspace_counters: null argument");
	HIP_TRY(hipSetDevice(ws->ix->device));
	HIP_TRY(hipDeviceSynchronize());
	unsigned long long ctl[kCtlWords];
	HIP_TRY(hipMemcpy(ctl, ws->d_ctl, sizeof(ctl), hipMemcpyDeviceToHost));
	out->searches = ctl[4]; out->lf1 = ctl[5]; out->lf2 = ctl[6]; out->inv = ctl[7];
	out->sa = ctl[8]; out->seeds = ctl[9]; out->bases = ctl[10];
	return KG_OK;
}

int kg_workspace_traffic(kg_workspace *ws, kg_traffic_t *out)
{
	if (!ws || !out) return fail(KG_ERR_ARG, "kg_workspace_traffic: null argument");
	HIP_TRY(hipSetDevice(ws->ix->device));
	HIP_TRY(hipDeviceSynchronize());
	unsigned long long ctl[kCtlWords];
	HIP_TRY(hipMemcpy(ctl, ws->d_ctl, sizeof(ctl), hipMemcpyDeviceToHost));
	out->table_lookups = ctl[17]; out->rank_steps = ctl[18]; out->rank_steps_two_lines = ctl[19];
	out->text_rounds = ctl[20]; out->window_words = ctl[21]; out->rank_steps_two_lines_narrow = ctl[22];
	out->sa_gathers = ctl[8]; out->hits = ctl[1]; out->searches = ctl[4];
	out->sa_entry_bytes = (ws->ix->view.fsa32 ? 4 : ws->ix->view.fsa40 ? 5 : 8);
	out->double_steps = ctl[23]; out->double_steps_two_lines = ctl[24]; out->double_step_bytes = ctl[25]; out->triple_steps = ctl[26];
	return KG_OK;
}

int kg_workspace_set_single_steps(kg_workspace *ws, int enabled)
{
	if (!ws) return fail(KG_ERR_ARG, "kg_workspace_set_single_steps: null workspace");
	ws->single_steps = enabled != 0;
	return KG_OK;
}

int kg_index_selfcheck(kg_index *ix, int64_t samples, uint64_t seed, uint64_t *disagreements)
{
	if (!ix || !disagreements || samples < 0) return fail(KG_ERR_ARG, "kg_index_selfcheck: bad argument");
	*disagreements = 0;
	if (!ix->view.planes2) return fail(KG_ERR_ARG, "kg_index_selfcheck: the index holds no two-step rank structure");
	HIP_TRY(hipSetDevice(ix->device));
	unsigned long long *bad = nullptr;
	HIP_TRY(hipMalloc((void **)&bad, 8));
	HIP_TRY(hipMemset(bad, 0, 8));
	HIP_TRY(hipDeviceSynchronize());
	hipError_t e = launch_planes2_check(ix->view, (uint64_t)samples, seed, bad, nullptr);
	unsigned long long h = 0;
	hipError_t e2 = hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
	(void)hipFree(bad);
	if (e != hipSuccess || e2 != hipSuccess) return fail(KG_ERR_NO_DEVICE, "kg_index_selfcheck: %s", hipGetErrorString(e != hipSuccess ? e : e2));
	*disagreements = h;
	return KG_OK;
}

int kg_workspace_set_profiling(kg_workspace *ws, int enabled)
{
	if (!ws) return fail(KG_ERR_ARG, "kg_workspace_set_profiling: null workspace");
	HIP_TRY(hipSetDevice(ws->ix->device));
	if (enabled && !ws->ev[0])
		for (int i = 0; i < 5; ++i) HIP_TRY(hipEventCreate(&ws->ev[i]));
	if (enabled && !ws->kt.b[0][0])
		for (int i = 0; i < KT_SLOTS; ++i)
			for (int j = 0; j < kKtRing; ++j) { HIP_TRY(hipEventCreate(&ws->kt.b[i][j])); HIP_TRY(hipEventCreate(&ws->kt.e[i][j])); }
	ws->profiling = enabled != 0;
	return KG_OK;
}

int kg_workspace_kernel_ms(kg_workspace *ws, float ms[4])
{
	if (!ws || !ms) return fail(KG_ERR_ARG, "kg_workspace_kernel_ms: null argument");
	if (!ws->profiling || !ws->ev[0]) return fail(KG_ERR_ARG, "kg_workspace_kernel_ms: profiling is not enabled");
	HIP_TRY(hipSetDevice(ws->ix->device));
	HIP_TRY(hipEventSynchronize(ws->ev[4]));
	for (int i = 0; i < 4; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], ws->ev[i], ws->ev[i + 1]));
	return KG_OK;
}

int64_t kg_workspace_overflow(kg_workspace *ws)
{
	if (!ws) return -1;
	if (hipSetDevice(ws->ix->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
	unsigned long long v = 0;
	if (hipMemcpy(&v, ws->d_ctl + 11, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return (int64_t)v;
}

int64_t kg_workspace_segment_fallbacks(kg_workspace *ws)
{
	return ws ? ws->segment_fallbacks : -1;
}

static int check_seed_args(kg_workspace *ws, int mode, int min_seed_len, int occ_thr, int64_t n_reads, int64_t n_bases)
{
	if (!ws) return fail(KG_ERR_ARG, "kg_seed_batch: null workspace");
	if ((mode & ~KG_INPUT_ASCII) != KG_MODE_FAST && (mode & ~KG_INPUT_ASCII) != KG_MODE_SENSITIVE) return fail(KG_ERR_ARG, "kg_seed_batch: unknown mode %d", mode);
	if (min_seed_len < 13 || min_seed_len > 16) return fail(KG_ERR_ARG, "kg_seed_batch: min_seed_len %d outside 13..16", min_seed_len);
	if (occ_thr < 1 || occ_thr > 1000000) return fail(KG_ERR_ARG, "kg_seed_batch: occ_thr %d out of range", occ_thr);
	if (n_reads < 0 || n_reads > ws->max_reads) return fail(KG_ERR_CAPACITY, "kg_seed_batch: %lld reads exceed the workspace (%lld)", (long long)n_reads, (long long)ws->max_reads);
	if (n_bases < 0 || n_bases > ws->max_bases) return fail(KG_ERR_CAPACITY, "kg_seed_batch: %lld bases exceed the workspace (%lld)", (long long)n_bases, (long long)ws->max_bases);
	return KG_OK;
}

int kg_seed_batch_device(kg_workspace *ws, int mode, int min_seed_len, int occ_thr, const uint8_t *d_enc_bases,
                         const int64_t *d_read_offsets, int64_t n_reads, int64_t n_bases, int64_t *d_seed_offsets,
                         kg_seed *d_seeds, int64_t seed_capacity, void *stream)
{
	int rc = check_seed_args(ws, mode, min_seed_len, occ_thr, n_reads, n_bases);
	if (rc != KG_OK) return rc;
	if (!d_read_offsets || !d_seed_offsets || (n_bases > 0 && !d_enc_bases) || (seed_capacity > 0 && !d_seeds) || seed_capacity < 0)
		return fail(KG_ERR_ARG, "kg_seed_batch_device: null buffer");
	HIP_TRY(hipSetDevice(ws->ix->device));
	hipStream_t st = (hipStream_t)stream;
	if (n_reads == 0) {
		HIP_TRY(hipMemsetAsync(d_seed_offsets, 0, 8, st));
		return KG_OK;
	}
	SeedArgs a;
	a.ix = ws->ix->view;
	a.enc = d_enc_bases;
	a.read_off = d_read_offsets;
	a.n_reads = n_reads;
	a.n_bases = n_bases;
	a.mode = mode & ~KG_INPUT_ASCII;
	a.ascii = (mode & KG_INPUT_ASCII) ? 1 : 0;
	a.min_seed_len = min_seed_len;
	a.occ_thr = occ_thr;
	a.packed = ws->d_packed;
	a.single_steps = ws->single_steps ? 1 : 0;
	a.read_order = nullptr; a.sort_keys = ws->d_sort_keys; a.sort_temp = ws->d_sort_temp; a.sort_temp_bytes = ws->sort_bytes;
	a.hits = ws->d_hits;
	a.max_hits = ws->max_hits;
	a.seeds_per_read = ws->d_seeds_per_read;
	a.read_queue = ws->d_ctl + 0;
	a.hit_count = ws->d_ctl + 1;
	a.locate_queue = ws->d_ctl + 2;
	a.counters = ws->d_ctl + 4;
	a.traffic = ws->d_ctl + 17;
	a.seed_off = d_seed_offsets;
	a.seeds = d_seeds;
	a.seed_capacity = seed_capacity;
	a.read_len = nullptr; a.n_seg = 0; a.seg_stride = 0;
	for (int64_t &x : a.seg_prefix) x = 0;
	// long reads in SensitiveMode: walks from every segment start instead of one lane per read (seed_kernels.hpp, SeedArgs::vr_read)
	// (KG_NO_SEGMENTS / KG_SEG_LEN are read per call: A/B runs and the tests switch them inside one process)
	const bool no_segments = getenv("KG_NO_SEGMENTS") != nullptr;
	const int seg_len = getenv("KG_SEG_LEN") ? std::max(128, atoi(getenv("KG_SEG_LEN"))) : 512;
	ws->last_segmented = false;
	if (a.mode == KG_MODE_SENSITIVE && !no_segments && ws->segments_off_batches == 0 && ws->group_segments == 0 && n_bases >= 4 * (int64_t)seg_len * n_reads) {
		ws->last_segmented = true;
		if (!ws->d_vr_n) {
			ws->vr_capacity = ws->max_bases / 128 + ws->max_reads + 64;
			HIP_TRY(hipMalloc((void **)&ws->d_vr_n, 4 * (size_t)(ws->max_reads + 2)));
			HIP_TRY(hipMalloc((void **)&ws->d_vr_off, 8 * (size_t)(ws->max_reads + 2)));
			HIP_TRY(hipMalloc((void **)&ws->d_vr_read, 4 * (size_t)ws->vr_capacity));
			HIP_TRY(hipMalloc((void **)&ws->d_vr_pos, 4 * (size_t)ws->vr_capacity));
			HIP_TRY(hipMalloc((void **)&ws->d_claim, 4 * (size_t)((ws->max_bases >> 5) + ws->max_reads + 64)));
			HIP_TRY(hipMalloc((void **)&ws->d_step, (size_t)ws->max_bases + 64));
		}
		a.seg_len = seg_len;
		a.vr_read = ws->d_vr_read; a.vr_pos = ws->d_vr_pos;
		a.vr_total = reinterpret_cast<const unsigned long long *>(ws->d_vr_off + n_reads);
		a.claim = ws->d_claim; a.step = ws->d_step;
		HIP_TRY(launch_seed_segments(a, ws->d_vr_n, ws->d_vr_off, ws->d_vr_read, ws->d_vr_pos, ws->d_scan_temp, ws->scan_bytes, ws->ix->n_cu, st));
	}
	if (ws->group_segments > 0) {          // (kgi_seed_group: one launch over several lanes' batches)
		a.read_len = ws->group_read_len;
		a.n_seg = ws->group_segments;
		a.seg_stride = ws->group_stride;
		for (int i = 0; i <= a.n_seg; ++i) a.seg_prefix[i] = ws->group_prefix[i];
	}
	HIP_TRY(launch_seed_batch(a, ws->d_scan_temp, ws->scan_bytes, ws->ix->n_cu, st, ws->profiling ? ws->ev : nullptr));
	return KG_OK;
}

// waits for everything enqueued on the workspace's stream.  The wait sleeps on an interrupt (hipEventBlockingSync) instead of
// spinning: several device threads wait at once in a pipelined run and the host's cores are the scarce resource there
hipError_t kgi_sync(kg_workspace *ws)
{
	static const bool spin = getenv("KG_SPIN_SYNC") != nullptr;
	if (spin) return hipStreamSynchronize(ws->stream);
	if (!ws->sync_ev) {
		hipError_t e = hipEventCreateWithFlags(&ws->sync_ev, hipEventBlockingSync | hipEventDisableTiming);
		if (e != hipSuccess) return e;
	}
	hipError_t e = hipEventRecord(ws->sync_ev, ws->stream);
	if (e != hipSuccess) return e;
	return hipEventSynchronize(ws->sync_ev);
}

// seeding of the batch resident in ws->d_enc / ws->d_read_off: the seeds stay in ws->d_seeds / ws->d_seed_off.  The output
// capacity grows on demand: run, and if the batch overflowed, re-run once with the exact size.
int kgi_seed_resident(kg_workspace *ws, int mode, int min_seed_len, int occ_thr, int64_t n_reads, int64_t n_bases, int64_t *total_out)
{
	int rc = check_seed_args(ws, mode, min_seed_len, occ_thr, n_reads, n_bases);
	if (rc != KG_OK) return rc;
	// (sized for the workspace's largest batch at once: a reallocation synchronises the whole device, and batches grow while a run ramps up)
	int64_t want = std::max<int64_t>(ws->seed_capacity, 8 * std::max(n_reads, ws->max_reads) + 1024);
	int64_t total = 0;
	bool retried_plain = false;
	for (int attempt = 0; attempt < 2; ++attempt) {
		if (want > ws->seed_capacity) {
			if (ws->d_seeds) HIP_TRY(hipFree(ws->d_seeds));
			ws->d_seeds = nullptr;
			HIP_TRY(hipMalloc((void **)&ws->d_seeds, sizeof(kg_seed) * (size_t)want));
			ws->seed_capacity = want;
		}
		rc = kg_seed_batch_device(ws, mode, min_seed_len, occ_thr, ws->d_enc, ws->d_read_off, n_reads, n_bases,
		                          ws->d_seed_off, ws->d_seeds, ws->seed_capacity, ws->stream);
		if (rc != KG_OK) return rc;
		unsigned long long *h = ws->h_small;           // (page-locked: the two copies below are truly asynchronous)
		HIP_TRY(hipMemcpyAsync(&h[0], ws->d_seed_off + n_reads, 8, hipMemcpyDeviceToHost, ws->stream));
		HIP_TRY(hipMemcpyAsync(&h[1], ws->d_ctl + 11, 8, hipMemcpyDeviceToHost, ws->stream));
		HIP_TRY(kgi_sync(ws));
		if (h[1] == (~0ull >> 1)) {
			// The hit list is sized for ONE walk per read (a hit per 13 bases at most).  The walks from the segment starts of a long read add to
			// that only until they merge, and they merge where an error resynchronises them: on exact or nearly exact long reads (HiFi, contigs)
			// they never do -- every search returns 30 bases, 512 k mod 30 differs for every k -- and all of a read's walks run to its end,
			// ~0.25 hits per base.  Such a batch is seeded again with one lane per read, the bound the list was sized for, and so are the
			// workspace's next batches (the data set is what it is); segments are tried again after those.
			if (ws->last_segmented && !retried_plain) {
				retried_plain = true;
				ws->segments_off_batches = 17;       // (this re-run and the next 16 batches)
				ws->segment_fallbacks++;
				--attempt;
				continue;
			}
			return fail(KG_ERR_CAPACITY, "kg_seed_batch: the hit list of the workspace overflowed");
		}
		total = (int64_t)h[0];
		if (total <= ws->seed_capacity) break;
		want = total;
		if (attempt == 1) return fail(KG_ERR_CAPACITY, "kg_seed_batch: seed buffer overflow persisted");
	}
	if (ws->segments_off_batches > 0) ws->segments_off_batches--;
	ws->last_reads = n_reads;
	ws->last_seeds = total;
	ws->last_cands = -1;
	ws->last_ascii = (mode & KG_INPUT_ASCII) != 0;
	*total_out = total;
	return KG_OK;
}

// ONE seeding launch over the parsed batches of several stream lanes (abi_stream.hip): `ws` is the group's workspace, sized for
// n_seg * stride read slots; ws->d_enc holds segment s's characters in its own part (the lanes materialise straight into it),
// ws->d_read_off / group_read_len every slot's offset and length (stream_kernels.hip: group_publish_kernel), counts[s] the reads
// segment s holds this round (0: the lane is absent).  The seeds of the whole group stay in ws->d_seeds / ws->d_seed_off;
// seed_base[s] (s = 0 .. n_seg) = first seed of segment s, seed_base[n_seg] the total -- the lanes cut their slices out of that.
int kgi_seed_group(kg_workspace *ws, int mode, int min_seed_len, int occ_thr, int n_seg, int64_t stride, const int64_t *counts, int64_t *seed_base)
{
	if (n_seg < 1 || n_seg > kMaxSeedSegments || stride < 1 || (int64_t)n_seg * stride > ws->max_reads) return fail(KG_ERR_ARG, "kgi_seed_group: %d segments of %lld slots do not fit the group's workspace", n_seg, (long long)stride);
	ws->group_segments = n_seg;
	ws->group_stride = stride;
	ws->group_prefix[0] = 0;
	for (int i = 0; i < n_seg; ++i) {
		if (counts[i] < 0 || counts[i] > stride) return fail(KG_ERR_ARG, "kgi_seed_group: segment %d holds %lld reads, its slots are %lld", i, (long long)counts[i], (long long)stride);
		ws->group_prefix[i + 1] = ws->group_prefix[i] + counts[i];
	}
	const int64_t n_slots = (int64_t)n_seg * stride;
	int64_t total = 0;
	int rc = kgi_seed_resident(ws, mode, min_seed_len, occ_thr, n_slots, ws->max_bases, &total);
	if (rc != KG_OK) return rc;
	// the segments' first seeds: seed_off at every segment's first slot (page-locked words 2 .. 2 + n_seg)
	unsigned long long *h = ws->h_small;
	for (int i = 0; i < n_seg; ++i) HIP_TRY(hipMemcpyAsync(&h[2 + i], ws->d_seed_off + (int64_t)i * stride, 8, hipMemcpyDeviceToHost, ws->stream));
	HIP_TRY(kgi_sync(ws));
	for (int i = 0; i < n_seg; ++i) seed_base[i] = (int64_t)h[2 + i];
	seed_base[n_seg] = total;
	return KG_OK;
}

int kg_seed_batch(kg_workspace *ws, int mode, int min_seed_len, int occ_thr, const uint8_t *enc_bases,
                  const int64_t *read_offsets, int64_t n_reads, int64_t *seed_offsets, const kg_seed **seeds)
{
	if (!ws || !read_offsets || !seed_offsets) return fail(KG_ERR_ARG, "kg_seed_batch: null argument");
	if (seeds) *seeds = nullptr;
	int64_t n_bases = n_reads > 0 ? read_offsets[n_reads] - read_offsets[0] : 0;
	if (n_reads > 0 && read_offsets[0] != 0) return fail(KG_ERR_ARG, "kg_seed_batch: read_offsets[0] must be 0");
	int rc = check_seed_args(ws, mode, min_seed_len, occ_thr, n_reads, n_bases);
	if (rc != KG_OK) return rc;
	if (n_reads == 0) { seed_offsets[0] = 0; return KG_OK; }
	HIP_TRY(hipSetDevice(ws->ix->device));
	if (!ws->d_enc) {
		HIP_TRY(hipMalloc((void **)&ws->d_enc, (size_t)ws->max_bases + 64));
		HIP_TRY(hipMalloc((void **)&ws->d_read_off, 8 * (size_t)(ws->max_reads + 1)));
		HIP_TRY(hipMalloc((void **)&ws->d_seed_off, 8 * (size_t)(ws->max_reads + 1)));
	}
	HIP_TRY(hipMemcpyAsync(ws->d_enc, enc_bases, (size_t)n_bases, hipMemcpyHostToDevice, ws->stream));
	HIP_TRY(hipMemcpyAsync(ws->d_read_off, read_offsets, 8 * (size_t)(n_reads + 1), hipMemcpyHostToDevice, ws->stream));
	int64_t total = 0;
	rc = kgi_seed_resident(ws, mode, min_seed_len, occ_thr, n_reads, n_bases, &total);
	if (rc != KG_OK) return rc;
	HIP_TRY(hipMemcpyAsync(seed_offsets, ws->d_seed_off, 8 * (size_t)(n_reads + 1), hipMemcpyDeviceToHost, ws->stream));
	HIP_TRY(kgi_sync(ws));
	if (!seeds) return KG_OK;             // the caller only wants the candidates: the seeds stay on the device
	if (total > ws->h_seed_capacity) {
		if (ws->h_seeds) HIP_TRY(hipHostFree(ws->h_seeds));
		ws->h_seeds = nullptr;
		int64_t cap = std::max<int64_t>(total, ws->seed_capacity);
		HIP_TRY(hipHostMalloc((void **)&ws->h_seeds, sizeof(kg_seed) * (size_t)cap, hipHostMallocDefault));
		ws->h_seed_capacity = cap;
	}
	if (total > 0) {
		HIP_TRY(hipMemcpyAsync(ws->h_seeds, ws->d_seeds, sizeof(kg_seed) * (size_t)total, hipMemcpyDeviceToHost, ws->stream));
		HIP_TRY(kgi_sync(ws));
	}
	*seeds = ws->h_seeds;
	ws->last_reads = n_reads;
	ws->last_seeds = total;
	return KG_OK;
}

// Replaces GenerateAlignmentCandidateForIlluminaSeq / ForPacBioSeq (reference src/AlignmentCandidates.cpp:82-130,
// 171-224) for every read of the batch the last kg_seed_batch call left on the device.
// chaining of the batch the last seeding call left on the device; totals[0] candidates, totals[1] candidate seeds (dense arrays
// ws->d_dense_cands / d_dense_seeds, per-read ranges ws->d_cand_off)
int kgi_chain_resident(kg_workspace *ws, int pacbio, int max_gaps, int64_t totals[2])
{
	int64_t n = ws->last_reads, m = ws->last_seeds;
	if (m + 1 > ws->cand_capacity) {
		for (void *p : {(void *)ws->d_cands, (void *)ws->d_cand_seeds, (void *)ws->d_taken, (void *)ws->d_dense_cands, (void *)ws->d_dense_seeds})
			if (p) HIP_TRY(hipFree(p));
		ws->d_cands = nullptr; ws->d_cand_seeds = nullptr; ws->d_taken = nullptr; ws->d_dense_cands = nullptr; ws->d_dense_seeds = nullptr;
		int64_t cap = std::max<int64_t>(m + m / 4, 8 * ws->max_reads) + 1024;
		HIP_TRY(hipMalloc((void **)&ws->d_cands, sizeof(kg_candidate) * (size_t)cap));
		HIP_TRY(hipMalloc((void **)&ws->d_cand_seeds, sizeof(kg_seed) * (size_t)cap));
		HIP_TRY(hipMalloc((void **)&ws->d_taken, (size_t)cap));
		HIP_TRY(hipMalloc((void **)&ws->d_dense_cands, sizeof(kg_candidate) * (size_t)cap));
		HIP_TRY(hipMalloc((void **)&ws->d_dense_seeds, sizeof(kg_seed) * (size_t)cap));
		ws->cand_capacity = cap;
	}
	if (n + 1 > ws->ncand_capacity) {
		for (void *p : {(void *)ws->d_n_cands, (void *)ws->d_used, (void *)ws->d_cand_off, (void *)ws->d_cseed_off})
			if (p) HIP_TRY(hipFree(p));
		ws->d_n_cands = nullptr; ws->d_used = nullptr; ws->d_cand_off = nullptr; ws->d_cseed_off = nullptr;
		int64_t cap = std::max(n, ws->max_reads) + 1024;
		HIP_TRY(hipMalloc((void **)&ws->d_n_cands, 4 * (size_t)cap));
		HIP_TRY(hipMalloc((void **)&ws->d_used, 4 * (size_t)cap));
		HIP_TRY(hipMalloc((void **)&ws->d_cand_off, 8 * (size_t)cap));
		HIP_TRY(hipMalloc((void **)&ws->d_cseed_off, 8 * (size_t)cap));
		ws->ncand_capacity = cap;
	}
	ChainArgs a;
	a.read_off = ws->d_read_off; a.n_reads = n; a.seed_off = ws->d_seed_off; a.seeds = ws->d_seeds;
	a.contig_end = ws->ix->d_contig_end; a.n_ends = ws->ix->n_ends;
	a.pacbio = pacbio ? 1 : 0; a.max_gaps = max_gaps;
	a.n_cands = ws->d_n_cands; a.used = ws->d_used; a.cands = ws->d_cands; a.cand_seeds = ws->d_cand_seeds; a.taken = ws->d_taken;
	a.cand_off = ws->d_cand_off; a.cseed_off = ws->d_cseed_off; a.dense_cands = ws->d_dense_cands; a.dense_seeds = ws->d_dense_seeds;
	HIP_TRY(launch_chain_batch(a, ws->d_scan_temp, ws->scan_bytes, ws->ix->n_cu, ws->stream));
	unsigned long long *h = ws->h_small;
	HIP_TRY(hipMemcpyAsync(&h[0], ws->d_cand_off + n, 8, hipMemcpyDeviceToHost, ws->stream));
	HIP_TRY(hipMemcpyAsync(&h[1], ws->d_cseed_off + n, 8, hipMemcpyDeviceToHost, ws->stream));
	HIP_TRY(kgi_sync(ws));
	totals[0] = (int64_t)h[0]; totals[1] = (int64_t)h[1];
	ws->last_cands = totals[0];
	ws->last_cand_seeds = totals[1];
	ws->last_pacbio = pacbio != 0;
	return KG_OK;
}

int kg_candidates_batch(kg_workspace *ws, int pacbio, int max_gaps, int64_t n_reads, int64_t n_seeds, int32_t *n_cands,
                         const kg_candidate **cands, int64_t *n_cands_total, const kg_seed **cand_seeds, int64_t *n_cand_seeds_total)
{
	if (!ws || !n_cands || !cands || !n_cands_total || !cand_seeds || !n_cand_seeds_total) return fail(KG_ERR_ARG, "kg_candidates_batch: null argument");
	*cands = nullptr; *cand_seeds = nullptr; *n_cands_total = 0; *n_cand_seeds_total = 0;
	if (ws->last_reads <= 0) return fail(KG_ERR_ARG, "kg_candidates_batch: no seeded batch on this workspace (call kg_seed_batch first)");
	if (max_gaps < 0) return fail(KG_ERR_ARG, "kg_candidates_batch: negative max_gaps");
	if (n_reads != ws->last_reads || n_seeds != ws->last_seeds)
		return fail(KG_ERR_ARG, "kg_candidates_batch: batch shape (%lld reads, %lld seeds) is not the one kg_seed_batch left on this workspace (%lld, %lld)",
		            (long long)n_reads, (long long)n_seeds, (long long)ws->last_reads, (long long)ws->last_seeds);
	HIP_TRY(hipSetDevice(ws->ix->device));
	const int64_t n = ws->last_reads;
	int64_t totals[2] = {0, 0};
	int rc = kgi_chain_resident(ws, pacbio, max_gaps, totals);
	if (rc != KG_OK) return rc;
	int64_t need = std::max(totals[0], totals[1]);
	const int slot = ws->ring_at;
	ws->ring_at = (ws->ring_at + 1) % kg_workspace::kRing;
	if (need > ws->ring_cand_capacity[slot]) {
		if (ws->ring_cands[slot]) HIP_TRY(hipHostFree(ws->ring_cands[slot]));
		if (ws->ring_seeds[slot]) HIP_TRY(hipHostFree(ws->ring_seeds[slot]));
		ws->ring_cands[slot] = nullptr; ws->ring_seeds[slot] = nullptr;
		// (sized for the workspace's largest batch at once: page-locked allocations are slow, and batches grow while a run ramps up)
		int64_t cap = std::max<int64_t>(need + need / 4 + 1024, std::min<int64_t>(3 * ws->max_reads, 2500000));
		HIP_TRY(hipHostMalloc((void **)&ws->ring_cands[slot], sizeof(kg_candidate) * (size_t)cap, hipHostMallocDefault));
		HIP_TRY(hipHostMalloc((void **)&ws->ring_seeds[slot], sizeof(kg_seed) * (size_t)cap, hipHostMallocDefault));
		ws->ring_cand_capacity[slot] = cap;
	}
	ws->h_cands = ws->ring_cands[slot]; ws->h_cand_seeds = ws->ring_seeds[slot];
	HIP_TRY(hipMemcpyAsync(n_cands, ws->d_n_cands, 4 * (size_t)n, hipMemcpyDeviceToHost, ws->stream));
	if (totals[0] > 0) HIP_TRY(hipMemcpyAsync(ws->h_cands, ws->d_dense_cands, sizeof(kg_candidate) * (size_t)totals[0], hipMemcpyDeviceToHost, ws->stream));
	if (totals[1] > 0) HIP_TRY(hipMemcpyAsync(ws->h_cand_seeds, ws->d_dense_seeds, sizeof(kg_seed) * (size_t)totals[1], hipMemcpyDeviceToHost, ws->stream));
	HIP_TRY(kgi_sync(ws));
	*cands = ws->h_cands; *cand_seeds = ws->h_cand_seeds;
	*n_cands_total = totals[0]; *n_cand_seeds_total = totals[1];
	return KG_OK;
}

// ---- NW -------------------------------------------------------------------------------------------

static int nw_acquire(kg_index *ix, size_t list_words, size_t dir_words, hipStream_t st, NwScratch **out)
{
	std::lock_guard<std::mutex> lock(ix->nw_mu);
	NwScratch *pick = nullptr;
	for (NwScratch *s : ix->nw_pool) {
		// free again for everybody: its kernels are through
		if (s->busy && !s->pending && hipEventQuery(s->done) == hipSuccess) s->busy = false;
		// ... or usable by THIS caller alone: its kernels were enqueued on this very stream, whose order keeps the next ones behind them
		// (back-to-back calls on one stream used to take a fresh 100 MB scratch each: 2.3 ms of hipMalloc per call of 0.5 ms of kernels).
		// It stays busy for everybody else -- rounds 2 and 3 cleared the flag here, and when this caller then picked ANOTHER scratch, a
		// lane on a different stream could take this one while its kernels still ran (seen as a read reported with another read's
		// alignment once eight lanes were in flight; tools/stress_groups.py).
		const bool usable = !s->busy || (!s->pending && s->last_stream == st);
		if (usable && (!pick || (s->list_words >= list_words && s->dir_words >= dir_words && !(pick->list_words >= list_words && pick->dir_words >= dir_words)))) pick = s;
	}
	if (!pick) {
		pick = new NwScratch();
		HIP_TRY(hipEventCreateWithFlags(&pick->done, hipEventDisableTiming));
		HIP_TRY(hipMalloc((void **)&pick->queue, 8 * (size_t)kNwQueueWords));
		ix->nw_pool.push_back(pick);
	}
	if (pick->list_words < list_words) {
		if (pick->lists) HIP_TRY(hipFree(pick->lists));
		pick->lists = nullptr;
		size_t want = list_words + list_words / 2 + 1024;
		HIP_TRY(hipMalloc((void **)&pick->lists, want * 4));
		pick->list_words = want;
	}
	if (pick->dir_words < dir_words) {
		if (pick->dir) HIP_TRY(hipFree(pick->dir));
		pick->dir = nullptr;
		HIP_TRY(hipMalloc((void **)&pick->dir, dir_words * 4));
		pick->dir_words = dir_words;
	}
	pick->busy = true;
	pick->pending = true;
	pick->last_stream = st;
	*out = pick;
	return KG_OK;
}

// the kernels that use the scratch are enqueued: from here on `done` speaks for this use
static hipError_t nw_submitted(kg_index *ix, NwScratch *sc, hipStream_t st)
{
	hipError_t e = hipEventRecord(sc->done, st);
	std::lock_guard<std::mutex> lock(ix->nw_mu);
	sc->pending = false;
	return e;
}

// shared launcher: scratch = 3n list words + 4 queue words (+ direction slabs when a pair is longer than 32)
static int nw_run(kg_index *ix, const char *d_frag1, const int64_t *d_off1, const char *d_frag2, const int64_t *d_off2, int64_t n,
                  int64_t max_len, uint8_t *d_ops, int32_t *d_aln_len, hipStream_t st)
{
	NwArgs a;
	a.f1 = d_frag1; a.off1 = d_off1; a.f2 = d_frag2; a.off2 = d_off2; a.n = n;
	a.ops = d_ops; a.aln_len = d_aln_len;
	return kgi_nw_launch(ix, a, max_len, st);
}

// the NW kernels for the pairs `a` names (offset arrays, or job descriptors written on the device): scratch for the longest pair
// from the index's pool, launch, hand the scratch back once the kernels are through
int kgi_nw_launch(kg_index *ix, NwArgs &a, int64_t max_len, hipStream_t st)
{
	const int64_t n = a.n;
	a.dir_scratch = nullptr;
	a.dir_words_per_wave = 0;
	a.big_waves = 0;
	a.big_lds_bytes = 0;
	a.gb_offset_words = 0;
	size_t dir_words = 0;
	if (max_len > 32) {
		a.big_lds_bytes = nw_big_lds_bytes((int)max_len);
		int per_cu = std::max(1, std::min(16, (160 * 1024) / std::max(a.big_lds_bytes, 1024)));
		int64_t waves = std::min<int64_t>((int64_t)ix->n_cu * per_cu, n);
		a.dir_words_per_wave = nw_dir_words((int)max_len);
		if (max_len > kNwMaxLen) {
			// beyond what the LDS holds: boundary column + codes behind the direction words of the wave's slab
			a.gb_offset_words = a.dir_words_per_wave;
			a.dir_words_per_wave += (((nw_big_lds_bytes((int)max_len) + 3) / 4 + 16) + 3) & ~3ll;      // (slabs stay 16-byte aligned)
			a.big_lds_bytes = 0;
			waves = std::min<int64_t>(waves, 64);        // such fragments are rare and each slab is large
			if (a.dir_words_per_wave * 4 > (64ll << 30)) return fail(KG_ERR_CAPACITY, "kg_nw_batch: a fragment of %lld bases needs more than 64 GiB of traceback words", (long long)max_len);
		}
		while (waves > 1 && waves * a.dir_words_per_wave * 4 > (8ll << 30)) waves /= 2;   // slab pool <= 8 GiB (one slab may exceed it)
		a.big_waves = (int)waves;
		dir_words = (size_t)(waves * a.dir_words_per_wave);
	}
	// Two tiers (long-read batches): one pair of a few thousand bases -- a head or tail fragment, a fragment without a common 8-mer -- sizes the LDS
	// block and the slab of EVERY wave of the launch, and the hundreds of thousands of 33 .. 256-base pairs beside it then run at 6 waves per CU:
	// pairs up to 256 bases get a launch of their own with 32 (417 ms of nw_big_kernel per 400 k x 7 kb reads, profiles/r05f_pacbio_kernel_stats.csv).
	static const bool no_tiers = getenv("KG_NW_NO_TIERS") != nullptr;
	size_t t1_words = 0;
	a.tier_len = 0;
	if (!no_tiers && max_len > 512 && max_len <= kNwMaxLen && a.desc != nullptr) {
		a.tier_len = 256;
		a.t1_lds_bytes = nw_big_lds_bytes(a.tier_len);
		const int per_cu = std::max(1, std::min(32, (160 * 1024) / std::max(a.t1_lds_bytes, 1024)));
		a.t1_waves = (int)std::min<int64_t>((int64_t)ix->n_cu * per_cu, n);
		a.t1_dir_words_per_wave = nw_dir_words(a.tier_len);
		t1_words = (size_t)a.t1_waves * (size_t)a.t1_dir_words_per_wave;
	}
	NwScratch *sc = nullptr;
	int rc = nw_acquire(ix, 4 * (size_t)n, dir_words + t1_words, st, &sc);
	if (rc != KG_OK) return rc;
	a.big_list = sc->lists;
	a.queue = sc->queue;
	if (dir_words) a.dir_scratch = sc->dir;
	if (t1_words) a.t1_dir_scratch = sc->dir + dir_words;
	hipError_t e = launch_nw_batch(a, ix->n_cu, st);
	hipError_t e2 = nw_submitted(ix, sc, st);
	if (e != hipSuccess || e2 != hipSuccess) return fail(KG_ERR_NO_DEVICE, "kg_nw_batch: %s", hipGetErrorString(e != hipSuccess ? e : e2));
	return KG_OK;
}

// The alignment stage (align_kernels.hip) for the chained batch resident on the workspace: buffers grown on demand, every kernel
// enqueued on the workspace's stream, nothing copied back.  `a` receives the argument block (a.records: n reads + the extra
// records of -m, as many as `host_record_capacity` -- the caller's destination array -- and the device array hold).
int kgi_align_resident(kg_workspace *ws, const int64_t *chunk_off, const uint8_t *chunk_paired, int n_chunks, int est_distance, int max_insert,
                       int max_gaps, int multi_hit, int unset_flag, int64_t host_record_capacity, AlnArgs &a)
{
	kg_index *ix = ws->ix;
	const int64_t n = ws->last_reads, nc = ws->last_cands;
	if (!ix->d_text) return fail(KG_ERR_ARG, "kg_align_batch: the index holds no text");
	if (!ws->last_ascii) return fail(KG_ERR_ARG, "kg_align_batch: the batch must have been seeded from read characters (KG_INPUT_ASCII): mismatch counting and CIGAR scoring compare raw characters");
	HIP_TRY(hipSetDevice(ix->device));
	hipStream_t st = ws->stream;
	// ---- buffers, grown on demand -------------------------------------------------------------------------------------------
	auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const int64_t task_cap = n / 4 + 4096;                 // rescue windows: room for one per two pairs (more: those pairs go to the host)
	if (nc + task_cap + 1 > ws->aln_cand_capacity) {
		if (ws->d_aln_cand) HIP_TRY(hipFree(ws->d_aln_cand));
		ws->d_aln_cand = nullptr;
		int64_t cap = std::max<int64_t>(nc + nc / 4, 3 * ws->max_reads) + task_cap + task_cap / 4 + ws->max_reads / 4 + 4096;
		HIP_TRY(hipMalloc(&ws->d_aln_cand, up(4 * (size_t)cap) * 10 + up(8 * (size_t)cap) + up((size_t)cap) * 2 + up((size_t)cap * KG_ALN_CIGAR_MAX)));      // (cap >= n: the slab of the pair list holds n / 2 entries)
		ws->aln_cand_capacity = cap;
	}
	if (n + 2 > ws->aln_read_capacity) {
		if (ws->d_aln_read) HIP_TRY(hipFree(ws->d_aln_read));
		for (void *p : {(void *)ws->d_spill, (void *)ws->d_jobs, (void *)ws->d_job_ops, (void *)ws->d_job_len, (void *)ws->d_tasks, (void *)ws->d_plans})
			if (p) HIP_TRY(hipFree(p));
		ws->d_aln_read = nullptr; ws->d_spill = nullptr; ws->d_jobs = nullptr; ws->d_job_ops = nullptr; ws->d_job_len = nullptr; ws->d_tasks = nullptr; ws->d_plans = nullptr;
		int64_t cap = std::max(n, ws->max_reads); cap += cap / 4 + 4096;
		HIP_TRY(hipMalloc(&ws->d_aln_read, 3 * up((size_t)cap) + up(4 * (size_t)cap) + sizeof(kg_aln_record) * (size_t)cap));
		ws->task_capacity = cap / 4 + 4096;
		HIP_TRY(hipMalloc(&ws->d_tasks, up(sizeof(RescueTask) * (size_t)ws->task_capacity) + up(8 * (size_t)ws->task_capacity) + up(4 * (size_t)ws->task_capacity) +
		                                    sizeof(kg_seed) * (size_t)ws->task_capacity * kAlnMaxSeeds));
		// about one candidate in ten waits for an alignment at 1 % error (one in three at 2 %): room for one per read
		ws->spill_capacity = cap + 4096;
		ws->job_capacity = cap + 4096;
		ws->ops_capacity = 64 * ws->job_capacity;
		HIP_TRY(hipMalloc((void **)&ws->d_spill, sizeof(AlnSpill) * (size_t)ws->spill_capacity));
		HIP_TRY(hipMalloc((void **)&ws->d_jobs, sizeof(NwJobDesc) * (size_t)ws->job_capacity));
		HIP_TRY(hipMalloc((void **)&ws->d_job_ops, (size_t)ws->ops_capacity + 1024));
		HIP_TRY(hipMalloc((void **)&ws->d_job_len, 4 * (size_t)ws->job_capacity));
		HIP_TRY(hipMalloc((void **)&ws->d_plans, sizeof(AlnPlan) * (size_t)ws->job_capacity + sizeof(AlnPiece) * 4 * (size_t)ws->job_capacity + 16 + sizeof(PartTask) * (size_t)ws->job_capacity));
		ws->aln_read_capacity = cap;
	}
	if (n_chunks > ws->chunk_capacity) {
		for (void *p : {(void *)ws->d_chunk_off, (void *)ws->d_chunk_paired, (void *)ws->d_chunk_stats})
			if (p) HIP_TRY(hipFree(p));
		ws->d_chunk_off = nullptr; ws->d_chunk_paired = nullptr; ws->d_chunk_stats = nullptr;
		int cap = n_chunks + n_chunks / 2 + 64;
		HIP_TRY(hipMalloc((void **)&ws->d_chunk_off, 8 * (size_t)(cap + 1)));
		HIP_TRY(hipMalloc((void **)&ws->d_chunk_paired, (size_t)cap));
		HIP_TRY(hipMalloc((void **)&ws->d_chunk_stats, sizeof(kg_chunk_stats) * (size_t)cap));
		ws->chunk_capacity = cap;
	}
	// (zeroed on THIS stream: a plain hipMemset runs on the null stream, asynchronously, unordered against the kernels below -- behind a
	//  long null-stream operation (the seeding groups' buffers are filled at kg_stream_open) it landed in the middle of the lane's first
	//  batch and took the counters of the rescue tasks / spills / NW jobs with it: pairs that needed mate rescue came out unpaired in
	//  4-11 of 16 runs, tools/stress_groups.py; latent since round 2)
	if (!ws->d_aln_ctl) { HIP_TRY(hipMalloc((void **)&ws->d_aln_ctl, 8 * 40)); HIP_TRY(hipMemsetAsync(ws->d_aln_ctl, 0, 8 * 40, st)); }
	HIP_TRY(hipMemcpyAsync(ws->d_chunk_off, chunk_off, 8 * (size_t)(n_chunks + 1), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(ws->d_chunk_paired, chunk_paired, (size_t)n_chunks, hipMemcpyHostToDevice, st));
	// ---- arguments ------------------------------------------------------------------------------------------------------------
	a.ix = ix->view;
	a.ix.text = ix->d_text;
	a.enc = ws->d_enc; a.read_off = ws->d_read_off; a.n_reads = n;
	a.chunk_off = ws->d_chunk_off; a.chunk_paired = ws->d_chunk_paired; a.n_chunks = n_chunks;
	a.all_paired = 1;
	for (int c = 0; c < n_chunks; ++c) a.all_paired = a.all_paired && chunk_paired[c] ? 1 : 0;
	a.cand_off = ws->d_cand_off; a.cands = ws->d_dense_cands; a.cand_seeds = ws->d_dense_seeds; a.n_cands = nc;
	a.contig_end = ix->d_contig_end; a.end_chr = ix->d_end_chr; a.n_ends = ix->n_ends;
	a.n_chr = (int)ix->contigs.size();
	a.chr_fwd_start = ix->d_chr_tab; a.chr_rev_start = ix->d_chr_tab + a.n_chr; a.chr_len = ix->d_chr_tab + 2 * a.n_chr;
	a.genome_size = ix->l_pac; a.two_genome_size = 2 * ix->l_pac;
	a.est_distance = est_distance; a.max_insert = max_insert; a.max_gaps = max_gaps;
	a.multi_hit = multi_hit ? 1 : 0; a.unset_flag = unset_flag;
	{ static const bool nopart = getenv("KG_DBG_NO_PARTITION") != nullptr; a.dbg_no_partition = nopart ? 1 : 0; }
	{ static const bool scan = getenv("KG_RESCUE_SCAN") != nullptr; a.dbg_rescue_scan = scan ? 1 : 0; }
	{ static const bool no_heavy = getenv("KG_ALN_NO_HEAVY") != nullptr; a.dbg_no_heavy = no_heavy ? 1 : 0; }
	{ static const bool plan_group = getenv("KG_ALN_PLAN_GROUP") != nullptr; a.dbg_plan_group = plan_group ? 1 : 0; }
	{ static const int pair_heavy = getenv("KG_ALN_PAIR_HEAVY") ? std::max(1, atoi(getenv("KG_ALN_PAIR_HEAVY"))) : 32; a.pair_heavy = pair_heavy; }
	{ static const int finish_form = getenv("KG_ALN_FINISH_LANES") ? 1 : getenv("KG_ALN_FINISH_WAVE") ? 2 : getenv("KG_ALN_FINISH_G16") ? 3 : 0; a.dbg_finish_lanes = finish_form; }
	// (KG_ALN_INLINE: a candidate whose alignments are all at most 8 x 8 makes them in its planning lane and is finished at once.  Built and measured
	//  in round 6: 24 % fewer parked candidates, but aln_plan 31 -> 47 ms per step and aln_finish unchanged -- the stage's kernels are bound by
	//  their heaviest lanes, not by the number of candidates -- so it is off unless asked for; profiles/r06i_*)
	{ static const bool inline_on = getenv("KG_ALN_INLINE") != nullptr; a.dbg_no_inline = inline_on ? 0 : 1; }
	a.extra_capacity = 0;                                  // (set below, once the pinned array of this call is known)
	a.mapq_tab = ix->d_mapq_tab;
	{
		char *p = (char *)ws->d_aln_cand;
		size_t cap = (size_t)ws->aln_cand_capacity;
		a.c_score = (int32_t *)p; p += up(4 * cap);
		a.c_mate = (int32_t *)p; p += up(4 * cap);
		a.c_read = (int32_t *)p; p += up(4 * cap);
		a.rep_score = (int32_t *)p; p += up(4 * cap);
		a.rep_chr = (int32_t *)p; p += up(4 * cap);
		a.rep_pos = (int64_t *)p; p += up(8 * cap);
		a.rep_fwd = (uint8_t *)p; p += up(cap);
		a.rep_cigar_len = (uint8_t *)p; p += up(cap);
		a.rep_cigar = p; p += up(cap * KG_ALN_CIGAR_MAX);
		static const bool no_bins = getenv("KG_ALN_NO_BINS") != nullptr;      // A/B aid: aln_plan_kernel takes the candidates in their own order
		a.plan_order = no_bins ? nullptr : (int32_t *)p;
		p += up(4 * cap);
		static const bool no_fast = getenv("KG_ALN_NO_FAST") != nullptr;      // A/B aid: every candidate takes the general plan kernel
		a.plan_slow = no_fast ? nullptr : (int32_t *)p;
		p += up(4 * cap);
		// the trivial pairs (one candidate per mate, mated, both decided in registers) are finished by aln_trivial_kernel; the other pairs and
		// their candidates are listed for the kernels behind it.  Paired batches only; KG_ALN_NO_TRIVIAL (A/B aid): every pair takes the general kernels
		static const bool no_trivial = getenv("KG_ALN_NO_TRIVIAL") != nullptr;
		const bool trivial = a.all_paired && !no_trivial && !no_fast;
		a.slow_cands = trivial ? (int32_t *)p : nullptr;
		p += up(4 * cap);
		a.slow_pairs = trivial ? (int32_t *)p : nullptr;
		char *q = (char *)ws->d_aln_read;
		size_t rcap = (size_t)ws->aln_read_capacity;
		a.r_host = (uint8_t *)q; q += up(rcap);
		a.r_pending = (uint8_t *)q; q += up(rcap);
		a.resc_n = (uint8_t *)q; q += up(rcap);
		a.resc_off = (int32_t *)q; q += up(4 * rcap);
		a.records = (kg_aln_record *)q;
		char *t = (char *)ws->d_tasks;
		size_t tcap = (size_t)ws->task_capacity;
		a.tasks = (RescueTask *)t; t += up(sizeof(RescueTask) * tcap);
		a.resc_posdiff = (int64_t *)t; t += up(8 * tcap);
		a.resc_count = (int32_t *)t; t += up(4 * tcap);
		a.resc_seeds = (kg_seed *)t;
		a.task_capacity = std::min<int64_t>(ws->task_capacity, task_cap);
	}
	a.spill = ws->d_spill; a.spill_capacity = ws->spill_capacity;
	a.jobs = ws->d_jobs; a.job_capacity = ws->job_capacity; a.ops_capacity = ws->ops_capacity;
	{
		// test aid: the lists behave as if they were this short (the allocations stay), so that the overflow paths -- reads handed back to the
		// host, empty entries for what was reserved inside the lists -- can be driven by a small batch (tests/test_sam_gpu.py)
		static const char *e_sp = getenv("KG_DBG_SPILL_CAPACITY"), *e_job = getenv("KG_DBG_JOB_CAPACITY"), *e_ops = getenv("KG_DBG_OPS_CAPACITY");
		if (e_sp) a.spill_capacity = std::min<int64_t>(a.spill_capacity, atoll(e_sp));
		if (e_job) a.job_capacity = std::min<int64_t>(a.job_capacity, atoll(e_job));
		if (e_ops) a.ops_capacity = std::min<int64_t>(a.ops_capacity, atoll(e_ops));
	}
	a.ctl = ws->d_aln_ctl;
	a.nw_ops = ws->d_job_ops; a.nw_len = ws->d_job_len;
	a.plans = (AlnPlan *)ws->d_plans; a.pieces = (AlnPiece *)((char *)ws->d_plans + sizeof(AlnPlan) * (size_t)ws->job_capacity);
	a.part_tasks = (PartTask *)((char *)a.pieces + ((sizeof(AlnPiece) * 4 * (size_t)ws->job_capacity + 15) & ~(size_t)15));
	a.chunk_stats = ws->d_chunk_stats;
	HIP_TRY(launch_align_front(a, ix->n_cu, st));
	// ---- gap closing: the NW kernels on the job descriptors, fragments read in place --------------------------------------------
	{
		NwArgs w;
		w.desc = ws->d_jobs; w.text2 = ix->d_text; w.n_dev = ws->d_aln_ctl + 1;
		w.f1 = (const char *)ws->d_enc; w.off1 = nullptr; w.f2 = nullptr; w.off2 = nullptr;
		w.n = a.job_capacity;
		w.ops = ws->d_job_ops; w.aln_len = ws->d_job_len;
		w.big_lds_bytes = nw_big_lds_bytes(kAlnMaxFrag);
		w.gb_offset_words = 0;
		w.dir_words_per_wave = nw_dir_words(kAlnMaxFrag);
		int per_cu = std::max(1, std::min(16, (160 * 1024) / std::max(w.big_lds_bytes, 1024)));
		w.big_waves = (int)std::min<int64_t>((int64_t)ix->n_cu * per_cu, ws->job_capacity);
		NwScratch *sc = nullptr;
		int rc = nw_acquire(ix, 3 * (size_t)ws->job_capacity, (size_t)w.big_waves * (size_t)w.dir_words_per_wave, st, &sc);
		if (rc != KG_OK) return rc;
		w.big_list = sc->lists; w.queue = sc->queue; w.dir_scratch = sc->dir;
		kt_begin(KT_NW, st);
		hipError_t e = launch_nw_batch(w, ix->n_cu, st);
		kt_end(KT_NW, st);
		hipError_t e2 = nw_submitted(ix, sc, st);
		if (e != hipSuccess || e2 != hipSuccess) return fail(KG_ERR_NO_DEVICE, "kg_align_batch: %s", hipGetErrorString(e != hipSuccess ? e : e2));
	}
	// -m: the further records of a read take slots behind the n per-read ones, as many as both arrays hold
	if (multi_hit) a.extra_capacity = std::max<int64_t>(0, std::min<int64_t>(ws->aln_read_capacity, host_record_capacity) - n);
	HIP_TRY(launch_align_back(a, ix->n_cu, st));
	return KG_OK;
}

// Replaces, for a batch, what ReadMapping() does per read between chaining and the SAM text (reference src/Mapping.cpp:542-578);
// see align_kernels.hip for the kernel <-> reference correspondence.
int kg_align_batch(kg_workspace *ws, const int64_t *chunk_off, const uint8_t *chunk_paired, int n_chunks, int est_distance, int max_insert,
                   int max_gaps, int multi_hit, int unset_flag, const kg_aln_record **records, kg_chunk_stats *chunk_stats)
{
	if (!ws || !chunk_off || !chunk_paired || !records || !chunk_stats || n_chunks <= 0) return fail(KG_ERR_ARG, "kg_align_batch: bad argument");
	*records = nullptr;
	if (ws->last_reads <= 0 || ws->last_cands < 0) return fail(KG_ERR_ARG, "kg_align_batch: no chained batch on this workspace (kg_seed_batch + kg_candidates_batch first)");
	kg_index *ix = ws->ix;
	const int64_t n = ws->last_reads;
	if (chunk_off[0] != 0 || chunk_off[n_chunks] != n) return fail(KG_ERR_ARG, "kg_align_batch: the chunks do not cover the %lld reads of the batch", (long long)n);
	for (int c = 0; c < n_chunks; ++c) {
		if (chunk_off[c + 1] < chunk_off[c]) return fail(KG_ERR_ARG, "kg_align_batch: chunk offsets must not decrease");
		if (chunk_paired[c] && ((chunk_off[c + 1] - chunk_off[c]) & 1)) return fail(KG_ERR_ARG, "kg_align_batch: a paired chunk holds an odd number of reads");
	}
	HIP_TRY(hipSetDevice(ix->device));
	hipStream_t st = ws->stream;
	{
		const int slot = ws->ring_rec_at;
		ws->ring_rec_at = (ws->ring_rec_at + 1) % kg_workspace::kRing;
		if (n > ws->ring_record_capacity[slot]) {
			if (ws->ring_records[slot]) HIP_TRY(hipHostFree(ws->ring_records[slot]));
			ws->ring_records[slot] = nullptr;
			int64_t cap = std::max<int64_t>(n + n / 4 + 4096, std::min<int64_t>(ws->max_reads, 1000000));
			HIP_TRY(hipHostMalloc((void **)&ws->ring_records[slot], sizeof(kg_aln_record) * (size_t)cap, hipHostMallocDefault));
			ws->ring_record_capacity[slot] = cap;
		}
		ws->h_records = ws->ring_records[slot];
	}
	AlnArgs a;
	int rc = kgi_align_resident(ws, chunk_off, chunk_paired, n_chunks, est_distance, max_insert, max_gaps, multi_hit, unset_flag,
	                            ws->ring_record_capacity[(ws->ring_rec_at + kg_workspace::kRing - 1) % kg_workspace::kRing], a);
	if (rc != KG_OK) return rc;
	HIP_TRY(hipMemcpyAsync(ws->h_records, a.records, sizeof(kg_aln_record) * (size_t)n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(chunk_stats, ws->d_chunk_stats, sizeof(kg_chunk_stats) * (size_t)n_chunks, hipMemcpyDeviceToHost, st));
	if (multi_hit) {
		unsigned long long *h = ws->h_small;
		HIP_TRY(hipMemcpyAsync(&h[0], ws->d_aln_ctl + 7, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(kgi_sync(ws));
		int64_t k = (int64_t)std::min<unsigned long long>(h[0], (unsigned long long)a.extra_capacity);
		if (k > 0) HIP_TRY(hipMemcpyAsync(ws->h_records + n, a.records + n, sizeof(kg_aln_record) * (size_t)k, hipMemcpyDeviceToHost, st));
	}
	HIP_TRY(kgi_sync(ws));
	*records = ws->h_records;
	return KG_OK;
}

int kg_align_reasons(kg_workspace *ws, uint64_t out[16])
{
	if (!ws || !out) return fail(KG_ERR_ARG, "kg_align_reasons: null argument");
	for (int i = 0; i < 16; ++i) out[i] = 0;
	if (!ws->d_aln_ctl) return KG_OK;
	HIP_TRY(hipSetDevice(ws->ix->device));
	HIP_TRY(hipMemcpy(out, ws->d_aln_ctl + 8, 8 * 16, hipMemcpyDeviceToHost));
	return KG_OK;
}

int kg_nw_batch_device(kg_index *ix, const char *d_frag1, const int64_t *d_off1, const char *d_frag2, const int64_t *d_off2,
                       int64_t n, int64_t max_len, uint8_t *d_ops, int32_t *d_aln_len, void *stream)
{
	if (!ix) return fail(KG_ERR_ARG, "kg_nw_batch_device: null index");
	if (n < 0 || n > 0x7fffffff) return fail(KG_ERR_ARG, "kg_nw_batch_device: bad pair count");
	if (n == 0) return KG_OK;
	if (!d_frag1 || !d_off1 || !d_frag2 || !d_off2 || !d_ops || !d_aln_len) return fail(KG_ERR_ARG, "kg_nw_batch_device: null buffer");
	if (max_len < 0 || max_len > 0x3fffffff) return fail(KG_ERR_ARG, "kg_nw_batch_device: bad max_len");
	HIP_TRY(hipSetDevice(ix->device));
	return nw_run(ix, d_frag1, d_off1, d_frag2, d_off2, n, max_len, d_ops, d_aln_len, (hipStream_t)stream);
}

int kg_nw_batch(kg_index *ix, const char *frag1, const int64_t *off1, const char *frag2, const int64_t *off2, int64_t n,
                uint8_t *ops, int32_t *aln_len)
{
	if (!ix) return fail(KG_ERR_ARG, "kg_nw_batch: null index");
	if (n < 0) return fail(KG_ERR_ARG, "kg_nw_batch: bad pair count");
	if (n == 0) return KG_OK;
	if (!frag1 || !off1 || !frag2 || !off2 || !ops || !aln_len) return fail(KG_ERR_ARG, "kg_nw_batch: null argument");
	if (off1[0] != 0 || off2[0] != 0) return fail(KG_ERR_ARG, "kg_nw_batch: offsets must start at 0");
	int64_t b1 = off1[n], b2 = off2[n], max_len = 0;
	for (int64_t i = 0; i < n; ++i) {
		int64_t m = off1[i + 1] - off1[i], q = off2[i + 1] - off2[i];
		if (m < 0 || q < 0) return fail(KG_ERR_ARG, "kg_nw_batch: offsets must be non-decreasing");
		max_len = std::max(max_len, std::max(m, q));
	}
	HIP_TRY(hipSetDevice(ix->device));
	// one cached device block per call: [frag1 | frag2 | off1 | off2 | ops | aln_len], 256-byte aligned parts
	auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
	size_t p_f1 = 0, p_f2 = p_f1 + up((size_t)b1 + 16), p_o1 = p_f2 + up((size_t)b2 + 16), p_o2 = p_o1 + up(8 * (size_t)(n + 1)),
	       p_ops = p_o2 + up(8 * (size_t)(n + 1)), p_len = p_ops + up((size_t)(b1 + b2) + 16), total = p_len + up(4 * (size_t)n);
	NwScratch *io = nullptr;
	{
		std::lock_guard<std::mutex> lock(ix->nw_mu);
		for (NwScratch *s : ix->nw_pool)
			if (!s->io_busy && s->io_bytes >= total) { io = s; break; }
		if (!io) {
			for (NwScratch *s : ix->nw_pool)
				if (!s->io_busy) { io = s; break; }
			if (!io) {
				io = new NwScratch();
				HIP_TRY(hipEventCreateWithFlags(&io->done, hipEventDisableTiming));
				HIP_TRY(hipMalloc((void **)&io->queue, 8 * (size_t)kNwQueueWords));
				ix->nw_pool.push_back(io);
			}
			if (io->io) HIP_TRY(hipFree(io->io));
			io->io = nullptr;
			size_t want = total + total / 4;
			HIP_TRY(hipMalloc((void **)&io->io, want));
			io->io_bytes = want;
		}
		io->io_busy = true;   // holds the staging block; the kernels' own scratch comes from nw_acquire below
	}
	char *base = io->io;
	// a stream of the staging block's own: the call used to run on the null stream and end in hipDeviceSynchronize(), i.e. every one of
	// the ~1300 small calls a 100 M-read run makes for the reads handed back waited for whatever ALL the lanes had in flight
	hipError_t e = hipSuccess;
	if (!io->io_stream) e = hipStreamCreateWithFlags(&io->io_stream, hipStreamNonBlocking);
	hipStream_t st = io->io_stream;
	if (e == hipSuccess) e = hipMemcpyAsync(base + p_f1, frag1, (size_t)b1, hipMemcpyHostToDevice, st);
	if (e == hipSuccess) e = hipMemcpyAsync(base + p_f2, frag2, (size_t)b2, hipMemcpyHostToDevice, st);
	if (e == hipSuccess) e = hipMemcpyAsync(base + p_o1, off1, 8 * (size_t)(n + 1), hipMemcpyHostToDevice, st);
	if (e == hipSuccess) e = hipMemcpyAsync(base + p_o2, off2, 8 * (size_t)(n + 1), hipMemcpyHostToDevice, st);
	int rc = KG_OK;
	if (e != hipSuccess) rc = fail(KG_ERR_NO_DEVICE, "kg_nw_batch: %s", hipGetErrorString(e));
	if (rc == KG_OK)
		rc = nw_run(ix, base + p_f1, (const int64_t *)(base + p_o1), base + p_f2, (const int64_t *)(base + p_o2), n, max_len,
		            (uint8_t *)(base + p_ops), (int32_t *)(base + p_len), st);
	if (rc == KG_OK) {
		e = hipMemcpyAsync(ops, base + p_ops, (size_t)(b1 + b2), hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipMemcpyAsync(aln_len, base + p_len, 4 * (size_t)n, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipStreamSynchronize(st);
		if (e != hipSuccess) rc = fail(KG_ERR_NO_DEVICE, "kg_nw_batch: %s", hipGetErrorString(e));
	}
	{
		std::lock_guard<std::mutex> lock(ix->nw_mu);
		io->io_busy = false;
	}
	return rc;
}

}  // extern "C"
