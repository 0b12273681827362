// abi_long.hip -- kg_longread_batch: the per-read report of long reads (-pacbio) on the device (declared in include/kart_amd.h).
//
// Replaces, for a batch, what ReadMapping()'s bPacBioData branch does per read between chaining and the SAM text
// (src/Mapping.cpp:513-530: RemoveRedundantCandidates, GenMappingReport, SetSingleAlignmentFlag, EvaluateMAPQ); the kernel <->
// reference correspondence is in long_kernels.hip, GenerateNormalPairAlignment is frag_kernels.hip on device-resident requests.
#include "abi_internal.hpp"
#include "frag_kernels.hpp"
#include "long_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define fail kg_fail

namespace {

// a device block that keeps its capacity
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	int ensure(size_t bytes)
	{
		if (bytes <= cap) return KG_OK;
		if (p) HIP_TRY(hipFree(p));
		p = nullptr;
		cap = 0;
		const size_t want = bytes + bytes / 4 + 4096;
		HIP_TRY(hipMalloc(&p, want));
		cap = want;
		return KG_OK;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct LongScratch {
	DevBuf cand_lc, lcs, pool, req_f1, req_g, req_rl, req_gl, req_oo, ctl, r_host, r_best, cig_bytes, records, scan_temp;
	DevBuf frag_work, elems, cigar;
	char *h_cigar[kg_workspace::kRing] = {nullptr, nullptr, nullptr, nullptr};     // page-locked, in rotation with the records
	size_t h_cigar_cap[kg_workspace::kRing] = {0, 0, 0, 0};
	unsigned long long *h_why = nullptr;                                             // page-locked, 6 words
	unsigned long long reasons[8] = {0, 0, 0, 0, 0, 0, 0, 0};                       // running tallies (kg_longread_reasons)
	unsigned long long frag_why[6] = {0, 0, 0, 0, 0, 0};                            // ... of the fragment kernels' hand-backs (FC_WHY)
	unsigned long long reads = 0, host_reads = 0;
};

}  // namespace

extern "C" void kgi_long_release(kg_workspace *ws)
{
	LongScratch *ls = static_cast<LongScratch *>(ws->lr);
	if (!ls) return;
	for (DevBuf *b : {&ls->cand_lc, &ls->lcs, &ls->pool, &ls->req_f1, &ls->req_g, &ls->req_rl, &ls->req_gl, &ls->req_oo, &ls->ctl, &ls->r_host, &ls->r_best, &ls->cig_bytes,
	                  &ls->records, &ls->scan_temp, &ls->frag_work, &ls->elems, &ls->cigar})
		b->release();
	for (int i = 0; i < kg_workspace::kRing; ++i)
		if (ls->h_cigar[i]) (void)hipHostFree(ls->h_cigar[i]);
	if (ls->h_why) (void)hipHostFree(ls->h_why);
	delete ls;
	ws->lr = nullptr;
}

extern "C" int kg_longread_batch(kg_workspace *ws, const kg_aln_record **records, const char **cigar_pool, int64_t *cigar_bytes, int64_t *n_host_reads)
{
	if (!ws || !records || !cigar_pool || !cigar_bytes) return fail(KG_ERR_ARG, "kg_longread_batch: null argument");
	*records = nullptr; *cigar_pool = nullptr; *cigar_bytes = 0;
	if (n_host_reads) *n_host_reads = 0;
	if (ws->last_reads <= 0 || ws->last_cands < 0) return fail(KG_ERR_ARG, "kg_longread_batch: no chained batch on this workspace (kg_seed_batch + kg_candidates_batch first)");
	if (!ws->last_ascii) return fail(KG_ERR_ARG, "kg_longread_batch: the resident reads must be characters (KG_INPUT_ASCII)");
	if (!ws->last_pacbio) return fail(KG_ERR_ARG, "kg_longread_batch: the resident candidates were not chained for long reads (kg_candidates_batch with pacbio != 0)");
	kg_index *ix = ws->ix;
	if (!ix->d_text) return fail(KG_ERR_ARG, "kg_longread_batch: the index holds no text");
	HIP_TRY(hipSetDevice(ix->device));
	hipStream_t st = ws->stream;
	if (!ws->lr) ws->lr = new LongScratch();
	LongScratch *ls = static_cast<LongScratch *>(ws->lr);
	if (!ls->h_why) HIP_TRY(hipHostMalloc((void **)&ls->h_why, 8 * 8, hipHostMallocDefault));
	const int64_t n = ws->last_reads, n_cands = ws->last_cands, n_cseeds = ws->last_cand_seeds;
	auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
	int rc;
	const int64_t pool_cap = 2 * n_cseeds + 3 * n_cands + 64, req_cap = n_cseeds + 2 * n_cands + 64;
#define ENSURE(buf, bytes) do { if ((rc = (buf).ensure(bytes)) != KG_OK) return rc; } while (0)
	ENSURE(ls->cand_lc, 4 * (size_t)(n_cands + 1));
	ENSURE(ls->lcs, sizeof(LrCand) * (size_t)(n_cands + 1));
	ENSURE(ls->pool, sizeof(LrPair) * (size_t)pool_cap);
	ENSURE(ls->req_f1, 8 * (size_t)req_cap); ENSURE(ls->req_g, 8 * (size_t)req_cap); ENSURE(ls->req_oo, 8 * (size_t)req_cap);
	ENSURE(ls->req_rl, 4 * (size_t)req_cap); ENSURE(ls->req_gl, 4 * (size_t)req_cap);
	ENSURE(ls->ctl, 8 * LC_WORDS);
	ENSURE(ls->r_host, (size_t)n + 64); ENSURE(ls->r_best, 4 * (size_t)n + 64); ENSURE(ls->cig_bytes, 8 * (size_t)(n + 1));
	ENSURE(ls->records, sizeof(kg_aln_record) * (size_t)n);
	ENSURE(ls->scan_temp, long_scan_temp_bytes(n + 1));
	// the page-locked results rotate with the workspace's other result arrays (valid while the next kRing - 1 batches pass)
	const int slot = ws->ring_rec_at;
	ws->ring_rec_at = (ws->ring_rec_at + 1) % kg_workspace::kRing;
	if (n > ws->ring_record_capacity[slot]) {
		if (ws->ring_records[slot]) HIP_TRY(hipHostFree(ws->ring_records[slot]));
		ws->ring_records[slot] = nullptr;
		const int64_t cap = n + n / 4 + 4096;
		HIP_TRY(hipHostMalloc((void **)&ws->ring_records[slot], sizeof(kg_aln_record) * (size_t)cap, hipHostMallocDefault));
		ws->ring_record_capacity[slot] = cap;
	}
	ws->h_records = ws->ring_records[slot];

	LrArgs a;
	a.enc = ws->d_enc; a.read_off = ws->d_read_off; a.n_reads = n;
	a.cand_off = ws->d_cand_off; a.cands = ws->d_dense_cands; a.seeds = ws->d_dense_seeds; a.n_cands = n_cands;
	a.text = ix->d_text; a.genome_size = ix->l_pac; a.two_genome_size = 2 * ix->l_pac;
	a.contig_end = ix->d_contig_end; a.end_chr = ix->d_end_chr; a.n_ends = ix->n_ends;
	a.chr_fwd_start = ix->d_chr_tab; a.n_chr = (int)ix->contigs.size();
	a.cand_lc = (int32_t *)ls->cand_lc.p; a.lcs = (LrCand *)ls->lcs.p; a.pool = (LrPair *)ls->pool.p; a.pool_capacity = pool_cap;
	a.req_f1 = (int64_t *)ls->req_f1.p; a.req_g = (int64_t *)ls->req_g.p; a.req_rl = (int32_t *)ls->req_rl.p; a.req_gl = (int32_t *)ls->req_gl.p;
	a.req_oo = (int64_t *)ls->req_oo.p; a.req_capacity = req_cap;
	a.ctl = (unsigned long long *)ls->ctl.p;
	a.ops = nullptr; a.aln_len = nullptr; a.runs = nullptr; a.status = nullptr; a.elems = nullptr; a.elem_capacity = 0;
	a.r_host = (uint8_t *)ls->r_host.p; a.cig_bytes = (int64_t *)ls->cig_bytes.p; a.r_best = (int32_t *)ls->r_best.p;
	a.records = (kg_aln_record *)ls->records.p; a.cigar = nullptr;

	// ---- candidates that take part, normal pairs, pass 1 (the fragment requests) ----
	HIP_TRY(hipMemsetAsync(a.ctl, 0, 8 * LC_WORDS, st));
	HIP_TRY(launch_long_plan(a, ix->n_cu, st));
	unsigned long long *h = ws->h_small;
	HIP_TRY(hipMemcpyAsync(h, a.ctl, 8 * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(kgi_sync(ws));
	const int64_t n_req = (int64_t)std::min<unsigned long long>(h[LC_REQ], (unsigned long long)req_cap), cols = (int64_t)h[LC_COLS], pool_used = (int64_t)h[LC_POOL];
	const int64_t max_len = std::max<int64_t>(1, (int64_t)h[LC_MAXLEN]);
	if (n_req > 0x7ffffff0) return fail(KG_ERR_ARG, "kg_longread_batch: too many fragment pairs in one batch (%lld)", (long long)n_req);

	// ---- GenerateNormalPairAlignment for the requests (frag_kernels.hip), on the arrays pass 1 left on the device ----
	const int64_t pool_waves = frag_pool_waves(n_req, ix->n_cu);        // (what the partition kernel's waves leave unused of the stretches they reserve)
	const int64_t task_cap = n_req + n_req / 2 + cols / 300 + 4096, piece_cap = 4 * n_req + cols / 8 + 4096 + pool_waves * kFragPieceChunk, job_cap = 2 * n_req + cols / 16 + 4096 + pool_waves * kFragJobChunk, jops_cap = cols + 4096 + pool_waves * kFragOpsChunk;
	const size_t w_tasks = 0, w_pieces = w_tasks + up(sizeof(FragTask) * (size_t)task_cap), w_jobs = w_pieces + up(sizeof(FragPiece) * (size_t)piece_cap),
	             w_jops = w_jobs + up(sizeof(NwJobDesc) * (size_t)job_cap), w_jlen = w_jops + up((size_t)jops_cap + 64), w_ctl = w_jlen + up(4 * (size_t)job_cap),
	             w_status = w_ctl + up(8 * FC_WORDS), w_ops = w_status + up((size_t)n_req + 64), w_len = w_ops + up((size_t)cols + 64), w_runs = w_len + up(4 * (size_t)n_req + 64), work_total = w_runs + up(4 * (size_t)n_req + 64);
	ENSURE(ls->frag_work, work_total);
	// merged CIGAR elements: at most two per pair and one per run of a column string -- a run every ~6 columns at 15 % error; a quarter of the columns
	// is room for a run every second column of every alignment (a candidate that needs more goes back to the host: LC_R_ELEMS)
	const int64_t elem_cap = std::min<int64_t>(pool_used, pool_cap) * 2 + cols / 4 + 4096;
	ENSURE(ls->elems, 4 * (size_t)elem_cap);
	char *fw = (char *)ls->frag_work.p;
	a.ops = (const uint8_t *)(fw + w_ops); a.aln_len = (const int32_t *)(fw + w_len); a.runs = (const int32_t *)(fw + w_runs); a.status = (const uint8_t *)(fw + w_status);
	a.elems = (uint32_t *)ls->elems.p; a.elem_capacity = elem_cap;
	bool have_why = false;
	if (n_req > 0) {
		FragArgs f;
		f.f1 = (const char *)ws->d_enc; f.off1 = a.req_f1; f.rlen = a.req_rl; f.gpos = a.req_g; f.glen = a.req_gl; f.n = n_req;
		f.text = ix->d_text; f.two_genome_size = 2 * ix->l_pac;
		f.pacbio = 1; f.max_gaps = 0;
		static const bool frag_prof = getenv("KG_FRAG_PROF") != nullptr;     // diagnostics: wave cycles per phase of the partition kernel
		f.prof = frag_prof ? 1 : 0;
		static const bool no_fast_pairs = getenv("KG_FRAG_NO_FAST_PAIRS") != nullptr;
		f.no_fast_pairs = no_fast_pairs ? 1 : 0;
		f.tasks = (FragTask *)(fw + w_tasks); f.task_capacity = task_cap;
		f.pieces = (FragPiece *)(fw + w_pieces); f.piece_capacity = piece_cap;
		f.jobs = (NwJobDesc *)(fw + w_jobs); f.job_capacity = job_cap; f.ops_capacity = jops_cap;
		f.job_ops = (uint8_t *)(fw + w_jops); f.job_len = (int32_t *)(fw + w_jlen);
		f.ctl = (unsigned long long *)(fw + w_ctl);
		f.status = (uint8_t *)(fw + w_status);
		f.ops = (uint8_t *)(fw + w_ops); f.ops_off = a.req_oo; f.aln_len = (int32_t *)(fw + w_len); f.runs = (int32_t *)(fw + w_runs);
		HIP_TRY(launch_frag_partition(f, ix->n_cu, st));
		NwArgs w;
		w.desc = f.jobs; w.text2 = ix->d_text; w.n_dev = f.ctl + FC_JOBS;
		w.f1 = f.f1; w.off1 = nullptr; w.f2 = nullptr; w.off2 = nullptr;
		w.n = job_cap;
		w.ops = f.job_ops; w.aln_len = f.job_len;
		rc = kgi_nw_launch(ix, w, max_len, st);
		if (rc != KG_OK) return rc;
		if (getenv("KG_LONG_DEBUG_JOBS")) {
			// diagnostics: the NW jobs of this batch by their longer side -- count and cells per class
			unsigned long long nj = 0;
			HIP_TRY(hipStreamSynchronize(st));
			(void)hipMemcpy(&nj, f.ctl + FC_JOBS, 8, hipMemcpyDeviceToHost);
			nj = std::min<unsigned long long>(nj, (unsigned long long)job_cap);
			std::vector<NwJobDesc> jd((size_t)nj);
			if (nj) (void)hipMemcpy(jd.data(), f.jobs, sizeof(NwJobDesc) * (size_t)nj, hipMemcpyDeviceToHost);
			static const int edge[] = {0, 8, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 1 << 30};
			unsigned long long cnt[12] = {}, cells[12] = {};
			int longest = 0;
			for (const NwJobDesc &j : jd) {
				const int mx = std::max(j.m, j.n);
				int b = 0;
				while (mx > edge[b]) ++b;
				cnt[b]++; cells[b] += (unsigned long long)j.m * (unsigned long long)j.n;
				longest = std::max(longest, mx);
			}
			fprintf(stderr, "KG_LONG_DEBUG_JOBS %llu job slots (longest side %d):", nj, longest);
			for (int b = 0; b < 12; ++b) if (cnt[b]) fprintf(stderr, " <=%d: %llu jobs %.1f Mcells;", edge[b], cnt[b], 1e-6 * (double)cells[b]);
			fprintf(stderr, "\n");
		}
		HIP_TRY(launch_frag_stitch(f, ix->n_cu, st));
		HIP_TRY(hipMemcpyAsync(ls->h_why, f.ctl + FC_WHY, 8 * 6, hipMemcpyDeviceToHost, st));
		have_why = true;
		if (frag_prof) {
			unsigned long long pc[FC_WORDS];
			HIP_TRY(hipMemcpyAsync(pc, f.ctl, 8 * FC_WORDS, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			const double nt = (double)std::max(1ull, pc[FC_PROF + 5]);
			fprintf(stderr, "kg_longread_batch: %lld requests, %llu tasks, %llu pieces, %llu NW jobs | wave cycles per task: load+pack %.0f, diagonal scan %.0f, sort %.0f, normal pairs %.0f, pieces %.0f | runs per task %.1f, columns per task %.0f | seed filters in %.1f %% of the tasks (%.0f wave cycles each), lane-0 gap pairs in %.1f %% (%.0f each)\n",
			        (long long)n_req, pc[FC_PROF + 5], pc[FC_PIECES], pc[FC_JOBS], (double)pc[FC_PROF] / nt, (double)pc[FC_PROF + 1] / nt, (double)pc[FC_PROF + 2] / nt, (double)pc[FC_PROF + 3] / nt,
			        (double)pc[FC_PROF + 4] / nt, (double)pc[FC_PROF + 6] / nt, (double)pc[FC_PROF + 7] / nt,
			        100.0 * (double)pc[FC_PROF + 8] / nt, (double)pc[FC_PROF + 10] / (double)std::max(1ull, pc[FC_PROF + 8]), 100.0 * (double)pc[FC_PROF + 9] / nt, (double)pc[FC_PROF + 11] / (double)std::max(1ull, pc[FC_PROF + 9]));
		}
	}

	// ---- pass 2, the records, the CIGAR strings ----
	HIP_TRY(launch_long_finish(a, ix->n_cu, st));
	HIP_TRY(launch_long_scan(a, ls->scan_temp.p, ls->scan_temp.cap, st));
	HIP_TRY(hipMemcpyAsync(&h[8], a.cig_bytes + n, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(kgi_sync(ws));
	const int64_t text_bytes = (int64_t)h[8];
	ENSURE(ls->cigar, (size_t)text_bytes + 64);
	if ((size_t)text_bytes + 64 > ls->h_cigar_cap[slot]) {
		if (ls->h_cigar[slot]) HIP_TRY(hipHostFree(ls->h_cigar[slot]));
		ls->h_cigar[slot] = nullptr;
		ls->h_cigar_cap[slot] = 0;
		const size_t want = (size_t)text_bytes + (size_t)text_bytes / 4 + 65536;
		HIP_TRY(hipHostMalloc((void **)&ls->h_cigar[slot], want, hipHostMallocDefault));
		ls->h_cigar_cap[slot] = want;
	}
	a.cigar = (char *)ls->cigar.p;
	HIP_TRY(launch_long_text(a, ix->n_cu, st));
	HIP_TRY(hipMemcpyAsync(ws->h_records, a.records, sizeof(kg_aln_record) * (size_t)n, hipMemcpyDeviceToHost, st));
	if (text_bytes > 0) HIP_TRY(hipMemcpyAsync(ls->h_cigar[slot], a.cigar, (size_t)text_bytes, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(h, a.ctl, 8 * LC_WORDS, hipMemcpyDeviceToHost, st));
	HIP_TRY(kgi_sync(ws));
	if (getenv("KG_LONG_DEBUG_STATUS") && n_req > 0) {
		// diagnostics: the requests the fragment kernels sent back, and why
		std::vector<uint8_t> stv((size_t)n_req);
		std::vector<int32_t> rl((size_t)n_req), gl((size_t)n_req);
		(void)hipMemcpy(stv.data(), a.status, (size_t)n_req, hipMemcpyDeviceToHost);
		(void)hipMemcpy(rl.data(), a.req_rl, 4 * (size_t)n_req, hipMemcpyDeviceToHost);
		(void)hipMemcpy(gl.data(), a.req_gl, 4 * (size_t)n_req, hipMemcpyDeviceToHost);
		int64_t bad = 0;
		for (int64_t i = 0; i < n_req; ++i)
			if (stv[(size_t)i]) { if (bad++ < 12) fprintf(stderr, "KG_LONG_DEBUG_STATUS request %lld: status %d, read side %d, text side %d\n", (long long)i, stv[(size_t)i], rl[(size_t)i], gl[(size_t)i]); }
		fprintf(stderr, "KG_LONG_DEBUG_STATUS %lld of %lld requests sent back; why: %llu %llu %llu %llu %llu %llu\n", (long long)bad, (long long)n_req,
		        ls->h_why[0], ls->h_why[1], ls->h_why[2], ls->h_why[3], ls->h_why[4], ls->h_why[5]);
	}
	if (const char *dbg = getenv("KG_LONG_DEBUG")) {
		// diagnostics: the pairs, the requests' op strings (as runs) and the merged elements of read <KG_LONG_DEBUG> of this batch
		const int64_t r = atoll(dbg);
		if (r >= 0 && r < n) {
			std::vector<int64_t> co(2);
			(void)hipMemcpy(co.data(), ws->d_cand_off + r, 16, hipMemcpyDeviceToHost);
			for (int64_t c = co[0]; c < co[1]; ++c) {
				int32_t lc = -1;
				(void)hipMemcpy(&lc, a.cand_lc + c, 4, hipMemcpyDeviceToHost);
				kg_candidate cd;
				(void)hipMemcpy(&cd, ws->d_dense_cands + c, sizeof(cd), hipMemcpyDeviceToHost);
				fprintf(stderr, "KG_LONG_DEBUG read %lld candidate %lld: score %d, %d seeds, lc %d\n", (long long)r, (long long)c, cd.score, cd.count, lc);
				if (lc < 0) continue;
				LrCand k;
				(void)hipMemcpy(&k, a.lcs + lc, sizeof(k), hipMemcpyDeviceToHost);
				fprintf(stderr, "  state %d, %d pairs, score %d, chr %d pos %lld fwd %d, %d elements, %d text bytes\n", k.state, k.n_pairs, k.score, k.chr, (long long)k.pos, k.fwd, k.n_elems, k.text_bytes);
				std::vector<LrPair> P((size_t)std::max(0, k.n_pairs));
				if (k.n_pairs > 0) (void)hipMemcpy(P.data(), a.pool + k.pair_off, sizeof(LrPair) * P.size(), hipMemcpyDeviceToHost);
				for (size_t j = 0; j < P.size(); ++j) {
					const LrPair &q = P[j];
					fprintf(stderr, "  pair %zu: kind %d r[%d +%d] g[%lld +%d] v %d op %c", j, q.kind, q.rPos, q.rLen, (long long)q.gPos, q.gLen, q.v, q.op ? q.op : '.');
					if (q.kind == LP_REQ && q.v >= 0 && q.v < n_req) {
						int32_t L = 0; int64_t oo = 0; uint8_t stt = 0;
						(void)hipMemcpy(&L, a.aln_len + q.v, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&oo, a.req_oo + q.v, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&stt, a.status + q.v, 1, hipMemcpyDeviceToHost);
						std::vector<uint8_t> o((size_t)std::max(0, L));
						if (L > 0) (void)hipMemcpy(o.data(), a.ops + oo, (size_t)L, hipMemcpyDeviceToHost);
						fprintf(stderr, " | request %d: status %d, %d columns:", q.v, stt, L);
						for (int t = 0; t < L;) { int e = t; while (e < L && o[(size_t)e] == o[(size_t)t]) ++e; fprintf(stderr, " %d%c", e - t, "MDI?"[o[(size_t)t] & 3]); t = e; }
					}
					fprintf(stderr, "\n");
				}
				std::vector<uint32_t> E((size_t)std::max(0, k.n_elems));
				if (k.n_elems > 0) (void)hipMemcpy(E.data(), a.elems + k.elem_off, 4 * E.size(), hipMemcpyDeviceToHost);
				fprintf(stderr, "  elements:");
				for (uint32_t e : E) fprintf(stderr, " %u%c", e >> 2, "MIDS"[e & 3]);
				fprintf(stderr, "\n");
			}
		}
	}
	int64_t n_host = 0;
	for (int64_t r = 0; r < n; ++r) n_host += ws->h_records[r].kind == KG_ALN_HOST;
	ls->reads += (unsigned long long)n; ls->host_reads += (unsigned long long)n_host;
	for (int i = 0; i < 5; ++i) ls->reasons[i] += h[LC_R_DASH + i];
	if (have_why) for (int i = 0; i < 6; ++i) ls->frag_why[i] += ls->h_why[i];
	*records = ws->h_records;
	*cigar_pool = ls->h_cigar[slot];
	*cigar_bytes = text_bytes;
	if (n_host_reads) *n_host_reads = n_host;
	return KG_OK;
#undef ENSURE
}

extern "C" int kg_longread_reasons(kg_workspace *ws, uint64_t out[16])
{
	if (!ws || !out) return fail(KG_ERR_ARG, "kg_longread_reasons: null argument");
	for (int i = 0; i < 16; ++i) out[i] = 0;
	const LongScratch *ls = static_cast<const LongScratch *>(ws->lr);
	if (!ls) return KG_OK;
	out[0] = ls->reads; out[1] = ls->host_reads;
	for (int i = 0; i < 5; ++i) out[2 + i] = ls->reasons[i];
	for (int i = 0; i < 6; ++i) out[8 + i] = ls->frag_why[i];
	return KG_OK;
}
