// abi_frag.hip -- kg_fragments_batch: GenerateNormalPairAlignment for a batch of fragment pairs (declared in include/kart_amd.h).
#include "abi_internal.hpp"
#include "frag_kernels.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define fail kg_fail

// device scratch of one call, cached in a process-wide pool keyed by the index (calls are rare and large: one per batch)
namespace {
struct FragScratch {
	kg_index *ix = nullptr;
	bool busy = false;
	hipStream_t stream = nullptr;
	hipEvent_t done = nullptr;
	char *in = nullptr;            // [frag1 | off1 | gpos | glen | ops_off]
	size_t in_bytes = 0;
	char *work = nullptr;          // [tasks | pieces | jobs | job_ops | job_len | ctl | status | ops | aln_len]
	size_t work_bytes = 0;
};
double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
std::mutex g_frag_mu;
std::vector<FragScratch *> g_frag_pool;
}  // namespace

// the index is going away: its scratches (device memory, stream, event) go with it -- called from kg_index_destroy, which the caller
// must not run while calls on that index are in flight
extern "C" void kgi_frag_release(kg_index *ix)
{
	std::lock_guard<std::mutex> lk(g_frag_mu);
	for (size_t i = 0; i < g_frag_pool.size();) {
		FragScratch *s = g_frag_pool[i];
		if (s->ix != ix) { ++i; continue; }
		if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
		if (s->done) (void)hipEventDestroy(s->done);
		if (s->in) (void)hipFree(s->in);
		if (s->work) (void)hipFree(s->work);
		delete s;
		g_frag_pool.erase(g_frag_pool.begin() + (long)i);
	}
}

extern "C" int kg_fragments_batch(kg_index *ix, const char *frag1, const int64_t *off1, const int64_t *gpos, const int32_t *glen, int64_t n, int pacbio,
                                  int max_gaps, uint8_t *ops, const int64_t *ops_off, int32_t *aln_len, uint8_t *status)
{
	const double t_in = wall();
	if (!ix) return fail(KG_ERR_ARG, "kg_fragments_batch: null index");
	if (n < 0 || n > 0x7ffffff0) return fail(KG_ERR_ARG, "kg_fragments_batch: bad request count");
	if (n == 0) return KG_OK;
	if (!frag1 || !off1 || !gpos || !glen || !ops || !ops_off || !aln_len || !status) return fail(KG_ERR_ARG, "kg_fragments_batch: null argument");
	if (!ix->d_text) return fail(KG_ERR_ARG, "kg_fragments_batch: the index holds no text");
	if (off1[0] != 0) return fail(KG_ERR_ARG, "kg_fragments_batch: offsets must start at 0");
	int64_t cols = 0, max_len = 1;
	for (int64_t i = 0; i < n; ++i) {
		const int64_t m = off1[i + 1] - off1[i];
		if (m < 0 || glen[i] < 0) return fail(KG_ERR_ARG, "kg_fragments_batch: negative fragment length");
		if (gpos[i] < 0 || gpos[i] + glen[i] > 2 * ix->l_pac) return fail(KG_ERR_ARG, "kg_fragments_batch: genome fragment %lld outside the text", (long long)i);
		if (ops_off[i] != cols) return fail(KG_ERR_ARG, "kg_fragments_batch: ops_off[i] must be the sum of the columns (rLen + gLen) of the requests before i");
		cols += m + glen[i];
		max_len = std::max<int64_t>(max_len, std::max<int64_t>(m, glen[i]));
	}
	HIP_TRY(hipSetDevice(ix->device));
	const int64_t b1 = off1[n];
	auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
	// every request becomes at most a few dozen pieces per 100 columns; lists that run full send their request back (status 1)
	const int64_t pool_waves = frag_pool_waves(n, ix->n_cu);        // (what the partition kernel's waves leave unused of the stretches they reserve)
	const int64_t task_cap = n + n / 2 + cols / 300 + 4096, piece_cap = 4 * n + cols / 8 + 4096 + pool_waves * kFragPieceChunk, job_cap = 2 * n + cols / 16 + 4096 + pool_waves * kFragJobChunk, jops_cap = cols + 4096 + pool_waves * kFragOpsChunk;
	size_t p_f1 = 0, p_off = p_f1 + up((size_t)b1 + 64), p_g = p_off + up(8 * (size_t)(n + 1)), p_gl = p_g + up(8 * (size_t)n), p_oo = p_gl + up(4 * (size_t)n),
	       in_total = p_oo + up(8 * (size_t)n);
	size_t w_tasks = 0, w_pieces = w_tasks + up(sizeof(FragTask) * (size_t)task_cap), w_jobs = w_pieces + up(sizeof(FragPiece) * (size_t)piece_cap),
	       w_jops = w_jobs + up(sizeof(NwJobDesc) * (size_t)job_cap), w_jlen = w_jops + up((size_t)jops_cap + 64), w_ctl = w_jlen + up(4 * (size_t)job_cap),
	       w_status = w_ctl + up(8 * FC_WORDS), w_ops = w_status + up((size_t)n), w_len = w_ops + up((size_t)cols + 64), work_total = w_len + up(4 * (size_t)n);
	FragScratch *sc = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_frag_mu);
		for (FragScratch *s : g_frag_pool)
			if (s->ix == ix && !s->busy) { sc = s; break; }
		if (!sc) {
			sc = new FragScratch();
			sc->ix = ix;
			HIP_TRY(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
			HIP_TRY(hipEventCreateWithFlags(&sc->done, hipEventBlockingSync | hipEventDisableTiming));
			g_frag_pool.push_back(sc);
		}
		sc->busy = true;
	}
	struct Release { FragScratch *s; ~Release() { std::lock_guard<std::mutex> lk(g_frag_mu); s->busy = false; } } release{sc};
	if (in_total > sc->in_bytes) {
		if (sc->in) HIP_TRY(hipFree(sc->in));
		sc->in = nullptr;
		sc->in_bytes = in_total + in_total / 4;
		HIP_TRY(hipMalloc((void **)&sc->in, sc->in_bytes));
	}
	if (work_total > sc->work_bytes) {
		if (sc->work) HIP_TRY(hipFree(sc->work));
		sc->work = nullptr;
		sc->work_bytes = work_total + work_total / 4;
		HIP_TRY(hipMalloc((void **)&sc->work, sc->work_bytes));
	}
	hipStream_t st = sc->stream;
	static const bool prof = getenv("KG_FRAG_PROF") != nullptr;
	hipEvent_t pe[6] = {};
	auto mark = [&](int k) { if (prof) { (void)hipEventCreate(&pe[k]); (void)hipEventRecord(pe[k], st); } };
	const double t_q = prof ? wall() : 0;
	mark(0);
	HIP_TRY(hipMemcpyAsync(sc->in + p_f1, frag1, (size_t)b1, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(sc->in + p_off, off1, 8 * (size_t)(n + 1), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(sc->in + p_g, gpos, 8 * (size_t)n, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(sc->in + p_gl, glen, 4 * (size_t)n, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(sc->in + p_oo, ops_off, 8 * (size_t)n, hipMemcpyHostToDevice, st));
	FragArgs a;
	a.f1 = sc->in + p_f1; a.off1 = (const int64_t *)(sc->in + p_off); a.gpos = (const int64_t *)(sc->in + p_g); a.glen = (const int32_t *)(sc->in + p_gl);
	a.n = n;
	a.text = ix->d_text; a.two_genome_size = 2 * ix->l_pac;
	a.pacbio = pacbio ? 1 : 0; a.max_gaps = max_gaps;
	a.prof = prof ? 1 : 0;
	static const bool no_fast_pairs = getenv("KG_FRAG_NO_FAST_PAIRS") != nullptr;
	a.no_fast_pairs = no_fast_pairs ? 1 : 0;
	a.tasks = (FragTask *)(sc->work + w_tasks); a.task_capacity = task_cap;
	a.pieces = (FragPiece *)(sc->work + w_pieces); a.piece_capacity = piece_cap;
	a.jobs = (NwJobDesc *)(sc->work + w_jobs); a.job_capacity = job_cap; a.ops_capacity = jops_cap;
	a.job_ops = (uint8_t *)(sc->work + w_jops); a.job_len = (int32_t *)(sc->work + w_jlen);
	a.ctl = (unsigned long long *)(sc->work + w_ctl);
	a.status = (uint8_t *)(sc->work + w_status);
	a.ops = (uint8_t *)(sc->work + w_ops); a.ops_off = (const int64_t *)(sc->in + p_oo); a.aln_len = (int32_t *)(sc->work + w_len);
	mark(1);
	HIP_TRY(launch_frag_partition(a, ix->n_cu, st));
	mark(2);
	{
		// nw_alignment for the jobs the partition wrote: read side from the uploaded characters, genome side from the 2-bit text
		NwArgs w;
		w.desc = a.jobs; w.text2 = ix->d_text; w.n_dev = a.ctl + FC_JOBS;
		w.f1 = a.f1; w.off1 = nullptr; w.f2 = nullptr; w.off2 = nullptr;
		w.n = job_cap;
		w.ops = a.job_ops; w.aln_len = a.job_len;
		int rc = kgi_nw_launch(ix, w, max_len, st);
		if (rc != KG_OK) return rc;
	}
	mark(3);
	HIP_TRY(launch_frag_stitch(a, ix->n_cu, st));
	mark(4);
	HIP_TRY(hipMemcpyAsync(ops, a.ops, (size_t)cols, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(aln_len, a.aln_len, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(status, a.status, (size_t)n, hipMemcpyDeviceToHost, st));
	unsigned long long pc[FC_WORDS];
	if (prof) HIP_TRY(hipMemcpyAsync(pc, a.ctl, 8 * FC_WORDS, hipMemcpyDeviceToHost, st));
	mark(5);
	HIP_TRY(hipEventRecord(sc->done, st));
	HIP_TRY(hipEventSynchronize(sc->done));
	if (prof) {
		float ms[5] = {};
		for (int k = 0; k < 5; ++k) (void)hipEventElapsedTime(&ms[k], pe[k], pe[k + 1]);
		for (int k = 0; k < 6; ++k) (void)hipEventDestroy(pe[k]);
		fprintf(stderr, "kg_fragments_batch: the whole call %.2f ms, of it before the first copy %.2f ms | H2D %.2f ms (%.1f MB), partition %.2f ms, nw %.2f ms, stitch %.2f ms, D2H %.2f ms (%.1f MB)\n", 1e3 * (wall() - t_in), 1e3 * (t_q - t_in), ms[0], 1e-6 * (double)(b1 + 28 * n), ms[1], ms[2], ms[3], ms[4], 1e-6 * (double)(cols + 5 * n));
	}
	if (prof)
		fprintf(stderr, "kg_fragments_batch: %lld requests, %llu tasks, %llu pieces, %llu NW jobs | wave cycles per task: load+pack %.0f, diagonal scan %.0f, sort %.0f, normal pairs %.0f, pieces %.0f | runs per task %.1f, columns per task %.0f\n",
		        (long long)n, pc[FC_PROF + 5], pc[FC_PIECES], pc[FC_JOBS], (double)pc[FC_PROF] / std::max(1ull, pc[FC_PROF + 5]), (double)pc[FC_PROF + 1] / std::max(1ull, pc[FC_PROF + 5]),
		        (double)pc[FC_PROF + 2] / std::max(1ull, pc[FC_PROF + 5]), (double)pc[FC_PROF + 3] / std::max(1ull, pc[FC_PROF + 5]), (double)pc[FC_PROF + 4] / std::max(1ull, pc[FC_PROF + 5]),
		        (double)pc[FC_PROF + 6] / std::max(1ull, pc[FC_PROF + 5]), (double)pc[FC_PROF + 7] / std::max(1ull, pc[FC_PROF + 5]));
	return KG_OK;
}
