// abi_stream.hip -- kg_stream_*: FASTQ text in, SAM text out (declared in include/kart_amd.h).
//
// One call chain per batch replaces what the reference's worker loop does for a chunk from GetNextChunk to the fprintf of its SAM
// lines (src/Mapping.cpp:488-637; src/GetData.cpp:109-143): the caller uploads the bytes of the input files as they lie there,
// and receives the text of the records.  A stream owns `lanes` independent lanes -- workspace, device text windows, page-locked
// staging and result buffers, one HIP stream each -- so that several batches are in flight on the device at once (the analogue of
// the reference's N workers each on their own chunk, src/Mapping.cpp:716-717).
#include "abi_internal.hpp"
#include "stream_kernels.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>

#define fail kg_fail

namespace {

struct Lane {
	kg_workspace *ws = nullptr;
	// device
	uint8_t *d_text[2] = {nullptr, nullptr};
	int32_t *d_tile[2] = {nullptr, nullptr};
	uint32_t *d_line_end[2] = {nullptr, nullptr};
	uint32_t *d_rec[2] = {nullptr, nullptr};           // six arrays of rec_capacity words: hdr, name, seq, qual, rlen, qlen
	int64_t *d_meta = nullptr;
	int32_t *d_read_len = nullptr;
	void *d_scan = nullptr;
	size_t scan_bytes = 0;
	int32_t *d_sam_len = nullptr, *d_host_list = nullptr;
	int64_t *d_sam_off = nullptr;
	unsigned long long *d_sam_ctl = nullptr;
	uint8_t *d_sam = nullptr;
	int64_t d_sam_capacity = 0;
	// page-locked
	char *h_text[2] = {nullptr, nullptr};
	int64_t *h_meta = nullptr;                         // [FQM_WORDS] + [24] spare words for totals
	char *h_sam = nullptr;
	int64_t h_sam_capacity = 0;
	int64_t *h_sam_off = nullptr, *h_cand_off = nullptr;
	int32_t *h_host_list = nullptr;
	uint32_t *h_rec_hdr[2] = {nullptr, nullptr};
	kg_aln_record *h_records = nullptr;
	int64_t record_capacity = 0;
	kg_chunk_stats *h_chunk_stats = nullptr;
	int64_t chunk_capacity = 0;
	kg_candidate *h_cands = nullptr;
	kg_seed *h_cand_seeds = nullptr;
	int64_t h_cand_capacity = 0;
	unsigned long long *h_ctl = nullptr;               // [kCtlWords] the search kernel's counters of the batch
	// the batch in the lane
	int64_t mapped_reads = 0;                          // reads of the batch the last kg_stream_map call mapped (still resident: kg_stream_fetch)
	const kg_aln_record *d_records = nullptr;          // ... and where its records lie
	kg_stream_window win{};
	kg_stream_parsed parsed{};
	bool have_batch = false;
	hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	hipEvent_t ev_parsed = nullptr;                    // the batch is materialised and published to its group (recorded on the lane's stream)
};

// ONE seeding launch over the parsed batches of `size` lanes (kg_stream_config::seed_group).  The reference's analogue is simply N
// workers on N chunks (src/Mapping.cpp:716-717); on the device a search launch costs ~0.65 ms + 0.52 ms per M reads -- at 2.5 reads
// per persistent lane a launch lasts as long as its slowest lanes -- so four lanes' 1 M-read batches in one launch cost what 1.6 of
// them cost alone, while chaining, the report and the text keep the lanes' granularity.
// Lanes [first, first + size) share the group: its workspace holds `size` segments of `stride` read slots; lane j materialises its
// reads straight into segment j of the group's character array and publishes offsets / lengths of its slots (group_publish_kernel).
// A ROUND: every lane of the group either arrives with a parsed batch (kg_stream_map) or is absent (kg_stream_group_absent); the
// thread that completes the round launches the seeding of all segments on the group's stream, waits for it and wakes the others;
// each lane then cuts its seeds out of the group's arrays on its own stream.  A lane can only arrive again after it has finished
// with the group's arrays, so the next round never overwrites what a slow lane still reads.
struct SeedGroup {
	int first = 0, size = 0;
	int64_t stride = 0, seg_bases = 0;
	kg_workspace *ws = nullptr;                        // the group's workspace (hits, packed reads, seeds of all segments)
	unsigned long long *h_ctl = nullptr;               // page-locked: the search kernel's counters of the round
	std::mutex mu;
	std::condition_variable cv;
	int64_t round = 0;
	int arrived = 0;
	bool present[kMaxSeedSegments] = {false};
	int absent_rounds[kMaxSeedSegments] = {0};         // 0: the lane takes part; > 0: absent for that many rounds; < 0: until further notice
	int64_t n_reads[kMaxSeedSegments] = {0};
	// results of the last round
	int rc = KG_OK;
	int64_t seed_base[kMaxSeedSegments + 1] = {0};
	bool aborted = false;
	// After the round the lanes run their own stages (chaining .. text) ONE AFTER THE OTHER, in lane order: started all at once they
	// share the device, all finish late and together, and the caller's writer -- which takes the batches in order -- first starves and
	// is then handed four batches at once (measured, 100 M reads: 27.6-28.4 M reads/s against 29.0-29.9 M with independent lanes).
	// A lane keeps the turn until its batch is back on the host (kg_stream_map's last sync).
	int order[kMaxSeedSegments] = {0};
	int n_order = 0, turn = 0;
};

// the turn of lane j of a group at its own stages (see SeedGroup::order); released by hand once, or by the destructor on an error path
struct GroupTurn {
	SeedGroup *sg = nullptr;
	bool held = false;
	bool acquire(SeedGroup &g, int j)
	{
		sg = &g;
		std::unique_lock<std::mutex> lk(g.mu);
		g.cv.wait(lk, [&]() { return g.aborted || (g.turn < g.n_order && g.order[g.turn] == j); });
		held = !g.aborted;
		return held;
	}
	void release()
	{
		if (!held) return;
		held = false;
		std::lock_guard<std::mutex> lk(sg->mu);
		sg->turn++;
		sg->cv.notify_all();
	}
	~GroupTurn() { release(); }
};

}  // namespace

struct kg_stream {
	kg_index *ix = nullptr;
	kg_stream_config cfg{};
	int64_t text_capacity = 0, line_capacity = 0, rec_capacity = 0;
	std::vector<Lane> lanes;
	std::vector<std::unique_ptr<SeedGroup>> groups;    // empty: every lane seeds its own batch
	uint8_t *d_chr_names = nullptr;
	int32_t *d_chr_name_off = nullptr;
	int min_seed_len = 13;
	std::mutex mu;
	kg_stream_timing_t total{};
};

namespace {

void free_lane(Lane &l)
{
	if (l.ws && l.ws->enc_borrowed) l.ws->d_enc = nullptr;        // (a part of its group's array)
	if (l.ev_parsed) (void)hipEventDestroy(l.ev_parsed);
	if (l.ws) kg_workspace_destroy(l.ws);
	for (int f = 0; f < 2; ++f) {
		if (l.d_text[f]) (void)hipFree(l.d_text[f]);
		if (l.d_tile[f]) (void)hipFree(l.d_tile[f]);
		if (l.d_line_end[f]) (void)hipFree(l.d_line_end[f]);
		if (l.d_rec[f]) (void)hipFree(l.d_rec[f]);
		if (l.h_text[f]) (void)hipHostFree(l.h_text[f]);
		if (l.h_rec_hdr[f]) (void)hipHostFree(l.h_rec_hdr[f]);
	}
	for (void *p : {(void *)l.d_meta, (void *)l.d_read_len, l.d_scan, (void *)l.d_sam_len, (void *)l.d_host_list, (void *)l.d_sam_off, (void *)l.d_sam_ctl, (void *)l.d_sam})
		if (p) (void)hipFree(p);
	for (void *p : {(void *)l.h_meta, (void *)l.h_sam, (void *)l.h_sam_off, (void *)l.h_cand_off, (void *)l.h_host_list, (void *)l.h_records, (void *)l.h_chunk_stats,
	                (void *)l.h_cands, (void *)l.h_cand_seeds, (void *)l.h_ctl})
		if (p) (void)hipHostFree(p);
	for (hipEvent_t e : l.ev)
		if (e) (void)hipEventDestroy(e);
	l = Lane();
}

FqWindow window_of(const kg_stream *s, const Lane &l, int f)
{
	FqWindow w;
	w.text = l.d_text[f];
	w.begin = l.win.begin[f]; w.end = l.win.end[f]; w.eof = l.win.eof[f];
	w.tile_lines = l.d_tile[f];
	w.line_end = l.d_line_end[f];
	w.line_capacity = s->line_capacity;
	uint32_t *r = l.d_rec[f];
	const size_t c = (size_t)s->rec_capacity;
	w.rec_hdr = r; w.rec_name = r + c; w.rec_seq = r + 2 * c; w.rec_qual = r + 3 * c;
	w.rec_rlen = (int32_t *)(r + 4 * c); w.rec_qlen = (int32_t *)(r + 5 * c);
	return w;
}

float elapsed(hipEvent_t a, hipEvent_t b)
{
	float ms = 0;
	if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return ms;
}

}  // namespace

extern "C" {

int kg_stream_open(kg_index *ix, const kg_stream_config *cfg, kg_stream **out)
{
	if (!ix || !cfg || !out) return fail(KG_ERR_ARG, "kg_stream_open: null argument");
	*out = nullptr;
	if (cfg->lanes < 1 || cfg->lanes > 16 || cfg->max_reads < 2 || cfg->max_window < 4096 || cfg->max_window > 0xF0000000ll)
		return fail(KG_ERR_ARG, "kg_stream_open: bad configuration (%d lanes, %lld reads, %lld bytes per window)", cfg->lanes, (long long)cfg->max_reads, (long long)cfg->max_window);
	const int group = cfg->seed_group > 1 ? cfg->seed_group : 0;
	if (group && (group > kMaxSeedSegments || cfg->lanes % group != 0))
		return fail(KG_ERR_ARG, "kg_stream_open: seed_group %d must divide the %d lanes and be at most %d", group, cfg->lanes, kMaxSeedSegments);
	if (!ix->d_text) return fail(KG_ERR_ARG, "kg_stream_open: the index holds no text");
	HIP_TRY(hipSetDevice(ix->device));
	std::unique_ptr<kg_stream, void (*)(kg_stream *)> s(new kg_stream(), kg_stream_close);
	s->ix = ix;
	s->cfg = *cfg;
	s->text_capacity = (cfg->max_window + 4096 + 255) & ~255ll;
	// a line of FASTQ text is rarely shorter than 16 bytes on average (header, sequence, '+', qualities); a window with more lines
	// than this goes back to the caller's reader (FQ_STOP_IRREGULAR)
	s->line_capacity = cfg->max_window / 16 + 4096;
	s->rec_capacity = s->line_capacity / 4 + 1;
	{
		kg_index_info_t info;
		kg_index_info(ix, &info);
		s->min_seed_len = info.min_seed_len;
		std::vector<int32_t> off{0};
		std::string names;
		for (const ContigRec &c : ix->contigs) { names += c.name; off.push_back((int32_t)names.size()); }
		HIP_TRY(hipMalloc((void **)&s->d_chr_names, names.size() + 16));
		HIP_TRY(hipMalloc((void **)&s->d_chr_name_off, 4 * off.size()));
		HIP_TRY(hipMemcpy(s->d_chr_names, names.data(), names.size(), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(s->d_chr_name_off, off.data(), 4 * off.size(), hipMemcpyHostToDevice));
	}
	s->lanes.resize((size_t)cfg->lanes);
	const int64_t n = cfg->max_reads;
	const int64_t seg_bases = (cfg->max_window + 4096 + 64 + 255) & ~255ll;      // a lane's part of its group's character array
	for (int g = 0; group && g < cfg->lanes / group; ++g) {
		// (in the stream's list at once: whatever fails below, kg_stream_close frees what the group holds -- ADVICE r4: a group that failed half-way
		//  used to keep its multi-GB arrays, and the host path the caller falls back to then ran out of device memory)
		s->groups.push_back(std::unique_ptr<SeedGroup>(new SeedGroup()));
		SeedGroup *sg = s->groups.back().get();
		sg->first = g * group; sg->size = group;
		sg->stride = n + 8; sg->seg_bases = seg_bases;
		int rc = kg_workspace_create(ix, sg->stride * group, seg_bases * group, &sg->ws);
		if (rc != KG_OK) return rc;
		kg_workspace *gw = sg->ws;
		HIP_TRY(hipMalloc((void **)&gw->d_enc, (size_t)gw->max_bases + 64));
		HIP_TRY(hipMalloc((void **)&gw->d_read_off, 8 * (size_t)(gw->max_reads + 1)));
		HIP_TRY(hipMalloc((void **)&gw->d_seed_off, 8 * (size_t)(gw->max_reads + 1)));
		HIP_TRY(hipMalloc((void **)&gw->group_read_len, 4 * (size_t)(gw->max_reads + 1)));
		HIP_TRY(hipMemsetAsync(gw->d_enc, 'N', (size_t)gw->max_bases + 64, gw->stream));
		HIP_TRY(hipMemsetAsync(gw->d_read_off, 0, 8 * (size_t)(gw->max_reads + 1), gw->stream));
		HIP_TRY(hipMemsetAsync(gw->group_read_len, 0, 4 * (size_t)(gw->max_reads + 1), gw->stream));
		HIP_TRY(hipStreamSynchronize(gw->stream));          // (the lanes write these arrays from their own streams)
		HIP_TRY(hipHostMalloc((void **)&sg->h_ctl, 8 * kCtlWords, hipHostMallocDefault));
		(void)kg_workspace_set_profiling(gw, 1);
	}
	for (size_t li = 0; li < s->lanes.size(); ++li) {
		Lane &l = s->lanes[li];
		// bases of a batch: what two windows can hold (every read costs its characters twice plus a header)
		int rc = kg_workspace_create(ix, n + 8, cfg->max_window + 4096, &l.ws);
		if (rc != KG_OK) return rc;
		kg_workspace *ws = l.ws;
		if (group) {
			SeedGroup &sg = *s->groups[li / (size_t)group];
			ws->d_enc = sg.ws->d_enc + (int64_t)(li % (size_t)group) * sg.seg_bases;
			ws->enc_borrowed = true;
			HIP_TRY(hipEventCreateWithFlags(&l.ev_parsed, hipEventDisableTiming));
		} else
		HIP_TRY(hipMalloc((void **)&ws->d_enc, (size_t)ws->max_bases + 64));
		HIP_TRY(hipMalloc((void **)&ws->d_read_off, 8 * (size_t)(ws->max_reads + 1)));
		HIP_TRY(hipMalloc((void **)&ws->d_seed_off, 8 * (size_t)(ws->max_reads + 1)));
		(void)kg_workspace_set_profiling(ws, 1);
		const int64_t n_tiles = s->text_capacity / kFqTile + 2;
		for (int f = 0; f < 2; ++f) {
			HIP_TRY(hipMalloc((void **)&l.d_text[f], (size_t)s->text_capacity));
			HIP_TRY(hipMalloc((void **)&l.d_tile[f], 4 * (size_t)(n_tiles + 1)));
			HIP_TRY(hipMalloc((void **)&l.d_line_end[f], 4 * (size_t)s->line_capacity));
			HIP_TRY(hipMalloc((void **)&l.d_rec[f], 4 * 6 * (size_t)s->rec_capacity));
			HIP_TRY(hipHostMalloc((void **)&l.h_text[f], (size_t)s->text_capacity, hipHostMallocDefault));
			HIP_TRY(hipHostMalloc((void **)&l.h_rec_hdr[f], 4 * (size_t)(n + 8), hipHostMallocDefault));
		}
		HIP_TRY(hipMalloc((void **)&l.d_meta, 8 * FQM_WORDS));
		HIP_TRY(hipMalloc((void **)&l.d_read_len, 4 * (size_t)(n + 8)));
		l.scan_bytes = fq_scan_temp_bytes(std::max<int64_t>(n + 8, n_tiles + 1));
		HIP_TRY(hipMalloc(&l.d_scan, l.scan_bytes ? l.scan_bytes : 256));
		HIP_TRY(hipMalloc((void **)&l.d_sam_len, 4 * (size_t)(n + 8)));
		HIP_TRY(hipMalloc((void **)&l.d_host_list, 4 * (size_t)(n + 8)));
		HIP_TRY(hipMalloc((void **)&l.d_sam_off, 8 * (size_t)(n + 8)));
		HIP_TRY(hipMalloc((void **)&l.d_sam_ctl, 8 * 4));
		HIP_TRY(hipHostMalloc((void **)&l.h_meta, 8 * (FQM_WORDS + 24), hipHostMallocDefault));
		HIP_TRY(hipHostMalloc((void **)&l.h_sam_off, 8 * (size_t)(n + 8), hipHostMallocDefault));
		HIP_TRY(hipHostMalloc((void **)&l.h_cand_off, 8 * (size_t)(n + 8), hipHostMallocDefault));
		HIP_TRY(hipHostMalloc((void **)&l.h_host_list, 4 * (size_t)(n + 8), hipHostMallocDefault));
		HIP_TRY(hipHostMalloc((void **)&l.h_ctl, 8 * kCtlWords, hipHostMallocDefault));
		// the text of a batch: its input's bytes and a little more per read (FLAG .. TLEN and the tags against '+' and the mate suffix)
		l.d_sam_capacity = 2 * cfg->max_window + 64 * n + 4096;
		HIP_TRY(hipMalloc((void **)&l.d_sam, (size_t)l.d_sam_capacity));
		l.h_sam_capacity = l.d_sam_capacity;
		HIP_TRY(hipHostMalloc((void **)&l.h_sam, (size_t)l.h_sam_capacity, hipHostMallocDefault));
		l.record_capacity = n + n / 4 + 4096;
		HIP_TRY(hipHostMalloc((void **)&l.h_records, sizeof(kg_aln_record) * (size_t)l.record_capacity, hipHostMallocDefault));
		for (hipEvent_t &e : l.ev) HIP_TRY(hipEventCreate(&e));
	}
	HIP_TRY(hipDeviceSynchronize());                     // everything above has landed before a lane's first batch
	*out = s.release();
	return KG_OK;
}

void kg_stream_close(kg_stream *s)
{
	if (!s) return;
	(void)hipSetDevice(s->ix->device);
	for (Lane &l : s->lanes) free_lane(l);
	for (std::unique_ptr<SeedGroup> &sg : s->groups) {
		if (sg->ws && sg->ws->group_read_len) { (void)hipFree(sg->ws->group_read_len); sg->ws->group_read_len = nullptr; }
		if (sg->ws) kg_workspace_destroy(sg->ws);
		if (sg->h_ctl) (void)hipHostFree(sg->h_ctl);
	}
	if (s->d_chr_names) (void)hipFree(s->d_chr_names);
	if (s->d_chr_name_off) (void)hipFree(s->d_chr_name_off);
	delete s;
}

char *kg_stream_staging(kg_stream *s, int lane, int file, int64_t *capacity)
{
	if (!s || lane < 0 || lane >= (int)s->lanes.size() || file < 0 || file > 1) return nullptr;
	if (capacity) *capacity = s->cfg.max_window;
	return s->lanes[(size_t)lane].h_text[file];
}

int kg_stream_upload(kg_stream *s, int lane, int file, int64_t from, int64_t to)
{
	if (!s || lane < 0 || lane >= (int)s->lanes.size() || file < 0 || file > 1) return fail(KG_ERR_ARG, "kg_stream_upload: bad argument");
	if (from < 0 || to < from || to > s->cfg.max_window) return fail(KG_ERR_ARG, "kg_stream_upload: range [%lld, %lld) outside the staging buffer", (long long)from, (long long)to);
	if (to == from) return KG_OK;
	Lane &l = s->lanes[(size_t)lane];
	HIP_TRY(hipSetDevice(s->ix->device));
	HIP_TRY(hipMemcpyAsync(l.d_text[file] + from, l.h_text[file] + from, (size_t)(to - from), hipMemcpyHostToDevice, l.ws->stream));
	return KG_OK;
}

int kg_stream_parse(kg_stream *s, int lane, const kg_stream_window *w, kg_stream_parsed *out)
{
	if (!s || !w || !out || lane < 0 || lane >= (int)s->lanes.size()) return fail(KG_ERR_ARG, "kg_stream_parse: bad argument");
	Lane &l = s->lanes[(size_t)lane];
	const int nf = w->two_files ? 2 : 1;
	for (int f = 0; f < nf; ++f)
		if (w->begin[f] < 0 || w->end[f] < w->begin[f] || w->end[f] > s->cfg.max_window) return fail(KG_ERR_ARG, "kg_stream_parse: window %d outside the staging buffer", f);
	if (w->chunk_reads < 2 || (w->chunk_reads & 1) || w->want_reads < w->chunk_reads || w->want_reads > s->cfg.max_reads || w->want_reads % w->chunk_reads)
		return fail(KG_ERR_ARG, "kg_stream_parse: want_reads must be a multiple of chunk_reads (an even number) within the stream's batch size");
	HIP_TRY(hipSetDevice(s->ix->device));
	l.win = *w;
	l.have_batch = false;
	kg_workspace *ws = l.ws;
	FqArgs a;
	a.w[0] = window_of(s, l, 0);
	a.w[1] = window_of(s, l, 1);
	if (!w->two_files) { a.w[1].begin = a.w[1].end = 0; a.w[1].eof = 1; }
	a.two_files = w->two_files ? 1 : 0;
	a.paired = w->paired ? 1 : 0;
	a.chunk_reads = w->chunk_reads;
	a.gz_lines = w->gz_lines ? 1 : 0;
	a.max_reads = s->cfg.max_reads;
	a.want_reads = w->want_reads;
	a.meta = l.d_meta;
	a.read_len = l.d_read_len;
	a.read_off = ws->d_read_off;
	a.enc = ws->d_enc;
	a.n_reads = 0;
	hipStream_t st = ws->stream;
	KtUse kt_use(ws->profiling ? &ws->kt : nullptr);
	HIP_TRY(hipEventRecord(l.ev[0], st));
	HIP_TRY(launch_fq_parse(a, l.d_scan, l.scan_bytes, s->ix->n_cu, st));
	HIP_TRY(hipMemcpyAsync(l.h_meta, l.d_meta, 8 * FQM_WORDS, hipMemcpyDeviceToHost, st));
	HIP_TRY(kgi_sync(ws));
	kg_stream_parsed &p = l.parsed;
	p.n_reads = l.h_meta[FQM_READS]; p.n_chunks = l.h_meta[FQM_CHUNKS]; p.n_bases = l.h_meta[FQM_BASES];
	p.used[0] = l.h_meta[FQM_USED0]; p.used[1] = w->two_files ? l.h_meta[FQM_USED1] : 0;
	p.stop = (int32_t)l.h_meta[FQM_STOP]; p.done = (int32_t)l.h_meta[FQM_DONE];
	if (p.n_bases > ws->max_bases) return fail(KG_ERR_CAPACITY, "kg_stream_parse: %lld bases exceed the lane's workspace (%lld)", (long long)p.n_bases, (long long)ws->max_bases);
	if (p.n_reads > 0) {
		a.n_reads = p.n_reads;
		HIP_TRY(launch_fq_materialise(a, s->ix->n_cu, st));
		// where every record of the batch starts in its window: the caller re-reads the few reads that come back as KG_ALN_HOST
		const int64_t r0 = w->two_files ? p.n_reads / 2 : p.n_reads;
		HIP_TRY(hipMemcpyAsync(l.h_rec_hdr[0], a.w[0].rec_hdr, 4 * (size_t)r0, hipMemcpyDeviceToHost, st));
		if (w->two_files) HIP_TRY(hipMemcpyAsync(l.h_rec_hdr[1], a.w[1].rec_hdr, 4 * (size_t)r0, hipMemcpyDeviceToHost, st));
	}
	if (!s->groups.empty() && p.n_reads > 0) {
		// the batch as a segment of its group's batch: every slot's offset in the group's character array and its length
		SeedGroup &sg = *s->groups[(size_t)lane / (size_t)s->cfg.seed_group];
		const int j = lane - sg.first;
		HIP_TRY(launch_group_publish(ws->d_read_off, p.n_reads, sg.stride, (int64_t)j * sg.seg_bases, sg.ws->d_read_off + (int64_t)j * sg.stride,
		                             sg.ws->group_read_len + (int64_t)j * sg.stride, s->ix->n_cu, st));
		HIP_TRY(hipEventRecord(l.ev_parsed, st));
	}
	HIP_TRY(hipEventRecord(l.ev[1], st));
	l.have_batch = p.n_reads > 0;
	*out = p;
	return KG_OK;
}

// the lane's part in a round of its group (see SeedGroup): returns when the round's seeding is done and the lane's seeds lie in its
// own workspace; *was_leader: this call launched the round (its search kernel counts once in the statistics)
static int group_seed(kg_stream *s, int lane, int64_t n, int64_t n_bases, int64_t *n_seeds, bool *was_leader, float *search_ms, double *useful_bytes)
{
	SeedGroup &sg = *s->groups[(size_t)lane / (size_t)s->cfg.seed_group];
	Lane &l = s->lanes[(size_t)lane];
	const int j = lane - sg.first;
	*was_leader = false;
	std::unique_lock<std::mutex> lk(sg.mu);
	if (sg.aborted) return fail(KG_ERR_ARG, "kg_stream_map: the stream's seeding groups were aborted");
	if (sg.absent_rounds[j] < 0) return fail(KG_ERR_ARG, "kg_stream_map: lane %d was declared absent until further notice (kg_stream_group_absent) and arrives with a batch", lane);
	// (a lane that sits out a round and is back with a batch before that round has run waits for it)
	if (!sg.cv.wait_for(lk, std::chrono::seconds(600), [&]() { return sg.absent_rounds[j] == 0 || sg.aborted; })) {
		sg.aborted = true;
		sg.cv.notify_all();
		return fail(KG_ERR_ARG, "kg_stream_map: lane %d waited 600 s for the round it declared itself absent from", lane);
	}
	if (sg.aborted) return fail(KG_ERR_ARG, "kg_stream_map: the stream's seeding groups were aborted");
	const int64_t my_round = sg.round;
	sg.present[j] = true;
	sg.n_reads[j] = n;
	sg.arrived++;
	auto complete = [&]() {
		int accounted = sg.arrived;
		for (int i = 0; i < sg.size; ++i) accounted += (!sg.present[i] && sg.absent_rounds[i] != 0) ? 1 : 0;
		return accounted >= sg.size;
	};
	auto run_round = [&]() {           // (sg.mu held by the caller; every other lane of the group waits or is absent)
		kg_workspace *gw = sg.ws;
		int64_t counts[kMaxSeedSegments];
		int rc = KG_OK;
		for (int i = 0; i < sg.size && rc == KG_OK; ++i) {
			counts[i] = sg.present[i] ? sg.n_reads[i] : 0;
			// every lane publishes its slots ON ITS OWN STREAM -- a batch in kg_stream_parse, an absence (all slots empty) in kg_stream_group_absent --
			// and records ev_parsed behind it: the round waits for all of them.  (Round 4 zeroed an absent lane's slots here, on the group's
			// stream: nothing ordered that against the lane's publish of its NEXT batch, which it may parse before this round has run --
			// the memset could then wipe the lengths and the whole batch came out unmapped.)
			if (hipStreamWaitEvent(gw->stream, s->lanes[(size_t)(sg.first + i)].ev_parsed, 0) != hipSuccess) rc = fail(KG_ERR_NO_DEVICE, "kg_stream_map: hipStreamWaitEvent");
		}
		if (rc == KG_OK) rc = kgi_seed_group(gw, KG_MODE_FAST | KG_INPUT_ASCII, s->min_seed_len, KG_OCC_THR_DEFAULT, sg.size, sg.stride, counts, sg.seed_base);
		if (rc == KG_OK && (hipMemcpyAsync(sg.h_ctl, gw->d_ctl, 8 * kCtlWords, hipMemcpyDeviceToHost, gw->stream) != hipSuccess || kgi_sync(gw) != hipSuccess))
			rc = fail(KG_ERR_NO_DEVICE, "kg_stream_map: reading the group's counters back");
		sg.rc = rc;
		sg.n_order = 0; sg.turn = 0;
		for (int i = 0; i < sg.size; ++i)
			if (sg.present[i]) sg.order[sg.n_order++] = i;
		for (int i = 0; i < sg.size; ++i) {
			sg.present[i] = false;
			if (sg.absent_rounds[i] > 0) sg.absent_rounds[i]--;
		}
		sg.arrived = 0;
		sg.round++;
		sg.cv.notify_all();
	};
	// whoever finds the round complete runs it: the lane that arrives last, or -- when an absence completed it -- the first waiting
	// lane to look (kg_stream_group_absent wakes them).  The wait is bounded: a caller that lets a lane neither arrive nor declare
	// itself absent would hang the others for ever.
	const std::chrono::steady_clock::time_point give_up = std::chrono::steady_clock::now() + std::chrono::seconds(600);
	auto leave = [&]() {                // (an aborted round: this lane is no longer part of it)
		if (sg.round == my_round && sg.present[j]) { sg.present[j] = false; sg.arrived--; }
	};
	for (;;) {
		if (sg.aborted) { leave(); return fail(KG_ERR_ARG, "kg_stream_map: the stream's seeding groups were aborted"); }
		if (sg.round != my_round) break;
		if (complete()) {
			run_round();
			*was_leader = true;
			if (sg.rc == KG_OK) {
				kg_workspace *gw = sg.ws;
				*search_ms = gw->profiling && gw->ev[0] ? elapsed(gw->ev[0], gw->ev[1]) : 0;
				const unsigned long long *c = sg.h_ctl;
				const double sa_bytes = s->ix->view.fsa32 ? 4.0 : 8.0;
				const int64_t reads = gw->group_prefix[sg.size];
				*useful_bytes = 8.0 * (double)c[17] + 32.0 * (double)c[18] + (double)c[25] + sa_bytes * (double)c[8] + 48.0 * (double)c[20] + 8.0 * (double)c[21] + 20.0 * (double)reads + 32.0 * (double)c[1];
			}
			break;
		}
		if (sg.cv.wait_until(lk, give_up) == std::cv_status::timeout) {
			sg.aborted = true;
			sg.cv.notify_all();
			leave();
			return fail(KG_ERR_ARG, "kg_stream_map: lane %d waited 600 s for the other lanes of its seeding group (a lane without a batch must call kg_stream_group_absent)", lane);
		}
	}
	if (sg.rc != KG_OK) return sg.rc;
	const int64_t first = sg.seed_base[j], count = (j + 1 < sg.size ? sg.seed_base[j + 1] : sg.seed_base[sg.size]) - first;
	lk.unlock();
	// ---- the lane's slice: offsets rebased, seeds copied into its own workspace (the group's arrays belong to the next round) ----
	kg_workspace *ws = l.ws, *gw = sg.ws;
	if (count > ws->seed_capacity) {
		if (ws->d_seeds) HIP_TRY(hipFree(ws->d_seeds));
		ws->d_seeds = nullptr;
		const int64_t want = std::max<int64_t>(count + count / 4, 8 * ws->max_reads + 1024);
		HIP_TRY(hipMalloc((void **)&ws->d_seeds, sizeof(kg_seed) * (size_t)want));
		ws->seed_capacity = want;
	}
	HIP_TRY(launch_group_rebase(gw->d_seed_off + (int64_t)j * sg.stride, n, first, ws->d_seed_off, s->ix->n_cu, ws->stream));
	if (count > 0) HIP_TRY(hipMemcpyAsync(ws->d_seeds, gw->d_seeds + first, sizeof(kg_seed) * (size_t)count, hipMemcpyDeviceToDevice, ws->stream));
	ws->last_reads = n; ws->last_seeds = count; ws->last_cands = -1; ws->last_ascii = true;
	(void)n_bases;
	*n_seeds = count;
	return KG_OK;
}

int kg_stream_map(kg_stream *s, int lane, const kg_stream_params *prm, kg_stream_result *out)
{
	if (!s || !prm || !out || lane < 0 || lane >= (int)s->lanes.size()) return fail(KG_ERR_ARG, "kg_stream_map: bad argument");
	Lane &l = s->lanes[(size_t)lane];
	if (!l.have_batch) return fail(KG_ERR_ARG, "kg_stream_map: no parsed batch in lane %d (kg_stream_parse first)", lane);
	memset(out, 0, sizeof(*out));
	kg_index *ix = s->ix;
	kg_workspace *ws = l.ws;
	hipStream_t st = ws->stream;
	HIP_TRY(hipSetDevice(ix->device));
	KtUse kt_use(ws->profiling ? &ws->kt : nullptr);
	const int64_t n = l.parsed.n_reads;
	const int n_chunks = (int)l.parsed.n_chunks;
	// ---- seeding (IdentifySeedPairs_FastMode) and chaining on the resident characters ---------------------------------------
	int64_t n_seeds = 0, totals[2] = {0, 0};
	const bool grouped = !s->groups.empty();
	bool group_leader = false;
	float group_search_ms = 0;
	double group_useful = 0;
	int rc;
	GroupTurn turn;
	if (grouped) {
		rc = group_seed(s, lane, n, l.parsed.n_bases, &n_seeds, &group_leader, &group_search_ms, &group_useful);
		static const bool no_turns = getenv("KG_GROUP_NO_TURNS") != nullptr;          // A/B aid: the lanes of a group start their stages at once
		SeedGroup &sg = *s->groups[(size_t)lane / (size_t)s->cfg.seed_group];
		if (rc == KG_OK && !no_turns && !turn.acquire(sg, lane - sg.first)) rc = fail(KG_ERR_ARG, "kg_stream_map: the stream's seeding groups were aborted");
		if (rc != KG_OK) {                  // (this lane will never take its turn: nobody may wait for it)
			std::lock_guard<std::mutex> lk(sg.mu);
			sg.aborted = true;
			sg.cv.notify_all();
		}
	} else rc = kgi_seed_resident(ws, KG_MODE_FAST | KG_INPUT_ASCII, s->min_seed_len, KG_OCC_THR_DEFAULT, n, l.parsed.n_bases, &n_seeds);
	if (rc != KG_OK) return rc;
	if (!grouped) HIP_TRY(hipMemcpyAsync(l.h_ctl, ws->d_ctl, 8 * kCtlWords, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipEventRecord(l.ev[2], st));
	rc = kgi_chain_resident(ws, 0, prm->max_gaps, totals);
	if (rc != KG_OK) return rc;
	HIP_TRY(hipEventRecord(l.ev[3], st));
	// ---- the per-read report ------------------------------------------------------------------------------------------------
	std::vector<int64_t> chunk_off((size_t)n_chunks + 1);
	std::vector<uint8_t> chunk_paired((size_t)n_chunks, (uint8_t)(l.win.paired ? 1 : 0));
	for (int c = 0; c <= n_chunks; ++c) chunk_off[(size_t)c] = std::min<int64_t>(n, (int64_t)c * l.win.chunk_reads);
	if (n_chunks > l.chunk_capacity) {
		if (l.h_chunk_stats) HIP_TRY(hipHostFree(l.h_chunk_stats));
		l.h_chunk_stats = nullptr;
		l.chunk_capacity = n_chunks + n_chunks / 2 + 64;
		HIP_TRY(hipHostMalloc((void **)&l.h_chunk_stats, sizeof(kg_chunk_stats) * (size_t)l.chunk_capacity, hipHostMallocDefault));
	}
	AlnArgs a;
	rc = kgi_align_resident(ws, chunk_off.data(), chunk_paired.data(), n_chunks, prm->est_distance, prm->max_insert, prm->max_gaps, prm->multi_hit,
	                        prm->unset_flag, l.record_capacity, a);
	if (rc != KG_OK) return rc;
	HIP_TRY(hipEventRecord(l.ev[4], st));
	// ---- the text -----------------------------------------------------------------------------------------------------------
	SamArgs q;
	q.w[0] = window_of(s, l, 0); q.w[1] = window_of(s, l, 1);
	q.two_files = l.win.two_files ? 1 : 0; q.paired = l.win.paired ? 1 : 0;
	q.enc = ws->d_enc; q.read_off = ws->d_read_off; q.n_reads = n;
	q.records = a.records;
	q.chr_names = s->d_chr_names; q.chr_name_off = s->d_chr_name_off;
	q.sam_len = l.d_sam_len; q.sam_off = l.d_sam_off; q.sam = l.d_sam; q.sam_capacity = l.d_sam_capacity;
	q.host_list = l.d_host_list; q.ctl = l.d_sam_ctl;
	HIP_TRY(launch_sam_size(q, l.d_scan, l.scan_bytes, ix->n_cu, st));
	int64_t *tot = l.h_meta + FQM_WORDS;              // [0] text bytes, [1] reads handed back, [2] format errors, [3] extra records of -m
	HIP_TRY(hipMemcpyAsync(&tot[0], l.d_sam_off + n, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&tot[1], l.d_sam_ctl, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&tot[3], ws->d_aln_ctl + 7, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(kgi_sync(ws));
	const int64_t sam_bytes = tot[0], n_host = tot[1];
	const int64_t extra = prm->multi_hit ? std::min<int64_t>(tot[3], a.extra_capacity) : 0;
	if (sam_bytes > l.d_sam_capacity) {
		// (never seen: the estimate in kg_stream_open is generous; -m on a repeat-rich genome could get here)
		HIP_TRY(hipFree(l.d_sam)); l.d_sam = nullptr;
		l.d_sam_capacity = sam_bytes + sam_bytes / 4;
		HIP_TRY(hipMalloc((void **)&l.d_sam, (size_t)l.d_sam_capacity));
		q.sam = l.d_sam; q.sam_capacity = l.d_sam_capacity;
	}
	if (sam_bytes > l.h_sam_capacity) {
		HIP_TRY(hipHostFree(l.h_sam)); l.h_sam = nullptr;
		l.h_sam_capacity = sam_bytes + sam_bytes / 4;
		HIP_TRY(hipHostMalloc((void **)&l.h_sam, (size_t)l.h_sam_capacity, hipHostMallocDefault));
	}
	const int64_t cand_need = std::max(totals[0], totals[1]);
	if (cand_need > l.h_cand_capacity) {
		if (l.h_cands) HIP_TRY(hipHostFree(l.h_cands));
		if (l.h_cand_seeds) HIP_TRY(hipHostFree(l.h_cand_seeds));
		l.h_cands = nullptr; l.h_cand_seeds = nullptr;
		l.h_cand_capacity = std::max<int64_t>(cand_need + cand_need / 4 + 1024, 3 * s->cfg.max_reads);
		HIP_TRY(hipHostMalloc((void **)&l.h_cands, sizeof(kg_candidate) * (size_t)l.h_cand_capacity, hipHostMallocDefault));
		HIP_TRY(hipHostMalloc((void **)&l.h_cand_seeds, sizeof(kg_seed) * (size_t)l.h_cand_capacity, hipHostMallocDefault));
	}
	HIP_TRY(launch_sam_format(q, ix->n_cu, st));
	// measurement aid (bench.py's gpu_pipeline leg): the text summed on the device, for runs that never copy it into file pages (read per call: a session switches it on and off)
	const bool checksum = getenv("KG_STREAM_CHECKSUM") != nullptr;
	if (checksum) HIP_TRY(launch_sam_checksum(q, ix->n_cu, st));
	HIP_TRY(hipEventRecord(l.ev[5], st));
	// (grouped seeding) the turn is kept until the batch is back on the host.  Passing it on here -- so that this lane's copies overlap the
	// next lane's kernels, KG_GROUP_TURN_EARLY=1 -- gives the same FASTQ -> SAM rate at 100 M reads (29.4-29.9 M against 29.8-30.4 M
	// mapped reads/s) while the next lane's first kernel then waits behind this one's copies: chain_kernel 5.0 ms per launch instead of 0.35,
	// the stage sums twice as long (profiles/r04u_trace_chain.log, r04v_ab_turn_late.log); the other group's lanes overlap either way.
	static const bool turn_early = getenv("KG_GROUP_TURN_EARLY") != nullptr;
	if (turn_early) turn.release();
	if (sam_bytes > 0) HIP_TRY(hipMemcpyAsync(l.h_sam, l.d_sam, (size_t)sam_bytes, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(l.h_sam_off, l.d_sam_off, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
	// the records, candidate offsets, candidates and their seeds stay on the device unless the caller wants them all: it asks for the chunks it needs
	// (kg_stream_fetch) -- 207 of the 605 bytes per read that used to cross the link
	const bool all = prm->fetch_all != 0;
	l.mapped_reads = n; l.d_records = a.records;
	if (all) HIP_TRY(hipMemcpyAsync(l.h_records, a.records, sizeof(kg_aln_record) * (size_t)(n + extra), hipMemcpyDeviceToHost, st));
	else if (extra > 0) HIP_TRY(hipMemcpyAsync(l.h_records + n, a.records + n, sizeof(kg_aln_record) * (size_t)extra, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(l.h_chunk_stats, ws->d_chunk_stats, sizeof(kg_chunk_stats) * (size_t)n_chunks, hipMemcpyDeviceToHost, st));
	if (all) HIP_TRY(hipMemcpyAsync(l.h_cand_off, ws->d_cand_off, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
	// the candidates travel with every batch: the reads handed back need them, and so does a pair the caller maps again because its
	// speculated EstDistance did not hold
	if (n_host > 0) HIP_TRY(hipMemcpyAsync(l.h_host_list, l.d_host_list, 4 * (size_t)n_host, hipMemcpyDeviceToHost, st));
	if (all && totals[0] > 0) HIP_TRY(hipMemcpyAsync(l.h_cands, ws->d_dense_cands, sizeof(kg_candidate) * (size_t)totals[0], hipMemcpyDeviceToHost, st));
	if (all && totals[1] > 0) HIP_TRY(hipMemcpyAsync(l.h_cand_seeds, ws->d_dense_seeds, sizeof(kg_seed) * (size_t)totals[1], hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&tot[2], l.d_sam_ctl + 1, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&tot[8], ws->d_aln_ctl, 8 * 7, hipMemcpyDeviceToHost, st));          // the alignment stage's list sizes of this batch ...
	HIP_TRY(hipMemcpyAsync(&tot[15], ws->d_aln_ctl + 32, 8, hipMemcpyDeviceToHost, st));        // ... and the candidates its fast plan kernel left to the general one
	tot[16] = tot[17] = 0;
	if (checksum) HIP_TRY(hipMemcpyAsync(&tot[16], l.d_sam_ctl + 2, 16, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&tot[14], ws->d_aln_ctl + 36, 8, hipMemcpyDeviceToHost, st));        // ... and the pairs aln_trivial_kernel decided (in the place of the piece count, which nobody reads)
	HIP_TRY(hipEventRecord(l.ev[6], st));
	HIP_TRY(kgi_sync(ws));
	if (tot[2] != 0) return fail(KG_ERR_NO_DEVICE, "kg_stream_map: %lld records were formatted to a size other than the one announced", (long long)tot[2]);
	std::sort(l.h_host_list, l.h_host_list + n_host);
	out->n_reads = n; out->n_chunks = n_chunks;
	out->sam = l.h_sam; out->sam_bytes = sam_bytes; out->sam_off = l.h_sam_off;
	out->records = l.h_records; out->n_records = n + extra;
	out->chunk_stats = l.h_chunk_stats;
	out->host_reads = l.h_host_list; out->n_host_reads = n_host;
	out->cand_off = l.h_cand_off; out->cands = l.h_cands; out->cand_seeds = l.h_cand_seeds;
	out->rec_start[0] = l.h_rec_hdr[0]; out->rec_start[1] = l.win.two_files ? l.h_rec_hdr[1] : nullptr;
	l.have_batch = false;
	{
		// what the batch cost on the device, stage by stage, and what its search kernel fetched (kg_workspace_traffic's formula)
		float sk = 0;
		if (!grouped && ws->profiling && ws->ev[0]) sk = elapsed(ws->ev[0], ws->ev[1]);
		const unsigned long long *c = l.h_ctl;
		const double sa_bytes = ix->view.fsa32 ? 4.0 : 8.0;
		double useful = grouped ? 0.0 : 8.0 * (double)c[17] + 32.0 * (double)c[18] + (double)c[25] + sa_bytes * (double)c[8] + 48.0 * (double)c[20] + 8.0 * (double)c[21] + 20.0 * (double)n + 32.0 * (double)c[1];
		// (a group's launch counts once, with the lane that launched it)
		if (grouped && group_leader) { sk = group_search_ms; useful = group_useful; }
		std::lock_guard<std::mutex> lk(s->mu);
		kg_stream_timing_t &t = s->total;
		t.batches += 1; t.reads += n;
		t.parse_ms += elapsed(l.ev[0], l.ev[1]); t.seed_ms += elapsed(l.ev[1], l.ev[2]); t.chain_ms += elapsed(l.ev[2], l.ev[3]);
		t.align_ms += elapsed(l.ev[3], l.ev[4]); t.format_ms += elapsed(l.ev[4], l.ev[5]); t.copy_ms += elapsed(l.ev[5], l.ev[6]);
		t.search_kernel_ms += sk; t.search_kernel_launches += (!grouped || group_leader) ? 1 : 0; t.search_useful_bytes += useful;
		t.text_in_bytes += (double)((l.parsed.used[0] - l.win.begin[0]) + (l.win.two_files ? l.parsed.used[1] - l.win.begin[1] : 0));
		t.text_out_bytes += (double)sam_bytes;
		t.candidates += (double)totals[0]; t.candidate_seeds += (double)totals[1];
		for (int i = 0; i < 8; ++i) t.aln_counts[i] += (double)tot[8 + i];
		t.text_checksum[0] += (double)(uint64_t)tot[16]; t.text_checksum[1] += (double)(uint64_t)tot[17];
		// the kernels' own launches of this batch (events around each, the lane's stream is synchronised)
		for (int i = 0; i < KT_SLOTS; ++i) {
			const int used = ws->kt.used[i];
			if (!used) continue;
			ws->kt.used[i] = 0;
			for (int j = 0; j < used; ++j) t.kernel_ms[i] += elapsed(ws->kt.b[i][j], ws->kt.e[i][j]);
			t.kernel_launches[i] += used;
		}
	}
	return KG_OK;
}

int kg_stream_group_absent(kg_stream *s, int lane, int rounds)
{
	if (!s || lane < 0 || lane >= (int)s->lanes.size()) return fail(KG_ERR_ARG, "kg_stream_group_absent: bad argument");
	if (s->groups.empty()) return KG_OK;
	SeedGroup &sg = *s->groups[(size_t)lane / (size_t)s->cfg.seed_group];
	const int j = lane - sg.first;
	std::unique_lock<std::mutex> lk(sg.mu);
	if (sg.present[j]) return fail(KG_ERR_ARG, "kg_stream_group_absent: lane %d is inside a round", lane);
	if (rounds != 0) {
		// the lane's slots of the group's batch are empty from here on (until its next kg_stream_parse publishes a batch): on the lane's own stream
		Lane &l = s->lanes[(size_t)lane];
		HIP_TRY(hipSetDevice(s->ix->device));
		HIP_TRY(launch_group_publish(l.ws->d_read_off, 0, sg.stride, (int64_t)j * sg.seg_bases, sg.ws->d_read_off + (int64_t)j * sg.stride, sg.ws->group_read_len + (int64_t)j * sg.stride,
		                             s->ix->n_cu, l.ws->stream));
		HIP_TRY(hipEventRecord(l.ev_parsed, l.ws->stream));
	}
	sg.absent_rounds[j] = rounds;
	if (rounds == 0) sg.aborted = false;               // (a new run: the lanes take part again; an aborted run's half-finished round is forgotten)
	// the round may be complete now: everybody else has arrived and waits
	int accounted = sg.arrived;
	for (int i = 0; i < sg.size; ++i) accounted += (!sg.present[i] && sg.absent_rounds[i] != 0) ? 1 : 0;
	if (rounds != 0 && sg.arrived > 0 && accounted >= sg.size) sg.cv.notify_all();     // a waiting lane runs the round (group_seed's loop)
	if (rounds > 0 && sg.arrived == 0 && accounted >= sg.size) {
		// every lane of the group sits this round out: nobody will run it, so it is over here (the lanes would wait for it for ever)
		for (int i = 0; i < sg.size; ++i)
			if (sg.absent_rounds[i] > 0) sg.absent_rounds[i]--;
		sg.round++;
		sg.cv.notify_all();
	}
	return KG_OK;
}

int kg_stream_group_abort(kg_stream *s)
{
	if (!s) return fail(KG_ERR_ARG, "kg_stream_group_abort: null stream");
	for (std::unique_ptr<SeedGroup> &sg : s->groups) {
		std::lock_guard<std::mutex> lk(sg->mu);
		sg->aborted = true;
		sg->cv.notify_all();
	}
	return KG_OK;
}

int kg_stream_fetch_reads(kg_stream *s, int lane, uint8_t *enc, int64_t *read_off)
{
	if (!s || lane < 0 || lane >= (int)s->lanes.size() || !read_off) return fail(KG_ERR_ARG, "kg_stream_fetch_reads: bad argument");
	Lane &l = s->lanes[(size_t)lane];
	if (!l.have_batch) return fail(KG_ERR_ARG, "kg_stream_fetch_reads: no parsed batch in lane %d", lane);
	HIP_TRY(hipSetDevice(s->ix->device));
	HIP_TRY(kgi_sync(l.ws));
	HIP_TRY(hipMemcpy(read_off, l.ws->d_read_off, 8 * (size_t)(l.parsed.n_reads + 1), hipMemcpyDeviceToHost));
	if (enc && l.parsed.n_bases > 0) HIP_TRY(hipMemcpy(enc, l.ws->d_enc, (size_t)l.parsed.n_bases, hipMemcpyDeviceToHost));
	return KG_OK;
}

int kg_stream_fetch(kg_stream *s, int lane, int64_t first, int64_t count)
{
	if (!s || lane < 0 || lane >= (int)s->lanes.size()) return fail(KG_ERR_ARG, "kg_stream_fetch: bad lane");
	Lane &l = s->lanes[(size_t)lane];
	if (first < 0 || count < 0 || first + count > l.mapped_reads || !l.d_records) return fail(KG_ERR_ARG, "kg_stream_fetch: reads [%lld, %lld) are not of the lane's mapped batch (%lld reads)", (long long)first, (long long)(first + count), (long long)l.mapped_reads);
	if (count == 0) return KG_OK;
	kg_workspace *ws = l.ws;
	HIP_TRY(hipSetDevice(ws->ix->device));
	hipStream_t st = ws->stream;
	int64_t *tot = l.h_meta + FQM_WORDS;              // (words 18 .. 19 of the lane's page-locked scratch: the seed offsets at either end)
	HIP_TRY(hipMemcpyAsync(l.h_records + first, l.d_records + first, sizeof(kg_aln_record) * (size_t)count, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(l.h_cand_off + first, ws->d_cand_off + first, 8 * (size_t)(count + 1), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&tot[18], ws->d_cseed_off + first, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&tot[19], ws->d_cseed_off + first + count, 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(kgi_sync(ws));
	const int64_t c0 = l.h_cand_off[first], c1 = l.h_cand_off[first + count], s0 = tot[18], s1 = tot[19];
	if (c0 < 0 || c1 < c0 || c1 > l.h_cand_capacity || s0 < 0 || s1 < s0 || s1 > l.h_cand_capacity) return fail(KG_ERR_NO_DEVICE, "kg_stream_fetch: the batch's candidate offsets are not what kg_stream_map left");
	if (c1 > c0) HIP_TRY(hipMemcpyAsync(l.h_cands + c0, ws->d_dense_cands + c0, sizeof(kg_candidate) * (size_t)(c1 - c0), hipMemcpyDeviceToHost, st));
	if (s1 > s0) HIP_TRY(hipMemcpyAsync(l.h_cand_seeds + s0, ws->d_dense_seeds + s0, sizeof(kg_seed) * (size_t)(s1 - s0), hipMemcpyDeviceToHost, st));
	HIP_TRY(kgi_sync(ws));
	return KG_OK;
}

int kg_stream_timing(kg_stream *s, kg_stream_timing_t *out, int reset)
{
	if (!s || !out) return fail(KG_ERR_ARG, "kg_stream_timing: null argument");
	std::lock_guard<std::mutex> lk(s->mu);
	*out = s->total;
	if (reset) s->total = kg_stream_timing_t{};
	return KG_OK;
}

}  // extern "C"
