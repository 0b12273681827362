// long_kernels.hip -- the per-read report of long reads (-pacbio) on the device (gfx950).
//
// Reference: the bPacBioData branch of ReadMapping(), src/Mapping.cpp:513-530, after chaining:
//   RemoveRedundantCandidates (src/Mapping.cpp:317-346; -pacbio: only the candidates of the best seed score stay)
//   GenMappingReport (src/AlignmentCandidates.cpp:624-745), which for -pacbio stops at the first candidate that scores (:640-644):
//     IdentifyNormalPairs(rlen, -1, SeedVec) (:420-490).  A PacBio candidate holds its seeds in (gPos, rPos) order with strictly
//       ascending read positions (GenerateAlignmentCandidateForPacBioSeq picks them that way, :171-224), so RemoveTandemRepeatSeeds
//       and RemoveTranslocatedSeeds (:235-321) find nothing; CheckOverlappingSeeds (:375-418) is sequential by nature and runs on
//       one lane when any two neighbours overlap; the gap pairs then interleave with the seeds (their keys sort right behind the
//       seed they follow), plus the head and tail pairs;
//     CheckCoordinateValidity (:582-610);
//     per pair: simple -> M; head / tail above 3000 -> S (:671-676, :690-695); ProcessNormalSequencePair's shortcuts
//       (src/tools.cpp:229-246); everything else is a request to the fragment kernels (GenerateNormalPairAlignment,
//       src/tools.cpp:142-223; frag_kernels.hip);
//     after those: AddNewCigarElements (src/tools.cpp:49-104), CheckLocalAlignmentQuality (:255-290) and the gap trimming of
//       ProcessHead/TailSequencePair (:292-397) read off the op strings, GenCoordinateInfo / GenerateCIGAR (:492-562);
//   SetSingleAlignmentFlag, EvaluateMAPQ (src/Mapping.cpp:49-70, 160-175).
// Kernels:
//   lr_select_kernel   one read per lane: which candidates take part, their slices of the pair pool
//   lr_plan_kernel     one wave per candidate: normal pairs, validity, pass 1 (immediate results, fragment requests)
//   lr_finish_kernel   one wave per candidate: pass 2 -- 64 columns of an op string per step (runs by ballot, identical bases by a
//                      compare against the 2-bit text), merged CIGAR elements, score, coordinates
//   lr_final_kernel    one read per lane: best / second best, flag, MAPQ, the record
//   lr_text_kernel     one wave per read: the CIGAR string (reversed for the reverse strand) into the text pool
// Byte / integer work, no MFMA.  A read the kernels do not take -- a literal '-' among its characters (the reference's scans take
// it for a gap column), a fragment outside the fragment kernels' envelope, seeds that CheckOverlappingSeeds leaves out of order --
// comes back as KG_ALN_HOST and is mapped by the caller's implementation of the same reference code.
#include "long_kernels.hpp"

#include <hipcub/hipcub.hpp>

namespace kg {

namespace {

__device__ __forceinline__ int text_code_at(const uint8_t *text, int64_t p) { return (text[(uint64_t)p >> 2] >> (((int)p & 3) << 1)) & 3; }
__device__ __forceinline__ int text_char_at(const uint8_t *text, int64_t p) { return (int)((0x54474341u >> (8 * text_code_at(text, p))) & 0xffu); }   // RefSequence[p]: "ACGT"[code]

__device__ __forceinline__ int lower_bound_end(const int64_t *ends, int n, int64_t g)     // ChrLocMap.lower_bound(g): first key >= g, n when none
{
	int lo = 0, hi = n;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (ends[mid] < g) lo = mid + 1;
		else hi = mid;
	}
	return lo;
}

__device__ __forceinline__ uint64_t below_mask(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

__device__ __forceinline__ long long shfl_i64(long long v, int src)
{
	return (long long)(((unsigned long long)(uint32_t)__shfl((int)(uint32_t)((unsigned long long)v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src));
}

// CheckSeedOverlapping, src/AlignmentCandidates.cpp:323-373 (simple pairs: rLen == gLen throughout)
__device__ bool resolve_overlap(kg_seed &p1, kg_seed &p2)
{
	bool master = true;
	int ov;
	if ((ov = p1.rPos + p1.len - p2.rPos) > 0) {
		if (p1.len < p2.len) {
			master = false;
			if (p1.len > ov) p1.len -= ov;
			else p1.len = 0;
		} else if (p2.len > ov) { p2.rPos += ov; p2.gPos += ov; p2.len -= ov; }
		else p2.len = 0;
	}
	if (p1.len > 0 && p2.len > 0 && (ov = (int)(p1.gPos + p1.len - p2.gPos)) > 0) {
		if (p1.len < p2.len) {
			master = false;
			if (p1.len > ov) p1.len -= ov;
			else p1.len = 0;
		} else if (p2.len > ov) { p2.rPos += ov; p2.gPos += ov; p2.len -= ov; }
		else p2.len = 0;
	}
	return master;
}

// CheckOverlappingSeeds + RemoveNullSeeds, src/AlignmentCandidates.cpp:226-233, 375-418, by one lane, in place; returns the new count
__device__ int check_overlaps(kg_seed *S, int num)
{
	for (int i = 0; i < num;) {
		if (S[i].len > 0) {
			kg_seed si = S[i];
			const int r_end = si.rPos + si.len - 1;
			const int64_t g_end = si.gPos + si.len - 1;
			for (int j = i + 1; j < num; ++j) {
				kg_seed sj = S[j];
				if (sj.len == 0) continue;
				if (r_end < sj.rPos && g_end < sj.gPos) break;
				const bool master = resolve_overlap(si, sj);
				S[j] = sj;
				if (!master) break;
			}
			S[i] = si;
			if (si.len == 0) {
				int q = i - 1;
				while (q > 0 && S[q].len == 0) q--;
				i = q < 0 ? 0 : q;
			} else i++;
		} else i++;
	}
	int w = 0;
	for (int i = 0; i < num; ++i) {
		const kg_seed s = S[i];
		if (s.len != 0) { if (w != i) S[w] = s; w++; }
	}
	return w;
}

__device__ __forceinline__ void store_pair(LrPair *p, int64_t gPos, int rPos, int rLen, int gLen, int kind)
{
	LrPair x;
	x.gPos = gPos; x.rPos = rPos; x.rLen = rLen; x.gLen = gLen; x.v = 0; x.kind = (uint8_t)kind; x.op = 0; x.pad[0] = x.pad[1] = 0;
	*p = x;
}

__device__ __forceinline__ int digits_of(int v)
{
	int d = 1;
	while (v >= 10) { v /= 10; d++; }
	return d;
}

}  // namespace

// ---- which candidates take part --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lr_select_kernel(LrArgs a)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (int64_t)gridDim.x * blockDim.x) {
		a.r_host[r] = 0;
		a.r_best[r] = -1;
		a.cig_bytes[r] = 0;
		if (r == 0) a.cig_bytes[a.n_reads] = 0;              // (the scan's tail: the total)
		const int64_t c0 = a.cand_off[r], c1 = a.cand_off[r + 1];
		// RemoveRedundantCandidates, src/Mapping.cpp:317-346: with more than one candidate only the best seed score stays (-pacbio: thr = score1)
		int s1 = 0;
		if (c1 - c0 > 1)
			for (int64_t c = c0; c < c1; ++c) s1 = max(s1, a.cands[c].score);
		for (int64_t c = c0; c < c1; ++c) {
			const kg_candidate cd = a.cands[c];
			int lc = -1;
			if (cd.score != 0 && cd.score >= s1 && cd.count > 0) {
				const unsigned long long need = 2ull * (unsigned long long)cd.count + 3ull;
				const unsigned long long po = atomicAdd(&a.ctl[LC_POOL], need);
				if (po + need <= (unsigned long long)a.pool_capacity) {
					lc = (int)atomicAdd(&a.ctl[LC_LIVE], 1ull);
					LrCand k;
					k.pair_off = (int64_t)po; k.elem_off = 0; k.pos = 0; k.cand = (int32_t)c; k.read = (int32_t)r; k.n_pairs = 0; k.state = LS_INVALID;
					k.score = 0; k.chr = 0; k.n_elems = 0; k.text_bytes = 0; k.fwd = 1;
					for (int t = 0; t < 7; ++t) k.pad[t] = 0;
					a.lcs[lc] = k;
				} else a.r_host[r] = 1;
			}
			a.cand_lc[c] = lc;
		}
	}
}

// ---- IdentifyNormalPairs of the read, validity, pass 1 ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lr_plan_kernel(LrArgs a)
{
	__shared__ int s_m;
	const int lane = threadIdx.x;
	const uint64_t below = below_mask(lane);
	const unsigned long long n_live = a.ctl[LC_LIVE];
	for (unsigned long long lc = blockIdx.x; lc < n_live; lc += gridDim.x) {
		LrCand &k = a.lcs[lc];
		const kg_candidate cd = a.cands[k.cand];
		kg_seed *S = a.seeds + cd.first;
		const int64_t base = a.read_off[k.read];
		const int rlen = (int)(a.read_off[k.read + 1] - base);
		LrPair *P = a.pool + k.pair_off;
		int m = cd.count;
		// ---- CheckOverlappingSeeds: nothing to do unless two neighbours overlap in the read or in the text ----
		bool bad = false;
		for (int i = lane; i + 1 < m; i += 64) {
			const kg_seed s0 = S[i], s1 = S[i + 1];
			bad = bad || !(s0.rPos + s0.len <= s1.rPos && s0.gPos + s0.len <= s1.gPos);
		}
		int state = LS_PLANNED;
		if (__ballot(bad)) {
			if (lane == 0) { s_m = check_overlaps(S, m); __threadfence(); }
			__syncthreads();
			m = s_m;
			__syncthreads();
			// what is left must be in order and apart in both sequences: the gap pairs then sort right behind the seed they follow
			bad = false;
			for (int i = lane; i < m; i += 64) {
				const kg_seed s0 = S[i];
				bad = bad || s0.len <= 0;
				if (i + 1 < m) { const kg_seed s1 = S[i + 1]; bad = bad || !(s0.rPos + s0.len <= s1.rPos && s0.gPos + s0.len <= s1.gPos); }
			}
			if (__ballot(bad) || m <= 0) {
				state = LS_HOST;
				if (lane == 0) atomicAdd(&a.ctl[LC_R_ORDER], 1ull);
			}
			if (lane == 0) atomicAdd(&a.ctl[LC_R_OVERLAP], 1ull);
		}
		int total = 0;
		if (state == LS_PLANNED) {
			// ---- the seeds with the gap pairs between neighbours interleaved, head and tail (:437-488; glen = -1: the text side's gap is the read's) ----
			const kg_seed first = S[0], last = S[m - 1];
			const int h_r = first.rPos > 0 ? first.rPos : 0;
			int before = h_r > 0 ? 1 : 0;
			for (int c0 = 0; c0 < m; c0 += 64) {
				const int i = c0 + lane;
				kg_seed s{}, nx{};
				bool gap = false;
				int rg = 0, gg = 0;
				if (i < m) {
					s = S[i];
					if (i + 1 < m) {
						nx = S[i + 1];
						rg = nx.rPos - (s.rPos + s.len);
						gg = (int)(nx.gPos - (s.gPos + s.len));
						gap = rg > 0 || gg > 0;
					}
				}
				const uint64_t mgap = __ballot(gap);
				if (i < m) {
					const int d = before + lane + __popcll(mgap & below);
					store_pair(&P[d], s.gPos, s.rPos, s.len, s.len, LP_SIMPLE);
					if (gap) store_pair(&P[d + 1], s.gPos + s.len, s.rPos + s.len, rg, gg, LP_NONE);
				}
				before += (m - c0 < 64 ? m - c0 : 64) + __popcll(mgap);
			}
			total = before;
			const int t_r = rlen - (last.rPos + last.len);
			if (lane == 0) {
				if (h_r > 0) { const int64_t g = first.gPos - h_r; store_pair(&P[0], g < 0 ? 0 : g, 0, h_r, h_r, LP_NONE); }
				if (t_r > 0) store_pair(&P[total], last.gPos + last.len, last.rPos + last.len, t_r, t_r, LP_NONE);
			}
			if (t_r > 0) total++;
			__threadfence();
			__syncthreads();
			// ---- CheckCoordinateValidity (:582-610): the first and the last pair both have a text side ----
			const LrPair pf = P[0], pl = P[total - 1];
			const int64_t g1 = pf.gPos, g2 = pl.gPos + pl.gLen - 1, L = a.genome_size;
			bool valid = !((g1 < L && g2 >= L) || (g1 >= L && g2 < L));
			if (valid) {
				const int i1 = lower_bound_end(a.contig_end, a.n_ends, g1), i2 = lower_bound_end(a.contig_end, a.n_ends, g2);
				valid = i1 < a.n_ends && i2 < a.n_ends && a.end_chr[i1] == a.end_chr[i2];
			}
			if (!valid) state = LS_INVALID;
		}
		if (state == LS_PLANNED) {
			// ---- pass 1: what every pair is (report_plan / plan_pair of the host pipeline; src/AlignmentCandidates.cpp:655-705, src/tools.cpp:225-397) ----
			for (int c0 = 0; c0 < total; c0 += 64) {
				const int j = c0 + lane;
				int kind = LP_NONE, op = 0, v = 0, cols = 0, side = 0;
				LrPair p{};
				if (j < total) {
					p = P[j];
					kind = p.kind;
					if (kind != LP_SIMPLE && !(p.rLen == 0 && p.gLen == 0)) {
						const bool end_pair = j == 0 || j == total - 1;
						const uint8_t *rd = a.enc + base + p.rPos;
						if (end_pair && p.rLen > 3000) { kind = LP_IMM; op = 'S'; v = -1; }                                 // :671-676, :690-695
						else if (!end_pair && (p.rLen == 0 || p.gLen == 0)) { kind = LP_IMM; op = p.rLen > 0 ? 'I' : 'D'; v = 0; }   // src/tools.cpp:229-233
						else {
							bool done = false;
							if (!end_pair && p.rLen == p.gLen) {
								// the <= 2-mismatch shortcut of ProcessNormalSequencePair (src/tools.cpp:240; raw characters)
								int n = 0;
								for (int t = 0; t < p.rLen && n <= 2; ++t) n += (int)rd[t] != text_char_at(a.text, p.gPos + t);
								if (n <= 2 && n <= (int)((double)p.rLen * 0.2)) { kind = LP_IMM; op = 'M'; v = p.rLen - n; done = true; }
							}
							if (!done && p.rLen == 1 && p.gLen == 1 && rd[0] != '-') {
								// one base against one base: nw_alignment can only answer with the diagonal, the quality check passes a single
								// column, nothing is trimmed, AddNewCigarElements books 1M with one identical base iff the characters are equal
								kind = LP_IMM; op = 'M'; v = (int)rd[0] == text_char_at(a.text, p.gPos) ? 1 : 0; done = true;
							}
							if (!done) { kind = LP_REQ; cols = p.rLen + p.gLen; side = p.rLen > p.gLen ? p.rLen : p.gLen; }
						}
					}
				}
				// the requests of this step: one reservation per list for the wave
				const uint64_t mreq = __ballot(kind == LP_REQ);
				if (mreq) {
					long long incl = cols;
					for (int off = 1; off < 64; off <<= 1) { const long long t = shfl_i64(incl, (lane - off) & 63); if (lane >= off) incl += t; }
					const long long sum = shfl_i64(incl, 63);
					int mx = side;
					for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
					unsigned long long rb = 0, cb = 0;
					if (lane == 0) {
						rb = atomicAdd(&a.ctl[LC_REQ], (unsigned long long)__popcll(mreq));
						cb = atomicAdd(&a.ctl[LC_COLS], (unsigned long long)sum);
						atomicMax(&a.ctl[LC_MAXLEN], (unsigned long long)mx);
					}
					rb = (unsigned long long)shfl_i64((long long)rb, 0); cb = (unsigned long long)shfl_i64((long long)cb, 0);
					if (kind == LP_REQ) {
						const unsigned long long q = rb + (unsigned long long)__popcll(mreq & below);
						if (q < (unsigned long long)a.req_capacity) {
							a.req_f1[q] = base + p.rPos; a.req_g[q] = p.gPos; a.req_rl[q] = p.rLen; a.req_gl[q] = p.gLen;
							a.req_oo[q] = (int64_t)(cb + (unsigned long long)(incl - cols));
							v = (int)q;
						} else { kind = LP_NONE; state = LS_HOST; }
					}
				}
				if (j < total && kind != LP_SIMPLE) { p.kind = (uint8_t)kind; p.op = (uint8_t)op; p.v = v; P[j] = p; }
			}
			if (__ballot(state == LS_HOST)) state = LS_HOST;
		}
		if (lane == 0) {
			k.n_pairs = total;
			k.state = state;
			if (state == LS_HOST) a.r_host[k.read] = 1;
		}
		__syncthreads();
	}
}

// ---- pass 2 -----------------------------------------------------------------------------------------------------------------------
namespace {

// the merged CIGAR of one candidate while it is being assembled: the run in progress and the elements closed so far
struct Cigar {
	uint32_t *E;
	int ne;           // elements written
	int op, len;      // the run in progress (op -1: none yet)
	bool raw;         // cigar_vec is not empty
};

__device__ __forceinline__ int cigar_op_of(int c) { return c == 'M' ? 0 : c == 'I' ? 1 : c == 'D' ? 2 : 3; }

__device__ __forceinline__ void cigar_emit(Cigar &cg, int op, int len, int lane)      // cigar_vec.push_back + GenerateCIGAR's merging (:492-513); wave-uniform
{
	cg.raw = true;
	if (op == cg.op) { cg.len += len; return; }
	if (cg.len > 0) { if (lane == 0) cg.E[cg.ne] = ((uint32_t)cg.len << 2) | (uint32_t)cg.op; cg.ne++; }
	cg.op = op; cg.len = len;
}

struct ScanStats {
	int runs, n, same;        // CheckLocalAlignmentQuality's iStatus, n, n - mis
	int q[4], o[3];           // start columns of the first four runs, ops of the first three
};

// columns [from, to) of the alignment (read characters at rd, text at g, op string): with emit, AddNewCigarElements
// (src/tools.cpp:49-104) into cg -- 64 columns per step; returns the identical bases.  a0 / b0: read / text bases consumed before `from`.
template <bool kEmit, bool kStats>
__device__ int scan_columns(const LrArgs &a, const uint8_t *rd, int64_t g, const uint8_t *ops, int from, int to, int a0, int b0, int lane, uint64_t below, Cigar &cg, ScanStats *st, bool &dash)
{
	int same = 0, a_run = a0, b_run = b0;
	int prev_raw = -1;        // (stats) the op of the column before this step
	for (int c0 = from; c0 < to; c0 += 64) {
		const int c = c0 + lane;
		const bool valid = c < to;
		const int op = valid ? (int)ops[c] : 255;
		const uint64_t mD = __ballot(op == KG_OP_DIAG), m1 = __ballot(op == KG_OP_GAP1), m2 = __ballot(op == KG_OP_GAP2);
		const int nvalid = to - c0 < 64 ? to - c0 : 64;
		const int ra = a_run + __popcll((mD | m2) & below), tb = b_run + __popcll((mD | m1) & below);
		const int rc = (valid && op != KG_OP_GAP1) ? (int)rd[ra] : 0;
		const bool eq = op == KG_OP_DIAG && rc == text_char_at(a.text, g + tb);
		same += __popcll(__ballot(eq));
		if (__ballot(rc == '-')) dash = true;      // a literal '-' in the read: the reference's scans take it for a gap column
		const int cop = op == KG_OP_DIAG ? 0 : op == KG_OP_GAP2 ? 1 : 2;      // M, I (gap in the text side), D (gap in the read side)
		if (kStats) {
			st->n += __popcll(mD);
			const int up = __shfl(op, (lane - 1) & 63);                        // (outside the conditional: lane 0 must take part for lane 1 to read it)
			const int prev = lane == 0 ? prev_raw : up;
			uint64_t B = __ballot(valid && op != prev);
			while (B) {
				const int p = __ffsll((unsigned long long)B) - 1;
				B &= B - 1;
				if (st->runs < 4) st->q[st->runs] = c0 + p;
				if (st->runs < 3) st->o[st->runs] = __shfl(op, p);
				st->runs++;
			}
			prev_raw = __shfl(op, nvalid - 1);
		}
		if (kEmit) {
			// the runs that END inside this step are closed by the lane that starts the next one
			const int up = __shfl(cop, (lane - 1) & 63);
			const int prev = lane == 0 ? cg.op : up;
			const uint64_t B = __ballot(valid && cop != prev);
			const bool skip0 = cg.len == 0 && (B & 1ull);                 // nothing was in progress at column 0
			if (valid && cop != prev) {
				const uint64_t lower = B & below;
				const int k_idx = __popcll(lower);                        // this lane's boundary is the k-th of the step
				int len_closed, op_closed;
				if (lower == 0) { len_closed = cg.len + lane; op_closed = cg.op; }
				else { const int pp = 63 - __clzll((long long)lower); len_closed = lane - pp; op_closed = prev; }
				if (len_closed > 0) cg.E[cg.ne + k_idx - (skip0 ? 1 : 0)] = ((uint32_t)len_closed << 2) | (uint32_t)op_closed;
			}
			if (B) {
				const int last = 63 - __clzll((long long)B);
				cg.ne += __popcll(B) - (skip0 ? 1 : 0);
				cg.len = nvalid - last;
			} else cg.len += nvalid;
			cg.op = __shfl(cop, nvalid - 1);
			cg.raw = true;
		}
		a_run += __popcll(mD | m2); b_run += __popcll(mD | m1);
	}
	return same;
}

}  // namespace

__global__ __launch_bounds__(64) void lr_finish_kernel(LrArgs a)
{
	const int lane = threadIdx.x;
	const uint64_t below = below_mask(lane);
	const unsigned long long n_live = a.ctl[LC_LIVE];
	for (unsigned long long lc = blockIdx.x; lc < n_live; lc += gridDim.x) {
		LrCand &k = a.lcs[lc];
		if (k.state != LS_PLANNED) continue;
		const int num = k.n_pairs;
		LrPair *P = a.pool + k.pair_off;
		const int64_t base = a.read_off[k.read];
		// room for the merged elements: at most two per pair (a soft clip beside a head's or tail's runs) and one per run of its alignments
		long long need = 0;
		bool host = false;
		for (int j = lane; j < num; j += 64) {
			const LrPair p = P[j];
			need += 2;
			if (p.kind == LP_REQ) { if (a.status[p.v]) host = true; else need += a.runs[p.v]; }
		}
		for (int off = 32; off > 0; off >>= 1) need += shfl_i64(need, lane ^ off);
		if (__ballot(host)) {
			if (lane == 0) { k.state = LS_HOST; a.r_host[k.read] = 1; atomicAdd(&a.ctl[LC_R_FRAG], 1ull); }
			continue;
		}
		unsigned long long eo = 0;
		if (lane == 0) eo = atomicAdd(&a.ctl[LC_ELEMS], (unsigned long long)need);
		eo = (unsigned long long)shfl_i64((long long)eo, 0);
		if (eo + (unsigned long long)need > (unsigned long long)a.elem_capacity) {
			if (lane == 0) { k.state = LS_HOST; a.r_host[k.read] = 1; atomicAdd(&a.ctl[LC_R_ELEMS], 1ull); }
			continue;
		}
		Cigar cg;
		cg.E = a.elems + eo; cg.ne = 0; cg.op = -1; cg.len = 0; cg.raw = false;
		int score = 0;
		bool dash = false;
		int64_t g_first = P[0].gPos, g_last = P[num - 1].gPos;
		int gl_last = P[num - 1].gLen;
		for (int j = 0; j < num; ++j) {
			const LrPair p = P[j];
			if (p.kind == LP_NONE) continue;
			if (p.kind == LP_SIMPLE) { cigar_emit(cg, 0, p.rLen, lane); score += p.rLen; continue; }
			const bool head = j == 0, tail = j == num - 1 && !head;
			int s = 0;
			if (p.kind == LP_IMM) {
				if (p.op) cigar_emit(cg, cigar_op_of(p.op), p.op == 'D' ? p.gLen : p.rLen, lane);
				s = p.v;
			} else {
				const uint8_t *rd = a.enc + base + p.rPos;
				const uint8_t *ops = a.ops + a.req_oo[p.v];
				const int L = a.aln_len[p.v];
				if (!head && !tail) {
					s = scan_columns<true, false>(a, rd, p.gPos, ops, 0, L, 0, 0, lane, below, cg, nullptr, dash);
				} else {
					// ProcessHeadSequencePair / ProcessTailSequencePair after the alignment (src/tools.cpp:314-339, 366-394): the quality check
					// passes at most three runs, so the first three say everything there is to trim and to print
					ScanStats st;
					st.runs = 0; st.n = 0; st.same = 0;
					for (int t = 0; t < 4; ++t) st.q[t] = L;
					for (int t = 0; t < 3; ++t) st.o[t] = -1;
					st.same = scan_columns<false, true>(a, rd, p.gPos, ops, 0, L, 0, 0, lane, below, cg, &st, dash);
					const int mis = st.n - st.same;
					const bool ok = !(st.runs >= 4 || (mis >= 3 && mis >= (int)((double)st.n * 0.3)));      // CheckLocalAlignmentQuality, :255-290
					if (!ok) { cigar_emit(cg, 3, p.rLen, lane); s = 0; }
					else {
						int rl[3];
						for (int t = 0; t < 3; ++t) rl[t] = t < st.runs ? (t + 1 < st.runs ? st.q[t + 1] : L) - st.q[t] : 0;
						int lo = 0, hi = st.runs;             // the runs that stay
						if (head) {
							if (lo < hi && st.o[lo] == KG_OP_GAP1) { g_first = p.gPos + rl[lo]; lo++; }         // leading text bases against gaps: the pair starts behind them
							if (lo < hi && st.o[lo] == KG_OP_GAP2) { cigar_emit(cg, 3, rl[lo], lane); lo++; }   // leading read bases against gaps: soft-clipped
							for (int t = lo; t < hi; ++t) cigar_emit(cg, st.o[t] == KG_OP_DIAG ? 0 : st.o[t] == KG_OP_GAP2 ? 1 : 2, rl[t], lane);
						} else {
							int clip = 0;
							if (lo < hi && st.o[hi - 1] == KG_OP_GAP1) { gl_last = p.gLen - rl[hi - 1]; hi--; }
							if (lo < hi && st.o[hi - 1] == KG_OP_GAP2) { clip = rl[hi - 1]; hi--; }
							for (int t = lo; t < hi; ++t) cigar_emit(cg, st.o[t] == KG_OP_DIAG ? 0 : st.o[t] == KG_OP_GAP2 ? 1 : 2, rl[t], lane);
							if (clip > 0) cigar_emit(cg, 3, clip, lane);
						}
						s = st.same;
					}
				}
			}
			if (head) {
				if (s > 0) score += s;
				else { g_first = P[1].gPos; }                                   // :674-686: the pair's text side collapses onto the next pair's start
			} else if (tail) {
				if (s > 0) score += s;
				else { const LrPair pv = P[j - 1]; g_last = pv.gPos + pv.gLen; gl_last = 0; }
			} else score += s;
		}
		if (cg.len > 0) { if (lane == 0) cg.E[cg.ne] = ((uint32_t)cg.len << 2) | (uint32_t)cg.op; cg.ne++; }
		if (dash) {
			if (lane == 0) { k.state = LS_HOST; a.r_host[k.read] = 1; atomicAdd(&a.ctl[LC_R_DASH], 1ull); }
			continue;
		}
		// GenCoordinateInfo (:515-562) for the first read of a "pair": forward strand below GenomeSize
		const int64_t g0 = g_first, g_end = g_last + gl_last - 1;
		int chr = 0;
		int64_t pos = 0;
		bool fwd = true;
		if (!cg.raw) score = 0;
		else {
			if (g0 < a.genome_size) {
				if (a.n_chr == 1) pos = g0 + 1;
				else {
					int i = lower_bound_end(a.contig_end, a.n_ends, g0);
					if (i >= a.n_ends) i = a.n_ends - 1;
					chr = a.end_chr[i];
					pos = g0 + 1 - a.chr_fwd_start[chr];
				}
			} else {
				fwd = false;
				if (a.n_chr == 1) pos = a.two_genome_size - g_end;
				else {
					int i = lower_bound_end(a.contig_end, a.n_ends, g0);
					if (i >= a.n_ends) i = a.n_ends - 1;
					pos = a.contig_end[i] - g_end + 1;
					chr = a.end_chr[i];
				}
			}
			if (pos <= 0) score = 0;
		}
		// the length of the CIGAR string (GenerateCIGAR prints "%d%c" per merged element)
		__threadfence();
		__syncthreads();
		int bytes = 0;
		for (int e = lane; e < cg.ne; e += 64) bytes += digits_of((int)(cg.E[e] >> 2)) + 1;
		for (int off = 32; off > 0; off >>= 1) bytes += __shfl_xor(bytes, off);
		if (lane == 0) {
			k.elem_off = (int64_t)eo; k.n_elems = cg.ne; k.text_bytes = bytes;
			k.score = score; k.chr = chr; k.pos = pos; k.fwd = fwd ? 1 : 0;
		}
	}
}

// ---- best / second best, flag, MAPQ: the record -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lr_final_kernel(LrArgs a)
{
	for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (int64_t)gridDim.x * blockDim.x) {
		kg_aln_record rec;
		rec.pos = 0; rec.mate_pos = 0; rec.kind = KG_ALN_UNMAPPED; rec.flag = 0x4; rec.chr = 0; rec.mapq = 0; rec.tlen = 0; rec.score = 0; rec.sub_score = 0;
		rec.est_lo = -1; rec.est_hi = 0x7fffffff; rec.has_mate = 0; rec.flip = 0; rec.cigar_len = 0; rec.rescue = 0;
		for (int t = 0; t < KG_ALN_CIGAR_MAX; ++t) rec.cigar[t] = 0;
		rec.next = -1; rec.primary = 1; rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
		a.cig_bytes[r] = 0;
		int best = -1;
		if (a.r_host[r]) rec.kind = KG_ALN_HOST;
		else {
			// GenMappingReport's loop (:632-745): candidates without a seed score are skipped, and once one has scored the rest only set sub_score (:640-644)
			int score = 0, sub = 0;
			for (int64_t c = a.cand_off[r]; c < a.cand_off[r + 1]; ++c) {
				const int lc = a.cand_lc[c];
				if (lc < 0) continue;
				if (score > 0) { sub = score; continue; }
				const LrCand &k = a.lcs[lc];
				if (k.state != LS_PLANNED) continue;                       // CheckCoordinateValidity failed (:653)
				if (k.score > score) { best = lc; sub = score; score = k.score; }
				else if (k.score == score) sub = score;
			}
			if (score > 0) {
				const LrCand &k = a.lcs[best];
				const int rlen = (int)(a.read_off[r + 1] - a.read_off[r]);
				rec.kind = KG_ALN_MAPPED;
				rec.flag = k.fwd ? 0 : 0x10;                               // SetSingleAlignmentFlag, src/Mapping.cpp:49-70
				rec.flip = k.fwd ? 0 : 1;
				rec.chr = k.chr; rec.pos = k.pos; rec.score = score; rec.sub_score = sub;
				int mapq = 0;
				if (score != sub) {                                        // EvaluateMAPQ, src/Mapping.cpp:160-175 (the bPacBioData branch)
					float scale = (float)(85.0 * (double)(int)ceil((double)(rlen / 100) + 0.5));
					if (scale > 2000.0f) scale = 2000.0f;
					mapq = (int)__fmul_rn(60.0f, __fdiv_rn((float)score, scale));
					if (mapq > 60) mapq = 60;
				}
				rec.mapq = mapq;
				rec.cigar_len = 255;                                       // the CIGAR lies in the text pool: lr_text_kernel fills offset and length
				a.cig_bytes[r] = k.text_bytes;
			} else best = -1;
		}
		a.r_best[r] = best;
		a.records[r] = rec;
	}
}

// ---- GenerateCIGAR (:492-513) into the text pool; the elements of a reverse-strand alignment in reverse order (GenCoordinateInfo, :540) ----
__global__ __launch_bounds__(64) void lr_text_kernel(LrArgs a)
{
	const int lane = threadIdx.x;
	for (int64_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
		const int lc = a.r_best[r];
		if (lc < 0) continue;
		const LrCand k = a.lcs[lc];
		const uint32_t *E = a.elems + k.elem_off;
		const int64_t at = a.cig_bytes[r];                                 // (scanned in place: now the offset)
		char *out = a.cigar + at;
		int run = 0;
		for (int e0 = 0; e0 < k.n_elems; e0 += 64) {
			const int e = e0 + lane;
			const bool valid = e < k.n_elems;
			const uint32_t v = valid ? E[k.fwd ? e : k.n_elems - 1 - e] : 0;
			int len = (int)(v >> 2);
			const int nd = valid ? digits_of(len) : 0, w = valid ? nd + 1 : 0;
			int incl = w;
			for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
			if (valid) {
				char *p = out + run + incl - w;
				for (int d = nd - 1; d >= 0; --d) { p[d] = (char)('0' + len % 10); len /= 10; }
				p[nd] = "MIDS"[v & 3];
			}
			run += __shfl(incl, 63);
		}
		if (lane == 0) {
			kg_aln_record &rec = a.records[r];
			*reinterpret_cast<int64_t *>(rec.cigar) = at;
			*reinterpret_cast<int32_t *>(rec.cigar + 8) = run;
		}
	}
}

static inline int grid_of(int64_t items, int block, int max_blocks)
{
	int64_t g = (items + block - 1) / block;
	if (g < 1) g = 1;
	if (g > max_blocks) g = max_blocks;
	return (int)g;
}

hipError_t launch_long_plan(const LrArgs &a, int n_cu, hipStream_t stream)
{
	hipLaunchKernelGGL(lr_select_kernel, dim3(grid_of(a.n_reads, 256, n_cu * 8)), dim3(256), 0, stream, a);
	hipLaunchKernelGGL(lr_plan_kernel, dim3(grid_of(a.n_cands, 1, n_cu * 32)), dim3(64), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_long_finish(const LrArgs &a, int n_cu, hipStream_t stream)
{
	hipLaunchKernelGGL(lr_finish_kernel, dim3(grid_of(a.n_cands, 1, n_cu * 32)), dim3(64), 0, stream, a);
	hipLaunchKernelGGL(lr_final_kernel, dim3(grid_of(a.n_reads, 256, n_cu * 8)), dim3(256), 0, stream, a);
	return hipGetLastError();
}

size_t long_scan_temp_bytes(int64_t max_items)
{
	size_t b = 0;
	(void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const int64_t *)nullptr, (int64_t *)nullptr, (int)max_items);
	return b + 256;
}

// cig_bytes[0 .. n_reads] -> its exclusive scan, in place (cig_bytes[n_reads] = bytes of the whole text pool)
hipError_t launch_long_scan(const LrArgs &a, void *temp, size_t temp_bytes, hipStream_t stream)
{
	return hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, (const int64_t *)a.cig_bytes, a.cig_bytes, (int)(a.n_reads + 1), stream);
}

hipError_t launch_long_text(const LrArgs &a, int n_cu, hipStream_t stream)
{
	hipLaunchKernelGGL(lr_text_kernel, dim3(grid_of(a.n_reads, 1, n_cu * 32)), dim3(64), 0, stream, a);
	return hipGetLastError();
}

}  // namespace kg
