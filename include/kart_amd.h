/* kart_amd.h -- C ABI of the MI355X-native Kart hot path (libkart_amd.so).
 *
 * The reference (hsinnan75/Kart v2.5.6) has no plugin/FFI layer: its "operator API" for the
 * seed-and-extend path is the set of extern C++ prototypes in src/structure.h:177-229 that the
 * worker loop ReadMapping() (src/Mapping.cpp:488-637) calls once per read.  This header is the
 * batched, device-backed replacement for exactly those calls.  Everything is plain C: opaque
 * handles, pointers and sizes, int status codes (0 = KG_OK) instead of the reference's exit(1).
 *
 *   reference call (file:line)                                   replaced by
 *   -----------------------------------------------------------  ---------------------------
 *   bwa_idx_load + RestoreReferenceInfo (src/bwt_index.cpp:148,230; src/main.cpp:192-207)
 *                                                                kg_index_load / kg_index_destroy
 *   Mapping(): MinSeedLength choice (src/Mapping.cpp:645)        kg_index_info().min_seed_len
 *   bwt_occ4 / bwt_sa (src/bwt_search.cpp:68-85,128-138)         kg_rank_sa_batch
 *   BWT_Search (src/structure.h:178, src/bwt_search.cpp:140)     } kg_seed_batch (+ _device form)
 *   IdentifySeedPairs_FastMode / _SensitiveMode                  }   mode KG_MODE_FAST / _SENSITIVE
 *       (src/structure.h:189,191; src/AlignmentCandidates.cpp:49,132)
 *   nw_alignment (src/structure.h:229, src/nw_alignment.cpp:18)  kg_nw_batch (+ _device form)
 *   GenerateAlignmentCandidateForIlluminaSeq / ForPacBioSeq      kg_candidates_batch
 *       (src/structure.h:193-194; src/AlignmentCandidates.cpp:82,171)
 *   CheckPairedAlignmentCandidates ... GenMappingReport ... EvaluateMAPQ (src/structure.h:192; src/Mapping.cpp:542-578)
 *                                                                kg_align_batch
 *
 * Threading: an index handle is immutable after load and may be shared by any number of host
 * threads; a workspace (kg_workspace) owns the scratch of one in-flight batch and must not be
 * used by two calls at once -- the analogue of the reference's "each worker owns its chunk".
 * There is no CPU fallback: every entry point fails with KG_ERR_NO_DEVICE when no gfx950
 * device is usable.
 */
#ifndef KART_AMD_H
#define KART_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KG_OK                0
#define KG_ERR_NO_DEVICE     1   /* no HIP device / HIP runtime error (see kg_last_error) */
#define KG_ERR_IO            2   /* index files missing or malformed */
#define KG_ERR_ARG           3   /* bad argument */
#define KG_ERR_CAPACITY      4   /* caller-provided output buffer too small (needed size reported) */
#define KG_ERR_NOMEM         5

#define KG_MODE_FAST         0   /* IdentifySeedPairs_FastMode      (Illumina) */
#define KG_MODE_SENSITIVE    1   /* IdentifySeedPairs_SensitiveMode (-pacbio)  */
#define KG_INPUT_ASCII    0x100   /* OR into mode: the read bytes are the characters themselves; the library applies
                                     nst_nt4_table (EnCodeReadSeq, src/Mapping.cpp:482-485) on the device */

#define KG_OCC_THR_DEFAULT   50  /* OCC_Thr, src/bwt_search.cpp:3 */

/* SA placement on the device: the reference layout (one sample per 32 ranks, LF-walk to locate,
 * src/bwt_search.cpp:128-138) or the full suffix array expanded once at load into HBM. */
#define KG_SA_SAMPLED        0
#define KG_SA_FULL           1
/* the smaller index: every 4th / 8th entry of the suffix array resident (built at load from the full expansion, which is then
 * freed), no triple planes; single-suffix searches still finish against the text after a walk of at most 3 / 7 rank steps */
#define KG_SA_DENSE4         4
#define KG_SA_DENSE8         8
/* the compact index: the whole suffix array in 5-byte entries (4-byte ones where the text allows), a q-mer table of a quarter the
 * size, no triple planes -- ~66 GB instead of ~168 GB for a human genome, no walks */
#define KG_SA_FULL40         5
/* the 5-byte suffix array with everything that saves rank steps: the full q-mer table (4^q ~ text length) and the triple planes --
 * ~150 GB for a human genome, three read bases per rank pair */
#define KG_SA_FULL40_WIDE    6
/* what the host pipeline asks for unless told otherwise: KG_SA_FULL for a text of fewer than 2^32 symbols (forward + reverse strand; there the
 * whole index is a few hundred MB either way); above, KG_SA_FULL40_WIDE where the device keeps 64 GB free behind it (one process on a 288 GB
 * device) and KG_SA_FULL40 -- 67 GB -- where it does not (several processes sharing a device).  kg_index_info::sa_mode reports what it became. */
#define KG_SA_AUTO           (-1)

typedef struct kg_index kg_index;
typedef struct kg_workspace kg_workspace;

/* One exact-match seed = the non-derivable part of SeedPair_t (src/structure.h:106-114):
 * bSimple = true, gLen = rLen = len, PosDiff = gPos - rPos. */
typedef struct {
	int64_t gPos;   /* text coordinate in [0, 2L): forward strand < L <= reverse strand */
	int32_t rPos;
	int32_t len;
} kg_seed;

typedef struct {
	int64_t  genome_size;    /* l_pac  (GenomeSize) */
	uint64_t seq_len;        /* 2*l_pac (TwoGenomeSize) = BWT length */
	uint64_t primary;
	int32_t  n_contigs;
	int32_t  min_seed_len;   /* 13..16, src/Mapping.cpp:645 */
	int32_t  sa_mode;        /* KG_SA_SAMPLED / KG_SA_FULL / KG_SA_FULL40 / KG_SA_FULL40_WIDE / KG_SA_DENSE4 / KG_SA_DENSE8 */
	int32_t  device;
	uint64_t device_bytes;   /* HBM held by the index */
} kg_index_info_t;

typedef struct {
	const char *name;
	int64_t fwd_start;       /* Chromosome_t::FowardLocation */
	int64_t rev_start;       /* Chromosome_t::ReverseLocation */
	int64_t len;
} kg_contig_t;

/* Work counters of the last kg_seed_batch* call on a workspace, in the units of the algorithmic
 * byte count bytes_seed = 64*(lf1 + 2*lf2 + inv) + 8*sa + bases + 16*seeds. */
typedef struct {
	uint64_t searches, lf1, lf2, inv, sa, seeds, bases;
} kg_counters_t;

/* What the search kernel of the last kg_seed_batch* call itself fetched (the IMPLEMENTED algorithm: q-mer table jump, rank
 * steps on the bit-plane layout, one suffix-array gather and text comparison once an interval is a single suffix), as opposed
 * to kg_counters_t's reference-algorithm accounting.  Useful bytes per launch =
 *   8*table_lookups + 32*rank_steps (two 16-byte rank segments) + sa_entry_bytes*sa_gathers + 48*text_rounds (16 B of text +
 *   32 B of read words) + 8*window_words + 20*reads (offsets, seed count) + 32*hits (hit records written). */
typedef struct {
	uint64_t table_lookups, rank_steps, rank_steps_two_lines, sa_gathers, text_rounds, window_words, hits, searches;
	uint64_t sa_entry_bytes;
	uint64_t rank_steps_two_lines_narrow;   /* two-line rank steps on an interval of fewer than 960 suffixes (what a 960-symbols-per-line plane layout would serve from one line) */
	uint64_t double_steps;                  /* two extension steps taken at once on the pair planes (one or two 128-byte lines, 896 rows of one pair of bases each) */
	uint64_t double_steps_two_lines;        /* ... whose interval ends lie in different lines */
	uint64_t triple_steps;                  /* ... of double_steps, those that took three steps (the index holds the 64 triple planes: devices with room for 9 bytes/symbol) */
	uint64_t double_step_bytes;             /* bytes those ranks read: per line its 16-byte header and the 16-byte segment(s) holding the rows asked for */
} kg_traffic_t;

const char *kg_last_error(void);            /* thread-local message of the last failure */
int  kg_device_count(void);                 /* number of usable HIP devices (0 if none) */

/* ---- index ------------------------------------------------------------------------------ */
/* Reads <prefix>.bwt/.sa/.ann/.amb/.pac (BWA 0.7-era format written by the reference's
 * bwt_index) and uploads the FM-index + 2-bit reference to `device`. */
int  kg_index_load(const char *prefix, int device, int sa_mode, kg_index **out);
void kg_index_destroy(kg_index *ix);
int  kg_index_info(const kg_index *ix, kg_index_info_t *info);
int  kg_index_contig(const kg_index *ix, int i, kg_contig_t *out);

/* bwt_occ4 (src/bwt_search.cpp:68-85) and bwt_sa (src/bwt_search.cpp:128-138) for n ranks k[i] in [0, 2L] (k = (uint64_t)-1
 * is accepted by the rank part, as in the reference), on the device's own rank / suffix-array layouts.  Host buffers.
 * occ4[4n] (may be NULL): occurrences of A,C,G,T in BWT[0..k].  sa_walk[n] (may be NULL): the reference's sampled walk.
 * sa_full[n] (may be NULL): the expanded suffix array's entry when the index was loaded with KG_SA_FULL (there SA[0] = 2L,
 * where the reference's sa[0] = -1 yields 2^64-1), else (uint64_t)-1. */
int  kg_rank_sa_batch(kg_index *ix, const uint64_t *k, int64_t n, uint64_t *occ4, uint64_t *sa_walk, uint64_t *sa_full);
/* Known-answer check of the device-private two-step rank structure: `samples` pseudo-random intervals of every width and
 * every pair of bases, one double step against two single BWT_Search steps (src/bwt_search.cpp:157-168) on the plain rank
 * structure.  *disagreements must come back 0. */
int  kg_index_selfcheck(kg_index *ix, int64_t samples, uint64_t seed, uint64_t *disagreements);

/* Page-locked host memory for the buffers a caller hands to kg_seed_batch (the copy to the device then runs at the link's
 * rate instead of through a staging buffer).  NULL when none can be had. */
void *kg_host_alloc(size_t bytes);
void  kg_host_free(void *p);

/* ---- workspace ---------------------------------------------------------------------------- */
/* Scratch for batches of up to max_reads reads / max_bases bases on the index's device. */
int  kg_workspace_create(kg_index *ix, int64_t max_reads, int64_t max_bases, kg_workspace **out);
void kg_workspace_destroy(kg_workspace *ws);
int  kg_workspace_counters(kg_workspace *ws, kg_counters_t *out);   /* synchronises the device */
int  kg_workspace_traffic(kg_workspace *ws, kg_traffic_t *out);     /* synchronises the device */
/* The search takes two extension steps at once where it can (pair planes); the reference's per-step block accounting
 * (kg_counters_t lf1 / lf2, bwt_2occ4's one-or-two-blocks split, src/bwt_search.cpp:92) is then exact for lf1 + lf2 only.
 * enabled != 0: single steps only -- same seeds, the lf1 / lf2 split exact as well. */
int  kg_workspace_set_single_steps(kg_workspace *ws, int enabled);
/* Per-kernel timing: when enabled, every kg_seed_batch* call brackets its kernels with HIP events
 * on the launch stream; kg_workspace_kernel_ms() synchronises and returns the durations of the
 * last call in milliseconds: ms[0] search, ms[1] scan+offsets, ms[2] locate, ms[3] sort. */
int  kg_workspace_set_profiling(kg_workspace *ws, int enabled);
int  kg_workspace_kernel_ms(kg_workspace *ws, float ms[4]);

/* ---- seeding ------------------------------------------------------------------------------ */
/* Host-buffer form.  enc_bases: concatenated reads, 1 byte/base, codes 0..3 = ACGT, >3 = ambiguous
 * (EnCodeReadSeq, src/Mapping.cpp:482-485); read_offsets[n_reads+1].  On return
 * seed_offsets[n_reads+1] is filled and *seeds points at a library-owned pinned array of
 * seed_offsets[n_reads] entries, valid until the next call on the same workspace (seeds may be NULL when only
 * kg_candidates_batch is going to look at them: they then stay on the device).  Per read the
 * entries equal what IdentifySeedPairs_{Fast,Sensitive}Mode returns, in the same order
 * (sorted by (PosDiff,rPos) resp. (gPos,rPos)). */
int  kg_seed_batch(kg_workspace *ws, int mode, int min_seed_len, int occ_thr,
                   const uint8_t *enc_bases, const int64_t *read_offsets, int64_t n_reads,
                   int64_t *seed_offsets, const kg_seed **seeds);

/* Device-pointer form (inputs already resident in HBM; asynchronous on `stream`, a hipStream_t
 * passed as void*, NULL = default stream).  d_seed_offsets[n_reads+1] and d_seeds[seed_capacity]
 * are device buffers; if the batch produces more than seed_capacity seeds the surplus is dropped
 * and kg_workspace_overflow() reports the required capacity after the stream is synchronised. */
int  kg_seed_batch_device(kg_workspace *ws, int mode, int min_seed_len, int occ_thr,
                          const uint8_t *d_enc_bases, const int64_t *d_read_offsets, int64_t n_reads,
                          int64_t n_bases, int64_t *d_seed_offsets, kg_seed *d_seeds,
                          int64_t seed_capacity, void *stream);
int64_t kg_workspace_overflow(kg_workspace *ws);   /* 0 = fitted, else seeds needed */
/* SensitiveMode on long reads walks from every 512th position of a read at once (IdentifySeedPairs_SensitiveMode,
 * src/AlignmentCandidates.cpp:132-169, is one walk per read); on exact long reads those walks never merge and the batch
 * can outgrow the workspace's hit list.  kg_seed_batch then seeds the batch again with one walk per read -- same seeds --
 * and this counts how often that happened on the workspace. */
int64_t kg_workspace_segment_fallbacks(kg_workspace *ws);

/* ---- chaining ----------------------------------------------------------------------------------- */
/* Candidates of every read of the batch that the LAST kg_seed_batch call on this workspace seeded (its
 * seeds are still resident on the device).  Per read r: n_cands[r] candidates, stored densely in read order --
 * those of read r are (*cands)[o .. o + n_cands[r]) with o = n_cands[0] + ... + n_cands[r-1] -- and their seeds
 * (re-sorted by (gPos,rPos) for Illumina, in pick order for PacBio, exactly as the reference leaves
 * AlignmentCandidate_t::SeedVec) at (*cand_seeds)[first .. first+count).  *cands and *cand_seeds point at library-owned
 * pinned arrays of *n_cands_total / *n_cand_seeds_total entries; the library rotates through four sets of them, so they stay
 * valid while the next three batches go through the workspace.
 * n_reads and n_seeds (= seed_offsets[n_reads]) must be those of the seeding call; a mismatch is KG_ERR_ARG. */
typedef struct {
	int64_t posDiff;    /* AlignmentCandidate_t::PosDiff (clamped at 0) */
	int32_t score;      /* sum of seed lengths */
	int32_t count;      /* seeds in the candidate */
	int64_t first;      /* index of its first seed in cand_seeds */
} kg_candidate;
int  kg_candidates_batch(kg_workspace *ws, int pacbio, int max_gaps, int64_t n_reads, int64_t n_seeds, int32_t *n_cands,
                         const kg_candidate **cands, int64_t *n_cands_total, const kg_seed **cand_seeds, int64_t *n_cand_seeds_total);

/* ---- the per-read report on the device --------------------------------------------------------------------------- */
/* Everything ReadMapping() does between chaining and the SAM text, for the reads of the batch the last kg_seed_batch +
 * kg_candidates_batch calls on this workspace left on the device, in the reference's short-read configuration (not -pacbio):
 *   CheckPairedAlignmentCandidates / RemoveUnMatedAlignmentCandidates / RemoveRedundantCandidates (src/Mapping.cpp:317-427),
 *   GenMappingReport: IdentifyNormalPairs + filters (src/AlignmentCandidates.cpp:226-490), Process{Head,Normal,Tail}SequencePair
 *   (src/tools.cpp:225-397) with nw_alignment on the device, GenCoordinateInfo / GenerateCIGAR / GapPenalty (:492-745),
 *   CheckPairedFinalAlignments, Set{Paired,Single}AlignmentFlag, EvaluateMAPQ (src/Mapping.cpp:49-175, 429-480),
 *   and what OutputPairedAlignments / OutputSingledAlignments would print for every read (:177-315), with or without -m.
 * One record per read (with multi_hit further records of a read are chained through `next`; a record whose FLAG the reference
 * never assigns -- SURVEY App. B-12 -- carries unset_flag).  A read pair the device path does not take -- RescueUnpairedAlignment is due, a gap fragment needs the
 * 8-mer partition (both sides > 30), a candidate has more seeds or a longer CIGAR than the kernels hold -- comes back with
 * kind KG_ALN_HOST on both reads: the caller maps it with its own implementation of the same reference code (the candidates
 * kg_candidates_batch returned are unmodified). */
#define KG_ALN_NONE      0   /* nothing is printed for this read (the reference's loop finds no candidate to print) */
#define KG_ALN_UNMAPPED  1   /* the unmapped record: FLAG as given */
#define KG_ALN_MAPPED    2
#define KG_ALN_HOST      3   /* not decided here */
#define KG_ALN_CIGAR_MAX 48
typedef struct {
	int64_t pos;             /* 1-based position on the contig (AlignmentReport_t::coor.gPos) */
	int64_t mate_pos;
	int32_t kind;
	int32_t flag, chr, mapq, tlen;
	int32_t score, sub_score;    /* AS / XS; NM = rlen - score */
	int32_t est_lo, est_hi;  /* pairs: the pair's own EstDistance validity interval (est_lo, est_hi] (on both mates' records) */
	uint8_t has_mate;        /* RNEXT "=" + PNEXT + TLEN, else "*\t0\t0" */
	uint8_t flip;            /* the record shows the reverse complement of the read as the caller holds it (mate 2 is held
	                            reverse-complemented, src/GetData.cpp:125-135) */
	uint8_t cigar_len;
	uint8_t rescue;          /* pairs: RescueUnpairedAlignment was due for this pair (its windows depend on min(EstDistance, MaxInsertSize)) */
	char    cigar[KG_ALN_CIGAR_MAX];
	int32_t next;            /* -m: index (into the same array, >= n_reads) of the next record printed for this read, or -1 */
	uint8_t primary;         /* the record of the read's best candidate (iBestAlnCanIdx): the one iPaired / iDistance count (:209-213) */
	uint8_t pad[3];
} kg_aln_record;

/* per 4000-read chunk: what the chunk adds to the run's pairing statistics (iPaired / iDistance, src/Mapping.cpp:209-213),
 * and for which EstDistance values its pairing decisions hold: every "dist < EstiDistance" test of
 * CheckPairedAlignmentCandidates (:372) made under est_distance comes out the same for any value in (lo, hi] */
typedef struct {
	int64_t paired, distance;
	int64_t lo, hi;
	int32_t unmapped, unique;    /* reads with score 0 / MAPQ 60 among the reads decided here */
	int32_t host_pairs;          /* reads handed back (KG_ALN_HOST) */
	int32_t rescue_wanted;       /* pairs for which RescueUnpairedAlignment was due (:559-560): its windows depend on EstDistance too */
} kg_chunk_stats;

/* chunk_off[n_chunks + 1]: read index ranges of the batch's chunks (GetNextChunk: 4000 reads each but possibly the last);
 * chunk_paired[c] != 0: the reads of chunk c are pairs (2q, 2q+1), mate 2 reverse-complemented by the caller.
 * *records points at a library-owned pinned array of n_reads entries (four rotate: valid while the next three batches go
 * through the workspace);
 * chunk_stats[n_chunks] is filled.  KG_ERR_ARG when no chained batch is resident or the chunks do not cover it. */
int  kg_align_batch(kg_workspace *ws, const int64_t *chunk_off, const uint8_t *chunk_paired, int n_chunks,
                    int est_distance, int max_insert, int max_gaps, int multi_hit, int unset_flag,
                    const kg_aln_record **records, kg_chunk_stats *chunk_stats);

/* ---- the per-read report of long reads (-pacbio) on the device ------------------------------------------------------- */
/* What ReadMapping()'s bPacBioData branch does per read between chaining and the SAM text (src/Mapping.cpp:513-530), for the reads
 * of the batch the last kg_seed_batch (KG_MODE_SENSITIVE | KG_INPUT_ASCII) + kg_candidates_batch (pacbio != 0) calls on this
 * workspace left on the device:
 *   RemoveRedundantCandidates (src/Mapping.cpp:317-346), GenMappingReport (src/AlignmentCandidates.cpp:624-745: IdentifyNormalPairs
 *   of the read :420-490, CheckCoordinateValidity :582-610, Process{Head,Normal,Tail}SequencePair src/tools.cpp:225-397 with
 *   GenerateNormalPairAlignment :142-223 and nw_alignment on the device, GenCoordinateInfo / GenerateCIGAR :492-562),
 *   SetSingleAlignmentFlag and EvaluateMAPQ (src/Mapping.cpp:49-70, 160-175).
 * One record per read, what OutputSingledAlignments prints for it (:272-315).  A mapped record's CIGAR does not fit the record:
 * cigar_len is KG_ALN_CIGAR_POOLED and the first 12 bytes of `cigar` hold {int64_t offset; int32_t bytes} into *cigar_pool.
 * A read the kernels do not take (a literal '-' among its characters, a fragment pair outside the fragment kernels' envelope, seeds
 * that CheckOverlappingSeeds leaves out of order) comes back with kind KG_ALN_HOST: the caller maps it with its own implementation of
 * the same reference code (the candidates kg_candidates_batch returned are unmodified).  *records and *cigar_pool are library-owned
 * page-locked arrays; four sets rotate: valid while the next three batches go through the workspace. */
#define KG_ALN_CIGAR_POOLED 255
int  kg_longread_batch(kg_workspace *ws, const kg_aln_record **records, const char **cigar_pool, int64_t *cigar_bytes, int64_t *n_host_reads);
/* Running tallies since the workspace was created: [0] reads through kg_longread_batch, [1] of them handed back (KG_ALN_HOST), and why
 * (candidates): [2] a literal '-' in the read, [3] a fragment pair outside the fragment kernels' envelope, [4] seeds out of order after
 * CheckOverlappingSeeds, [5] element pool full; [6] candidates that went through CheckOverlappingSeeds' sequential form; [8..13] fragment
 * tasks that sent their request back: a side above 8192 bases, a read character other than A/C/G/T, more than 383 exact matches, more normal pairs than
 * the kernel holds, a work list full, recursion deeper than 6.  Diagnostics only. */
int  kg_longread_reasons(kg_workspace *ws, uint64_t out[16]);

/* Running tallies (since the workspace was created) of why read pairs came back as KG_ALN_HOST: [0] candidate product too large,
 * [1] a mate-2 rescue window would be scanned, [2] rescue window too long, [3] mate not plain A/C/G/T or too long for the rescue
 * kernel, [4] too many exact-match runs in a window, [5] rescued candidate with too many pairs, [6] candidate with too many
 * seeds, [7] too many gap pairs, [8] a gap fragment needs the 8-mer partition (both sides > 30), [9] a device list was full,
 * [10] CIGAR too long, [11] score beyond the MAPQ table, [12] read too long.  Diagnostics only. */
int  kg_align_reasons(kg_workspace *ws, uint64_t out[16]);

/* ---- Needleman-Wunsch gap closing ----------------------------------------------------------- */
/* n fragment pairs: frag1 (read side, raw characters) concatenated with offsets off1[n+1], frag2
 * (genome side) with off2[n+1].  For pair i the alignment is returned as ops[ops_off[i] ..
 * ops_off[i]+aln_len[i]) with ops_off[i] = off1[i]+off2[i], one byte per column, left to right:
 *   KG_OP_DIAG both consume a character; KG_OP_GAP1 '-' inserted into frag1 (consumes frag2);
 *   KG_OP_GAP2 '-' inserted into frag2 (consumes frag1).
 * Re-inserting the gaps gives byte-identical strings to nw_alignment()'s in-place result. */
#define KG_OP_DIAG 0
#define KG_OP_GAP1 1
#define KG_OP_GAP2 2
int  kg_nw_batch(kg_index *ix, const char *frag1, const int64_t *off1, const char *frag2,
                 const int64_t *off2, int64_t n, uint8_t *ops, int32_t *aln_len);
int  kg_nw_batch_device(kg_index *ix, const char *d_frag1, const int64_t *d_off1, const char *d_frag2,
                        const int64_t *d_off2, int64_t n, int64_t max_len, uint8_t *d_ops,
                        int32_t *d_aln_len, void *stream);

/* GenerateNormalPairAlignment (src/tools.cpp:142-223) for n fragment pairs: the read fragment i is frag1[off1[i], off1[i+1])
 * (raw characters, host buffer), the genome fragment the glen[i] bases of the indexed text at coordinate gpos[i].  Fragments with
 * both sides above 30 are partitioned at their common 8-mers (shift limit: min(50, 20 % of the longer side) with pacbio != 0,
 * else max_gaps), IdentifyNormalPairs(rLen, gLen, ...) runs on the matches, the pieces between them are aligned by
 * nw_alignment, and with pacbio != 0 a piece with a side above 300 goes through the same procedure again (:197).  The result
 * is the alignment as kg_nw_batch reports one: aln_len[i] op codes at ops[ops_off[i] ...], where ops_off[i] must be the
 * number of columns (rLen + gLen) of the requests before i.  status[i] != 0: the request lies outside the kernels' envelope (a
 * read character other than A/C/G/T, more than 383 matches, a side above 8192) -- the caller plans it itself. */
int  kg_fragments_batch(kg_index *ix, const char *frag1, const int64_t *off1, const int64_t *gpos, const int32_t *glen, int64_t n,
                        int pacbio, int max_gaps, uint8_t *ops, const int64_t *ops_off, int32_t *aln_len, uint8_t *status);

/* ---- FASTQ text in, SAM text out ------------------------------------------------------------------------------------ */
/* The reference's worker takes a chunk of reads from GetNextChunk (src/GetData.cpp:109-143: four getline() calls per record,
 * the name cut out of the header by IdentifyHeaderBegPos / EndPos, mate 2 reverse-complemented), maps it, and prints every
 * record with fprintf (OutputPairedAlignments / OutputSingledAlignments, src/Mapping.cpp:177-315).  A stream does the same for
 * a whole batch on the device, so that the caller only moves bytes: it uploads the text of the input file(s) as it lies there
 * and receives the text of the SAM records.  `lanes` batches are in flight at once (the reference's N worker threads each on
 * their own chunk, src/Mapping.cpp:716-717): every lane owns a workspace, device text windows, page-locked staging and
 * result buffers and a HIP stream; the calls of one lane must not overlap, different lanes may be driven from different threads.
 * Short reads, plain 4-line FASTQ, the Illumina configuration (not -pacbio).
 *
 *   per batch:  fill kg_stream_staging(lane, f) -> kg_stream_upload (any number of pieces) -> kg_stream_parse -> kg_stream_map
 */
typedef struct kg_stream kg_stream;
typedef struct {
	int64_t max_reads;       /* reads per batch the lanes are sized for */
	int64_t max_window;      /* bytes of FASTQ text per file and batch (the staging buffers' size) */
	int32_t lanes;           /* batches in flight */
	int32_t seed_group;      /* > 1: ONE seeding launch (FM-index search, locate, sort) over the parsed batches of this many lanes -- lanes
	                            [g * seed_group, (g + 1) * seed_group) form group g; must divide `lanes`, at most 8.  A search launch costs
	                            ~0.65 ms + 0.52 ms per M reads, so four lanes' 1 M-read batches in one launch take what 1.6 of them take
	                            alone; chaining, the report and the text keep the lanes' granularity (the reference's N workers on N
	                            chunks, src/Mapping.cpp:716-717).  0 / 1: every lane seeds its own batch */
} kg_stream_config;
int   kg_stream_open(kg_index *ix, const kg_stream_config *cfg, kg_stream **out);
void  kg_stream_close(kg_stream *s);
/* page-locked staging buffer of input file `file` (0 / 1) in lane `lane`, *capacity = max_window bytes */
char *kg_stream_staging(kg_stream *s, int lane, int file, int64_t *capacity);
/* staging[file][from, to) -> the lane's device window, asynchronously: a caller reads the next piece meanwhile */
int   kg_stream_upload(kg_stream *s, int lane, int file, int64_t from, int64_t to);

typedef struct {
	int64_t begin[2], end[2];    /* the window of file f is staging[f][begin, end); begin lies on a record boundary */
	int32_t eof[2];              /* the window ends where the file ends */
	int32_t two_files;           /* -f / -f2: read 2q comes from file 0, read 2q+1 from file 1; else every read from file 0 */
	int32_t paired;              /* the second read of every pair is held reverse-complemented, qualities reversed (src/GetData.cpp:125-135) */
	int32_t chunk_reads;         /* ReadChunkSize (src/structure.h:21): 4000 */
	int32_t gz_lines;            /* != 0: the text was inflated from a gz file, which the reference reads through gzgets() with a 1000-byte buffer
	                                (src/GetData.cpp:145-219): a record with a line of more than 999 bytes (its newline included), or whose header line
	                                does not start with '@' / '>' or names nothing (:162), is read differently there -- it ends the batch
	                                (KG_STREAM_STOP_IRREGULAR) and the caller's gz reader continues in front of it */
	int64_t want_reads;          /* take at most this many reads (a multiple of chunk_reads, <= max_reads) */
} kg_stream_window;
#define KG_STREAM_STOP_NONE      0
#define KG_STREAM_STOP_IRREGULAR 1   /* behind the reads taken lies a record the device parser leaves to the caller's own reader: an
                                        empty read (it ends a chunk early, src/GetData.cpp:117,121), a NUL byte, an overlong header */
#define KG_STREAM_STOP_TAIL      2   /* the file ends in less than a chunk that is not regular (a lone mate, a partial record) */
typedef struct {
	int64_t n_reads, n_chunks, n_bases;   /* whole chunks of chunk_reads reads, or everything up to a regular end of the file(s) */
	int64_t used[2];                       /* the next window of file f starts at staging offset used[f] of this one */
	int32_t stop;                          /* KG_STREAM_STOP_*: != NONE: after this batch the caller's own reader continues at used[] */
	int32_t done;                          /* the files ended regularly with this batch */
} kg_stream_parsed;
/* GetNextChunk for the whole window: lines, records, the reads as the reference holds them (left on the device for kg_stream_map) */
int   kg_stream_parse(kg_stream *s, int lane, const kg_stream_window *w, kg_stream_parsed *out);

typedef struct {
	int32_t est_distance, max_insert, max_gaps, multi_hit, unset_flag;   /* as kg_align_batch */
	int32_t fetch_all;           /* != 0: records / cand_off / cands / cand_seeds of the result hold the whole batch; 0: they hold what kg_stream_fetch
	                                brought over, and (multi_hit) the chained extra records -- the text, its offsets, the chunk statistics and the list
	                                of reads handed back always come whole (a run touches the records and candidates of ~2 % of its chunks) */
} kg_stream_params;
typedef struct {
	int64_t n_reads, n_chunks;
	const char *sam;                 /* the SAM lines of the reads decided on the device, in read order (page-locked, the lane's) */
	int64_t sam_bytes;
	const int64_t *sam_off;          /* [n_reads + 1]: the lines of read r are sam[sam_off[r], sam_off[r+1]) -- empty for a read handed back */
	const kg_aln_record *records;    /* [n_records] as kg_align_batch (the first n_reads are the reads' own) */
	int64_t n_records;
	const kg_chunk_stats *chunk_stats;   /* [n_chunks] */
	const int32_t *host_reads;       /* reads whose record is KG_ALN_HOST, ascending */
	int64_t n_host_reads;
	const int64_t *cand_off;         /* [n_reads + 1] the candidates of read r: cands[cand_off[r], cand_off[r+1]) (kg_candidates_batch) */
	const kg_candidate *cands;
	const kg_seed *cand_seeds;
	const uint32_t *rec_start[2];    /* staging offset of the header line of record j of file f (read r = record r/2 of file r%2 with two files) */
} kg_stream_result;
/* seeding (FastMode), chaining, the per-read report (kg_align_batch) and the SAM text for the batch kg_stream_parse left in the
 * lane.  The result's arrays belong to the lane: valid until its next kg_stream_parse. */
int   kg_stream_map(kg_stream *s, int lane, const kg_stream_params *prm, kg_stream_result *out);
/* The records, candidate offsets, candidates and candidate seeds of reads [first, first + count) of the batch the lane's last kg_stream_map call mapped
 * -- still resident on the device until the lane's next kg_stream_parse -- into the arrays that call's result points at, at their own places
 * (records[first ..], cand_off[first .. first + count], cands[cand_off[first] ..], ...).  What a caller needs for the reads the device handed back
 * (KG_ALN_HOST: their candidates) and for a chunk it maps again under another EstDistance (the pairs' validity intervals in the records).  Blocks
 * until the data is there; may be called from any thread while no other call is using the lane. */
int   kg_stream_fetch(kg_stream *s, int lane, int64_t first, int64_t count);
/* Seeding groups (seed_group > 1) work in ROUNDS: in every round each lane of a group either calls kg_stream_map with a parsed batch --
 * the call returns when the round's one seeding launch is done -- or is absent.  rounds > 0: lane `lane` has no batch for that many
 * rounds; < 0: until further notice (its input has ended); 0: it takes part again (call it for every lane before a run; also clears an
 * abort).  kg_stream_group_abort wakes every lane that waits for its group with an error (a caller's failure path). */
int   kg_stream_group_absent(kg_stream *s, int lane, int rounds);
int   kg_stream_group_abort(kg_stream *s);
/* test aid: the reads of the parsed batch as the seeding stage sees them (characters, offsets[n_reads + 1]); enc may be NULL */
int   kg_stream_fetch_reads(kg_stream *s, int lane, uint8_t *enc, int64_t *read_off);

/* device time since the stream was opened (or last reset), summed over the batches of all lanes: HIP events on each lane's
 * stream around its stages, the search kernel's own launches, and the bytes that kernel fetched (kg_workspace_traffic's formula) */
typedef struct {
	int64_t batches, reads;
	double parse_ms, seed_ms, chain_ms, align_ms, format_ms, copy_ms;
	double search_kernel_ms;
	int64_t search_kernel_launches;
	double search_useful_bytes;
	double text_in_bytes, text_out_bytes;
	double candidates, candidate_seeds;     /* chained candidates and their seeds of the batches (what the alignment stage reads per candidate) */
	/* the batches' kernels one by one (HIP events around each launch on the lane's stream; other lanes' kernels share the device):
	 * [0] chain, [1] aln_pair, [2] aln_rescue + post_rescue, [3] aln_plan_fast, [4] aln_plan, [5] aln_partition, [6] the NW kernels,
	 * [7] aln_finish, [8] aln_final, [9] sam_size + scan, [10] sam_format, [11] fq_count / index / record / plan, [12] fq_materialise,
	 * [13] locate + sort, [14] aln_trivial */
	double kernel_ms[16];
	int64_t kernel_launches[16];
	/* what the alignment stage's lists held, summed over the batches: [0] candidates parked for NW, [1] NW jobs, [2] bytes of their op strings,
	 * [3] fragment pairs through the 8-mer partition, [4] rescue windows, [5] partition plans, [6] pairs aln_trivial decided start to finish, [7] candidates aln_plan_fast left to aln_plan */
	double aln_counts[8];
	/* KG_STREAM_CHECKSUM set (a measurement aid): the SAM text of the batches summed on the device -- [0] the sum of its bytes, [1] its line feeds
	 * (exact: both stay far below 2^53) -- so that a run whose output is never copied into file pages still names the text it made; else 0 */
	double text_checksum[2];
} kg_stream_timing_t;
int   kg_stream_timing(kg_stream *s, kg_stream_timing_t *out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* KART_AMD_H */
