/* oracle/kart_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C API of the CPU restatement of Kart's seed-and-extend hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (kart_amd/) never links or calls it.
 *
 * Pinned against the unmodified reference (oracle/_ref/libkartref.so) by
 * oracle/pin_against_ref.py and by the committed fixtures in tests/golden/.
 */
#ifndef KART_ORACLE_H
#define KART_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ko_index ko_index;

/* one exact-match seed ("simple pair"), reference SeedPair_t (src/structure.h:106-114)
 * reduced to the fields that are not derivable: PosDiff = gPos - rPos, gLen = rLen. */
typedef struct {
	int64_t gPos;
	int32_t rPos;
	int32_t len;
} ko_seed;

/* full SeedPair_t image used by the chaining restatement */
typedef struct {
	int64_t gPos;
	int64_t PosDiff;
	int32_t rPos;
	int32_t rLen;
	int32_t gLen;
	int32_t bSimple;
} ko_pair;

/* work counters in the units of SURVEY.md section 8(d) */
typedef struct {
	uint64_t searches;   /* BWT_Search calls */
	uint64_t lf1;        /* extension steps whose two ranks share one Occ block */
	uint64_t lf2;        /* extension steps touching two Occ blocks */
	uint64_t inv;        /* bwt_invPsi steps during SA recovery */
	uint64_t sa;         /* located hits (bwt_sa calls) */
	uint64_t seeds;      /* emitted seeds */
	uint64_t bases;      /* read bases consumed */
} ko_counters;

ko_index *ko_index_load(const char *prefix);
void      ko_index_free(ko_index *ix);
int64_t   ko_genome_size(const ko_index *ix);      /* l_pac */
uint64_t  ko_seq_len(const ko_index *ix);          /* 2 * l_pac */
uint64_t  ko_primary(const ko_index *ix);
int       ko_n_contigs(const ko_index *ix);
int       ko_min_seed_len(const ko_index *ix);     /* Mapping.cpp:645 */
const char *ko_ref_sequence(const ko_index *ix);   /* char[2L], fwd + revcomp */

/* L1 primitives */
uint64_t ko_occ(const ko_index *ix, uint64_t k, int c);
void     ko_occ4(const ko_index *ix, uint64_t k, uint64_t cnt[4]);
uint64_t ko_sa(const ko_index *ix, uint64_t k);
/* BWT_Search: returns freq (0 if none), *len always set; locs needs room for 50 */
int ko_bwt_search(const ko_index *ix, const uint8_t *seq, int start, int stop, int min_seed_len,
                  int *len, uint64_t *locs);

/* seeding: mode 0 = FastMode, 1 = SensitiveMode.  Returns number of seeds
 * (sorted with the mode's comparator) or -(needed) if cap is too small. */
int ko_seed_read(const ko_index *ix, int mode, int min_seed_len, const uint8_t *enc, int rlen,
                 ko_seed *out, int cap);
/* batch form used as the CPU baseline: reads concatenated, offsets[n+1].  Returns total seeds;
 * seed_offsets[n+1] filled.  threads >= 1 (std::thread fan-out over read ranges). */
int64_t ko_seed_batch(const ko_index *ix, int mode, int min_seed_len, const uint8_t *enc,
                      const int64_t *offsets, int64_t n_reads, int64_t *seed_offsets,
                      ko_seed *out, int64_t cap, int threads);
void ko_counters_get(ko_counters *c);
void ko_counters_reset(void);

/* nw_alignment: s1 (length m) vs s2 (length n), raw characters.  out1/out2 need m+n+1 bytes.
 * Returns aligned length. */
int ko_nw(const char *s1, int m, const char *s2, int n, char *out1, char *out2);

/* GenerateNormalPairAlignment (src/tools.cpp:142-223): 8-mer partition within the shift limit (pacbio: min(50, 20 % of the longer
 * side), else max_gaps), IdentifyNormalPairs on the fragment, nw_alignment of the pieces, itself again for pacbio pieces above 300.
 * out1 / out2 need m + n + 1 bytes.  Returns the aligned length. */
int ko_normal_pair_alignment(int pacbio, int max_gaps, const char *s1, int m, const char *s2, int n, char *out1, char *out2);

/* chaining */
int64_t ko_alignment_boundary(const ko_index *ix, int64_t gPos);
/* Illumina: cand_off[ncand+1] index into out_pairs; score[ncand]; posdiff[ncand].  Returns ncand. */
int ko_candidates_illumina(const ko_index *ix, int rlen, int max_gaps, const ko_seed *seeds, int n,
                           int *cand_off, int *score, int64_t *posdiff, ko_pair *out_pairs,
                           int cand_cap, int pair_cap);
int ko_candidates_pacbio(const ko_index *ix, int rlen, const ko_seed *seeds, int n,
                         int *cand_off, int *score, int64_t *posdiff, ko_pair *out_pairs,
                         int cand_cap, int pair_cap);
/* IdentifyNormalPairs on one candidate's seed vector (in/out). Returns new count. */
int ko_identify_normal_pairs(int rlen, int glen, ko_pair *pairs, int n, int cap);

#ifdef __cplusplus
}
#endif
#endif
