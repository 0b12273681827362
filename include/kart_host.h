/* kart_host.h -- C ABI of the host pipeline (libkart_host.so): the reference's driver as a library.
 *
 * The reference is a program, not a library: main() loads the index once (bwa_idx_load + RestoreReferenceInfo,
 * src/main.cpp:192-207) and calls Mapping() (src/Mapping.cpp:639-742), which maps every input library and writes SAM/BAM.
 * These entry points are that split made explicit, so that a caller (bench.py, a service, another language over FFI) can
 * keep the index resident in HBM and map several inputs against it:
 *
 *   reference (file:line)                                        here
 *   -----------------------------------------------------------  ------------------------------------------
 *   bwa_idx_load + RestoreReferenceInfo (src/main.cpp:192-207)    kh_open
 *   Mapping() with the globals main() parsed (src/main.cpp:123-176, src/Mapping.cpp:639-742)
 *                                                                kh_map (argv = the reference's own flags)
 *   exit                                                         kh_close
 *
 * kart_amd/bin/kart-amd is this library's only other client.  The kernels are reached through libkart_amd.so
 * (include/kart_amd.h); there is no CPU path: kh_open fails without a usable HIP device.
 */
#ifndef KART_HOST_H
#define KART_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kh_session kh_session;

/* end-of-run statistics of Mapping() (src/Mapping.cpp:730-741) plus the pipeline's own timers */
typedef struct {
	int64_t total_reads;     /* iTotalReadNum */
	int64_t unmapped;        /* iUnMapping */
	int64_t unique;          /* reads with MAPQ 60 */
	int64_t paired;          /* iPaired */
	int64_t distance;        /* iDistance */
	int64_t respeculated;    /* chunks re-mapped because a speculated EstDistance did not hold */
	double  map_seconds;     /* first read in -> last output byte written (index load excluded) */
	int32_t sharded;         /* 1: the counts are this process's shard of a -shard r/N run */
	int32_t pad;
	/* the run's batches on the device (kg_stream_timing_t, include/kart_amd.h): HIP events on the lanes' streams, summed */
	int64_t stream_reads;    /* reads that went through the device's FASTQ-in / SAM-out stream */
	int64_t stream_batches;
	double  stage_ms[6];     /* parse, seed, chain, align, format, copy-out */
	double  search_kernel_ms;        /* search_kernel's own launches (events around each) */
	int64_t search_kernel_launches;
	double  search_useful_bytes;     /* bytes the implemented search fetched in them (kg_traffic_t's formula) */
	double  text_in_bytes, text_out_bytes;
	double  candidates, candidate_seeds;     /* of the batches that went through the device stream (kg_stream_timing_t) */
	double  kernel_ms[16];           /* the batches' kernels one by one (kg_stream_timing_t::kernel_ms: chain, aln_pair, aln_rescue, aln_plan_fast, ...) */
	int64_t kernel_launches[16];
	double  aln_counts[8];           /* kg_stream_timing_t::aln_counts */
	/* what the lanes' host threads spent their time on, seconds summed over the `lanes` lane threads of the run's device stream:
	 * [0] waiting for the writer to have copied the lane's previous text out (the output side holds the lane back), [1] reading + uploading
	 * the input block, [2] waiting for the batch before to be parsed, [3] the parse call, [4] the map call (the lane's batch on the device,
	 * its copies back included), [5] the reads handed back to the host */
	double  lane_seconds[6];
	int32_t lanes;
	int32_t pad2;
	double  text_checksum[2];        /* kg_stream_timing_t::text_checksum (KG_STREAM_CHECKSUM) */
} kh_stats_t;

const char *kh_last_error(void);

/* Loads <prefix>.{bwt,sa,pac,ann,amb} onto HIP device `device` (full suffix array, 2-bit text, q-mer table) and the
 * host-side reference; `threads` = worker threads of later kh_map calls (the reference's -t). */
int  kh_open(const char *index_prefix, int device, int threads, kh_session **out);

/* One mapping run.  argv holds the reference's command-line flags for the run (src/main.cpp:123-176): -f <files> [-f2
 * <files>] -o|-bo <out> [-m] [-p] [-pacbio] [-g INT] [-silent], plus this pipeline's -shard r/N -rendezvous FILE; -i, -t and
 * -gpu are fixed by the session.  Returns 0 on success. */
int  kh_map(kh_session *s, int argc, const char *const *argv, kh_stats_t *stats);
/* (measurement aid: with KART_AMD_OUTPUT_NULL=1 in the environment of a kh_map call the text goes to /dev/null instead of the file named by
 *  -o -- everything up to the host's copy into the file's pages runs as usual; bench.py's gpu_pipeline leg) */

void kh_close(kh_session *s);

#ifdef __cplusplus
}
#endif
#endif /* KART_HOST_H */
