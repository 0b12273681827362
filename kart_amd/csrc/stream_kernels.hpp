// stream_kernels.hpp -- argument blocks of the FASTQ-text-in / SAM-text-out kernels (stream_kernels.hip) shared with the C ABI.
//
// The reference parses its input one record at a time on the host (GetNextEntry / GetNextChunk, src/GetData.cpp:51-143) and prints
// its output one record at a time (OutputPairedAlignments / OutputSingledAlignments, src/Mapping.cpp:177-315).  Here both ends run
// on the device, so that the host only moves bytes: the FASTQ text of a batch is uploaded as it lies in the file, the device finds
// the lines and the records, builds the read characters the seeding stage wants (mate 2 reverse-complemented), and after the
// alignment stage formats every kg_aln_record as the SAM line(s) the reference would print.
#pragma once
#include "seed_kernels.hpp"

namespace kg {

constexpr int kFqTile = 4096;            // bytes of text per block of the line-index kernels (256 threads x 16 bytes)

// why the parser stopped in front of the end of its window (kg_stream_parsed::stop)
enum { FQ_STOP_NONE = 0, FQ_STOP_IRREGULAR = 1, FQ_STOP_TAIL = 2 };

// words of the parser's control block (device, int64)
enum {
	FQM_LINES0 = 0, FQM_LINES1 = 1,      // lines found in window 0 / 1 (a last line without newline counts)
	FQM_BAD0 = 2, FQM_BAD1 = 3,          // first record of window f the device path does not take (empty read, overlong header), or INT64_MAX
	FQM_NUL = 4,                         // a NUL byte somewhere in a window (the reference treats lines as C strings)
	FQM_OVERFLOW = 5,                    // more lines than the line table holds
	FQM_READS = 6, FQM_CHUNKS = 7, FQM_BASES = 8, FQM_USED0 = 9, FQM_USED1 = 10, FQM_STOP = 11, FQM_DONE = 12,
	FQM_WORDS = 16
};

struct FqWindow {
	const uint8_t *text;       // device buffer; the window is text[begin, end)
	int64_t begin, end;
	int eof;                   // the window ends where the file ends
	int32_t *tile_lines;       // [n_tiles + 1] newlines per tile, then (in place) their exclusive scan
	uint32_t *line_end;        // [line_capacity] offset (in `text`) one past each line
	int64_t line_capacity;
	// per record (four lines), capacity line_capacity / 4
	uint32_t *rec_hdr;         // offset of the header line
	uint32_t *rec_name;        // name: (offset from the header line's start) | (length << 16)   (IdentifyHeaderBegPos / EndPos)
	uint32_t *rec_seq;         // offset of the sequence line
	uint32_t *rec_qual;        // offset of the quality line
	int32_t *rec_rlen;         // sequence line length - 1
	int32_t *rec_qlen;         // min(quality line length, rlen)
};

struct FqArgs {
	FqWindow w[2];
	int two_files;             // mates alternate between the windows (else every read comes from window 0)
	int paired;                // the second read of every pair is held reverse-complemented (src/GetData.cpp:125-135)
	int chunk_reads;           // ReadChunkSize (4000)
	int gz_lines;              // the text comes out of a gz file: records that gzgets() with its 1000-byte buffer reads differently end the batch (kg_stream_window)
	int64_t max_reads;         // capacity of the batch (a multiple of chunk_reads)
	int64_t want_reads;        // take at most this many (a multiple of chunk_reads, <= max_reads)
	int64_t *meta;             // [FQM_WORDS]
	int32_t *read_len;         // [max_reads + 1]
	int64_t *read_off;         // [max_reads + 1] exclusive scan of read_len
	uint8_t *enc;              // the reads' characters as the reference holds them
	int64_t n_reads;           // (materialise) reads of the batch, as the plan settled it
};

// the SAM text of a batch
struct SamArgs {
	FqWindow w[2];
	int two_files, paired;
	const uint8_t *enc;
	const int64_t *read_off;
	int64_t n_reads;
	const kg_aln_record *records;    // [n_reads + extras]
	const uint8_t *chr_names;        // contig names, concatenated ...
	const int32_t *chr_name_off;     // ... [n_chr + 1]
	int32_t *sam_len;                // [n_reads + 1] bytes of text per read (0 for reads handed back)
	int64_t *sam_off;                // [n_reads + 1] exclusive scan
	uint8_t *sam;                    // the text
	int64_t sam_capacity;
	int32_t *host_list;              // reads handed back (KG_ALN_HOST), in no particular order
	unsigned long long *ctl;         // [0] entries of host_list, [1] format errors (a record whose text is not the size announced), [2] / [3] sam_checksum_kernel's sums
};

size_t fq_scan_temp_bytes(int64_t max_items);
// line index + record table of both windows, read lengths and their scan, the plan (meta)
hipError_t launch_fq_parse(const FqArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream);
hipError_t launch_fq_materialise(const FqArgs &a, int n_cu, hipStream_t stream);
hipError_t launch_sam_size(const SamArgs &a, void *scan_temp, size_t scan_temp_bytes, int n_cu, hipStream_t stream);
hipError_t launch_sam_format(const SamArgs &a, int n_cu, hipStream_t stream);
hipError_t launch_sam_checksum(const SamArgs &a, int n_cu, hipStream_t stream);      // measurement aid: ctl[2] += byte sum, ctl[3] += line feeds of the text
// grouped seeding: a lane's parsed batch published as a segment of its group's batch; the lane's seed offsets cut out of the group's
hipError_t launch_group_publish(const int64_t *local_off, int64_t n, int64_t slots, int64_t enc_base, int64_t *g_off, int32_t *g_len, int n_cu, hipStream_t stream);
hipError_t launch_group_rebase(const int64_t *g_seed_off, int64_t n, int64_t first, int64_t *seed_off, int n_cu, hipStream_t stream);

}  // namespace kg
