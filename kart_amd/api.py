"""Host-side mirror of Kart's operator interface for the seed-and-extend path, bound to
libkart_amd.so (HIP kernels for gfx950) through its C ABI (include/kart_amd.h) with ctypes.

The names follow the reference's extern prototypes (reference src/structure.h:177-229):

    reference (one read per call)                  here (one batch per call)
    ---------------------------------------------  -----------------------------------------
    bwa_idx_load / RestoreReferenceInfo            Index(prefix, device, sa_mode)
    IdentifySeedPairs_FastMode(rlen, EncodeSeq)    Index.IdentifySeedPairs_FastMode(reads)
    IdentifySeedPairs_SensitiveMode(...)           Index.IdentifySeedPairs_SensitiveMode(reads)
    nw_alignment(m, s1, n, s2)  (in-place)         Index.nw_alignment(pairs) -> gapped strings

There is no CPU fallback here: if the shared library or a GPU is missing the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkart_amd.so")

KG_OK = 0
KG_MODE_FAST, KG_MODE_SENSITIVE = 0, 1
KG_INPUT_ASCII = 0x100   # OR into mode: reads are given as characters, encoded on the device
KG_SA_SAMPLED, KG_SA_FULL = 0, 1
KG_SA_AUTO = -1                        # full below 2^32 text symbols; above: wide where the device has room, else compact (the host pipeline's default)
KG_SA_FULL40 = 5                       # the compact index: 5-byte suffix-array entries, a quarter of the q-mer table, no triple planes
KG_SA_FULL40_WIDE = 6                  # 5-byte suffix-array entries with the full q-mer table and the triple planes
KG_SA_DENSE4, KG_SA_DENSE8 = 4, 8       # the smaller index: every 4th / 8th suffix-array entry resident, no triple planes
KG_OCC_THR_DEFAULT = 50
KG_OP_DIAG, KG_OP_GAP1, KG_OP_GAP2 = 0, 1, 2

SEED_DT = np.dtype([("gPos", "<i8"), ("rPos", "<i4"), ("len", "<i4")])
CAND_DT = np.dtype([("posDiff", "<i8"), ("score", "<i4"), ("count", "<i4"), ("first", "<i8")])

# every symbol include/kart_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "kg_last_error", "kg_device_count", "kg_index_load", "kg_index_destroy", "kg_index_info",
    "kg_index_contig", "kg_host_alloc", "kg_host_free", "kg_rank_sa_batch", "kg_workspace_create", "kg_workspace_destroy", "kg_workspace_counters", "kg_workspace_traffic",
    "kg_workspace_overflow", "kg_workspace_segment_fallbacks", "kg_workspace_set_profiling", "kg_workspace_set_single_steps", "kg_index_selfcheck", "kg_workspace_kernel_ms", "kg_seed_batch", "kg_candidates_batch", "kg_align_batch", "kg_align_reasons", "kg_seed_batch_device", "kg_nw_batch", "kg_nw_batch_device",
    "kg_fragments_batch", "kg_longread_batch", "kg_longread_reasons",
    "kg_stream_open", "kg_stream_close", "kg_stream_staging", "kg_stream_upload", "kg_stream_parse", "kg_stream_map", "kg_stream_fetch", "kg_stream_fetch_reads", "kg_stream_timing",
    "kg_stream_group_absent", "kg_stream_group_abort",
)


HOST_LIB_PATH = os.path.join(_HERE, "libkart_host.so")
HOST_ABI_SYMBOLS = ("kh_last_error", "kh_open", "kh_map", "kh_close")      # every symbol include/kart_host.h declares


class KartAmdError(RuntimeError):
    pass


class IndexInfo(C.Structure):
    _fields_ = [("genome_size", C.c_int64), ("seq_len", C.c_uint64), ("primary", C.c_uint64),
                ("n_contigs", C.c_int32), ("min_seed_len", C.c_int32), ("sa_mode", C.c_int32),
                ("device", C.c_int32), ("device_bytes", C.c_uint64)]


class Contig(C.Structure):
    _fields_ = [("name", C.c_char_p), ("fwd_start", C.c_int64), ("rev_start", C.c_int64), ("len", C.c_int64)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("searches", "lf1", "lf2", "inv", "sa", "seeds", "bases")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    def algorithmic_bytes(self) -> int:
        """bytes_seed of SURVEY.md section 8(d)."""
        return 64 * (self.lf1 + 2 * self.lf2 + self.inv) + 8 * self.sa + self.bases + 16 * self.seeds


class Traffic(C.Structure):
    """kg_traffic_t: what the search kernel itself fetched in the last batch (the implemented algorithm)."""
    _fields_ = [(n, C.c_uint64) for n in ("table_lookups", "rank_steps", "rank_steps_two_lines", "sa_gathers", "text_rounds", "window_words",
                                          "hits", "searches", "sa_entry_bytes", "rank_steps_two_lines_narrow", "double_steps", "double_steps_two_lines",
                                          "triple_steps", "double_step_bytes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    def useful_bytes(self, n_reads: int) -> int:
        """bytes the implemented search needs per launch (see include/kart_amd.h, kg_traffic_t)"""
        return (8 * self.table_lookups + 32 * self.rank_steps + self.double_step_bytes + self.sa_entry_bytes * self.sa_gathers + 48 * self.text_rounds
                + 8 * self.window_words + 20 * n_reads + 32 * self.hits)

    def min_lines(self, n_reads: int) -> int:
        """128-byte lines a launch cannot avoid touching with this layout: one per table lookup / SA gather, one or two per rank
        single step, one or two per double step, the text and read lines of a comparison round (a 16-byte text window straddles a line boundary 1 time in 8, a
        32-byte read window 1 in 4), the hit records and per-read words (dense)"""
        return int(self.table_lookups + self.rank_steps + self.rank_steps_two_lines + self.double_steps + self.double_steps_two_lines + self.sa_gathers + self.text_rounds * (1.125 + 1.25)
                   + (8 * self.window_words + 20 * n_reads + 32 * self.hits) / 128)


class StreamConfig(C.Structure):
    _fields_ = [("max_reads", C.c_int64), ("max_window", C.c_int64), ("lanes", C.c_int32), ("seed_group", C.c_int32)]


class StreamWindow(C.Structure):
    _fields_ = [("begin", C.c_int64 * 2), ("end", C.c_int64 * 2), ("eof", C.c_int32 * 2), ("two_files", C.c_int32), ("paired", C.c_int32),
                ("chunk_reads", C.c_int32), ("gz_lines", C.c_int32), ("want_reads", C.c_int64)]


class StreamParsed(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_chunks", C.c_int64), ("n_bases", C.c_int64), ("used", C.c_int64 * 2), ("stop", C.c_int32), ("done", C.c_int32)]


class StreamParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("est_distance", "max_insert", "max_gaps", "multi_hit", "unset_flag", "fetch_all")]


class StreamResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_chunks", C.c_int64), ("sam", C.c_void_p), ("sam_bytes", C.c_int64), ("sam_off", C.POINTER(C.c_int64)),
                ("records", C.c_void_p), ("n_records", C.c_int64), ("chunk_stats", C.c_void_p), ("host_reads", C.POINTER(C.c_int32)), ("n_host_reads", C.c_int64),
                ("cand_off", C.POINTER(C.c_int64)), ("cands", C.c_void_p), ("cand_seeds", C.c_void_p), ("rec_start", C.POINTER(C.c_uint32) * 2)]


class StreamTiming(C.Structure):
    _fields_ = [("batches", C.c_int64), ("reads", C.c_int64)] + [(n, C.c_double) for n in ("parse_ms", "seed_ms", "chain_ms", "align_ms", "format_ms", "copy_ms", "search_kernel_ms")] + \
               [("search_kernel_launches", C.c_int64)] + [(n, C.c_double) for n in ("search_useful_bytes", "text_in_bytes", "text_out_bytes", "candidates", "candidate_seeds")] + \
               [("kernel_ms", C.c_double * 16), ("kernel_launches", C.c_int64 * 16), ("aln_counts", C.c_double * 8), ("text_checksum", C.c_double * 2)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n.startswith(("kernel_", "aln_counts")) else getattr(self, n)) for n, _ in self._fields_}


_lib = None


def _check_one_hip_runtime(L):
    """Two copies of libamdhip64 in one process (PyTorch's and /opt/rocm's) each own a HIP runtime, and the one that touches the device second
    finds none: kg_device_count() then returns 0 with no hint why.  Say so at load time -- only when that is what happened."""
    try:
        paths = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
    except OSError:
        return
    if len({os.path.realpath(p_) for p_ in paths}) > 1 and "torch" in sys.modules and L.kg_device_count() <= 0:
        raise KartAmdError("two HIP runtimes are loaded in this process (%s): import torch BEFORE kart_amd.api.load_library(), or set KART_AMD_NO_TORCH=1 and do "
                           "not import torch afterwards" % ", ".join(sorted(paths)))


def load_library() -> C.CDLL:
    """Load libkart_amd.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KartAmdError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    # ONE HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's); if
    # this library is loaded first it pulls in /opt/rocm's copies, a later `import torch` adds its own, and whichever of the two
    # runtimes touches the device second finds none (measured on the GPU box: kg_device_count() == 0 while torch computes,
    # tools/probe_fork_hip.py).  Loaded after torch, the library resolves to torch's copies and both share the device.  A process
    # that has PyTorch installed and may use it (index_build, bench.py, the tests) therefore imports it before this library;
    # KART_AMD_NO_TORCH=1 skips that for a torch-free embedding.
    if "torch" not in sys.modules and not os.environ.get("KART_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:          # (not installed, or installed and broken -- an OSError from a missing shared object, ...: the library then brings /opt/rocm's runtime)
            pass
    L = C.CDLL(LIB_PATH)
    _check_one_hip_runtime(L)
    L.kg_last_error.restype = C.c_char_p
    L.kg_index_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.kg_index_destroy.argtypes = [C.c_void_p]
    L.kg_index_destroy.restype = None
    L.kg_index_info.argtypes = [C.c_void_p, C.POINTER(IndexInfo)]
    L.kg_index_contig.argtypes = [C.c_void_p, C.c_int, C.POINTER(Contig)]
    L.kg_rank_sa_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kg_workspace_create.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]
    L.kg_workspace_destroy.argtypes = [C.c_void_p]
    L.kg_workspace_destroy.restype = None
    L.kg_workspace_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
    L.kg_workspace_traffic.argtypes = [C.c_void_p, C.POINTER(Traffic)]
    L.kg_workspace_overflow.argtypes = [C.c_void_p]
    L.kg_workspace_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.kg_workspace_set_single_steps.argtypes = [C.c_void_p, C.c_int]
    L.kg_index_selfcheck.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.kg_workspace_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float * 4)]
    L.kg_workspace_overflow.restype = C.c_int64
    L.kg_workspace_segment_fallbacks.argtypes = [C.c_void_p]
    L.kg_workspace_segment_fallbacks.restype = C.c_int64
    L.kg_seed_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_void_p, C.POINTER(C.c_void_p)]
    L.kg_seed_batch_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.kg_candidates_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.kg_nw_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.kg_nw_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
    L.kg_fragments_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kg_stream_open.argtypes = [C.c_void_p, C.POINTER(StreamConfig), C.POINTER(C.c_void_p)]
    L.kg_stream_close.argtypes = [C.c_void_p]
    L.kg_stream_close.restype = None
    L.kg_stream_staging.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    L.kg_stream_staging.restype = C.c_void_p
    L.kg_stream_upload.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64]
    L.kg_stream_parse.argtypes = [C.c_void_p, C.c_int, C.POINTER(StreamWindow), C.POINTER(StreamParsed)]
    L.kg_stream_map.argtypes = [C.c_void_p, C.c_int, C.POINTER(StreamParams), C.POINTER(StreamResult)]
    L.kg_stream_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64]
    L.kg_stream_fetch_reads.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.kg_stream_group_absent.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kg_stream_group_abort.argtypes = [C.c_void_p]
    L.kg_stream_timing.argtypes = [C.c_void_p, C.POINTER(StreamTiming), C.c_int]
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != KG_OK:
        raise KartAmdError(f"{what} failed (status {rc}): {load_library().kg_last_error().decode()}")


def device_count() -> int:
    return int(load_library().kg_device_count())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def concat_reads(reads):
    """list of uint8 code arrays -> (enc, offsets)."""
    lens = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
    offsets = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    enc = np.concatenate(reads).astype(np.uint8, copy=False) if len(reads) else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(enc), offsets


class Workspace:
    def __init__(self, index: "Index", max_reads: int, max_bases: int):
        self.index = index
        self.lib = index.lib
        h = C.c_void_p()
        _check(self.lib.kg_workspace_create(index.h, max_reads, max_bases, C.byref(h)), "kg_workspace_create")
        self.h = h
        self.max_reads, self.max_bases = max_reads, max_bases

    def close(self):
        if self.h:
            self.lib.kg_workspace_destroy(self.h)
            self.h = None

    def counters(self) -> Counters:
        c = Counters()
        _check(self.lib.kg_workspace_counters(self.h, C.byref(c)), "kg_workspace_counters")
        return c

    def traffic(self) -> Traffic:
        t = Traffic()
        _check(self.lib.kg_workspace_traffic(self.h, C.byref(t)), "kg_workspace_traffic")
        return t

    def overflow(self) -> int:
        return int(self.lib.kg_workspace_overflow(self.h))

    def segment_fallbacks(self) -> int:
        """batches of long reads re-seeded with one walk per read because the segment walks outgrew the hit list"""
        return int(self.lib.kg_workspace_segment_fallbacks(self.h))

    def set_single_steps(self, enabled: bool = True):
        """single extension steps only: the reference's lf1 / lf2 block accounting is then exact (kg_workspace_set_single_steps)"""
        _check(self.lib.kg_workspace_set_single_steps(self.h, int(enabled)), "kg_workspace_set_single_steps")

    def set_profiling(self, enabled: bool = True):
        _check(self.lib.kg_workspace_set_profiling(self.h, int(enabled)), "kg_workspace_set_profiling")

    def kernel_ms(self):
        """(search, scan, locate, sort) durations of the last seed batch, HIP events on its stream."""
        ms = (C.c_float * 4)()
        _check(self.lib.kg_workspace_kernel_ms(self.h, C.byref(ms)), "kg_workspace_kernel_ms")
        return [float(x) for x in ms]

    def seed_batch(self, enc: np.ndarray, offsets: np.ndarray, mode: int, min_seed_len=None, occ_thr=KG_OCC_THR_DEFAULT):
        """Host-buffer form: returns (seed_offsets[n+1], seeds structured array)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        so = np.zeros(n + 1, dtype=np.int64)
        out = C.c_void_p()
        _check(self.lib.kg_seed_batch(self.h, mode, min_seed_len or self.index.min_seed_len, occ_thr, _ptr(enc), _ptr(offsets),
                                      n, _ptr(so), C.byref(out)), "kg_seed_batch")
        total = int(so[n])
        if total == 0:
            return so, np.zeros(0, dtype=SEED_DT)
        buf = (C.c_char * (total * SEED_DT.itemsize)).from_address(out.value)
        return so, np.frombuffer(buf, dtype=SEED_DT).copy()

    def candidates_batch(self, seed_offsets: np.ndarray, pacbio: bool = False, max_gaps: int = 5):
        """GenerateAlignmentCandidateFor{Illumina,PacBio}Seq for the batch the last seed_batch() call seeded.
        Returns one list per read of (score, posDiff, seeds structured array)."""
        n = len(seed_offsets) - 1
        total = int(seed_offsets[n])
        ncand = np.zeros(n + 1, dtype=np.int32)
        pc, ps, nc, ns = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
        _check(self.lib.kg_candidates_batch(self.h, int(pacbio), max_gaps, n, total, _ptr(ncand), C.byref(pc), C.byref(nc), C.byref(ps), C.byref(ns)),
               "kg_candidates_batch")
        cands = np.frombuffer((C.c_char * (nc.value * CAND_DT.itemsize)).from_address(pc.value), dtype=CAND_DT).copy() if nc.value else np.zeros(0, CAND_DT)
        cseeds = np.frombuffer((C.c_char * (ns.value * SEED_DT.itemsize)).from_address(ps.value), dtype=SEED_DT).copy() if ns.value else np.zeros(0, SEED_DT)
        out, at = [], 0
        for r in range(n):
            lst = []
            for c in cands[at:at + ncand[r]]:
                lst.append((int(c["score"]), int(c["posDiff"]), cseeds[c["first"]:c["first"] + c["count"]].copy()))
            at += int(ncand[r])
            out.append(lst)
        return out

    def seed_batch_device(self, d_enc, d_offsets, n_reads, n_bases, d_seed_offsets, d_seeds, seed_capacity, mode,
                          min_seed_len=None, occ_thr=KG_OCC_THR_DEFAULT, stream=0):
        """Device-pointer form: arguments are raw device addresses (e.g. torch tensor .data_ptr())."""
        _check(self.lib.kg_seed_batch_device(self.h, mode, min_seed_len or self.index.min_seed_len, occ_thr, d_enc, d_offsets,
                                             n_reads, n_bases, d_seed_offsets, d_seeds, seed_capacity, stream),
               "kg_seed_batch_device")


class Index:
    """Device-resident FM-index + reference (bwa_idx_load + RestoreReferenceInfo of the reference)."""

    def __init__(self, prefix: str, device: int = 0, sa_mode: int = KG_SA_SAMPLED):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.kg_index_load(prefix.encode(), device, sa_mode, C.byref(h)), "kg_index_load")
        self.h = h
        info = IndexInfo()
        _check(self.lib.kg_index_info(self.h, C.byref(info)), "kg_index_info")
        self.info = info
        self.genome_size = int(info.genome_size)
        self.seq_len = int(info.seq_len)
        self.min_seed_len = int(info.min_seed_len)
        self.n_contigs = int(info.n_contigs)
        self.contigs = []
        for i in range(self.n_contigs):
            c = Contig()
            _check(self.lib.kg_index_contig(self.h, i, C.byref(c)), "kg_index_contig")
            self.contigs.append((c.name.decode(), int(c.fwd_start), int(c.rev_start), int(c.len)))
        self._ws = None

    def close(self):
        if self._ws is not None:
            self._ws.close()
            self._ws = None
        if self.h:
            self.lib.kg_index_destroy(self.h)
            self.h = None

    def workspace(self, n_reads: int, n_bases: int) -> Workspace:
        ws = self._ws
        if ws is None or ws.max_reads < n_reads or ws.max_bases < n_bases:
            if ws is not None:
                ws.close()
            ws = self._ws = Workspace(self, max(n_reads, 1024), max(n_bases, 1 << 16))
        return ws

    # -- rank / suffix array (bwt_occ4, bwt_sa) ------------------------------------------------------
    def selfcheck(self, samples: int = 1 << 20, seed: int = 1) -> int:
        """disagreements between one double step on the pair planes and two single BWT_Search steps (kg_index_selfcheck)"""
        bad = C.c_uint64(0)
        _check(self.lib.kg_index_selfcheck(self.h, samples, seed, C.byref(bad)), "kg_index_selfcheck")
        return int(bad.value)

    def rank_sa(self, ks):
        """(occ4[n,4], sa_walk[n], sa_full[n]) for ranks ks: bwt_occ4 / bwt_sa of the reference on the device layouts."""
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        n = len(ks)
        occ4 = np.zeros((n, 4), dtype=np.uint64)
        walk = np.zeros(n, dtype=np.uint64)
        full = np.zeros(n, dtype=np.uint64)
        _check(self.lib.kg_rank_sa_batch(self.h, _ptr(ks), n, _ptr(occ4), _ptr(walk), _ptr(full)), "kg_rank_sa_batch")
        return occ4, walk, full

    # -- seeding ------------------------------------------------------------------------------
    def _seed(self, reads, mode, min_seed_len, occ_thr):
        enc, offsets = concat_reads(reads)
        ws = self.workspace(len(reads), len(enc))
        so, seeds = ws.seed_batch(enc, offsets, mode, min_seed_len, occ_thr)
        return [seeds[so[i]:so[i + 1]] for i in range(len(reads))]

    def IdentifySeedPairs_FastMode(self, reads, min_seed_len=None, occ_thr=KG_OCC_THR_DEFAULT):
        """reads: list of uint8 code arrays (EnCodeReadSeq output).  Returns one seed array per read,
        ordered as the reference's vector<SeedPair_t> (sorted by (PosDiff, rPos))."""
        return self._seed(reads, KG_MODE_FAST, min_seed_len, occ_thr)

    def IdentifySeedPairs_SensitiveMode(self, reads, min_seed_len=None, occ_thr=KG_OCC_THR_DEFAULT):
        return self._seed(reads, KG_MODE_SENSITIVE, min_seed_len, occ_thr)

    # -- NW -----------------------------------------------------------------------------------------
    def nw_ops(self, pairs):
        """pairs: list of (bytes s1, bytes s2).  Returns list of uint8 op arrays (KG_OP_*)."""
        n = len(pairs)
        if n == 0:
            return []
        off1 = np.zeros(n + 1, dtype=np.int64)
        off2 = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(a) for a, _ in pairs], out=off1[1:])
        np.cumsum([len(b) for _, b in pairs], out=off2[1:])
        f1 = np.frombuffer(b"".join(a for a, _ in pairs) + b"\0", dtype=np.uint8).copy()
        f2 = np.frombuffer(b"".join(b for _, b in pairs) + b"\0", dtype=np.uint8).copy()
        ops = np.zeros(int(off1[n] + off2[n]) + 1, dtype=np.uint8)
        alen = np.zeros(n, dtype=np.int32)
        _check(self.lib.kg_nw_batch(self.h, _ptr(f1), _ptr(off1), _ptr(f2), _ptr(off2), n, _ptr(ops), _ptr(alen)), "kg_nw_batch")
        oo = off1 + off2
        return [ops[oo[i]:oo[i] + alen[i]].copy() for i in range(n)]

    def fragments_ops(self, frags, gpos, glen, pacbio=True, max_gaps=5):
        """kg_fragments_batch: GenerateNormalPairAlignment(rLen, frag1, gLen, frag2) for read fragments `frags` (bytes) against the
        genome fragments [gpos[i], gpos[i] + glen[i]) of the indexed text.  Returns (list of uint8 op arrays, uint8 status array)."""
        n = len(frags)
        if n == 0:
            return [], np.zeros(0, np.uint8)
        off1 = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(a) for a in frags], out=off1[1:])
        f1 = np.frombuffer(b"".join(frags) + b"\0" * 64, dtype=np.uint8).copy()
        g = np.ascontiguousarray(gpos, dtype=np.int64)
        gl = np.ascontiguousarray(glen, dtype=np.int32)
        oo = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.diff(off1) + gl, out=oo[1:])
        ops = np.zeros(int(oo[n]) + 64, dtype=np.uint8)
        alen = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.uint8)
        _check(self.lib.kg_fragments_batch(self.h, _ptr(f1), _ptr(off1), _ptr(g), _ptr(gl), n, int(pacbio), max_gaps, _ptr(ops), _ptr(oo), _ptr(alen), _ptr(status)), "kg_fragments_batch")
        return [ops[oo[i]:oo[i] + alen[i]].copy() for i in range(n)], status

    def GenerateNormalPairAlignment(self, frags, gpos, glen, text, pacbio=True, max_gaps=5):
        """Mirror of GenerateNormalPairAlignment (src/tools.cpp:142): the two gapped strings per pair (None where the request was handed back);
        `text` = the indexed text as characters (RefSequence), from which frag2 is cut as in the callers"""
        ops, status = self.fragments_ops(frags, gpos, glen, pacbio, max_gaps)
        out = []
        for i, a in enumerate(frags):
            out.append(None if status[i] else apply_ops(a, bytes(text[int(gpos[i]):int(gpos[i]) + int(glen[i])]), ops[i]))
        return out, status

    def nw_alignment(self, pairs):
        """Mirror of nw_alignment(m, s1, n, s2): returns the two gapped strings for every pair."""
        out = []
        for (s1, s2), ops in zip(pairs, self.nw_ops(pairs)):
            out.append(apply_ops(s1, s2, ops))
        return out


class Stream:
    """kg_stream_*: FASTQ text in, SAM text out (GetNextChunk ... Output*Alignments of the reference on the device)."""

    def __init__(self, index: "Index", max_reads: int = 16000, max_window: int = 8 << 20, lanes: int = 1, seed_group: int = 0):
        self.lib = load_library()
        self.index = index
        cfg = StreamConfig(max_reads, max_window, lanes, seed_group)
        h = C.c_void_p()
        _check(self.lib.kg_stream_open(index.h, C.byref(cfg), C.byref(h)), "kg_stream_open")
        self.h = h
        self.max_window = max_window

    def close(self):
        if self.h:
            self.lib.kg_stream_close(self.h)
            self.h = None

    def group_absent(self, lane: int, rounds: int):
        """seeding groups: lane `lane` has no batch for `rounds` rounds (< 0: until further notice, 0: it takes part again)"""
        _check(self.lib.kg_stream_group_absent(self.h, lane, rounds), "kg_stream_group_absent")

    def group_abort(self):
        """every lane that waits for its seeding group returns with an error (a caller's failure path)"""
        _check(self.lib.kg_stream_group_abort(self.h), "kg_stream_group_abort")

    def parse(self, text1: bytes, text2: bytes | None = None, paired: bool = True, chunk_reads: int = 4000, want_reads: int | None = None,
              eof=(True, True), begin=(0, 0), lane: int = 0) -> StreamParsed:
        """uploads the window(s) (placed at staging offset begin[f]) and runs GetNextChunk's arithmetic on the device"""
        w = StreamWindow()
        texts = [text1] + ([text2] if text2 is not None else [])
        for f, t in enumerate(texts):
            cap = C.c_int64()
            p = self.lib.kg_stream_staging(self.h, lane, f, C.byref(cap))
            assert p and begin[f] + len(t) <= cap.value
            C.memmove(p + begin[f], t, len(t))
            w.begin[f], w.end[f], w.eof[f] = begin[f], begin[f] + len(t), 1 if eof[f] else 0
            _check(self.lib.kg_stream_upload(self.h, lane, f, begin[f], begin[f] + len(t)), "kg_stream_upload")
        w.two_files = 1 if text2 is not None else 0
        w.paired = 1 if paired else 0
        w.chunk_reads = chunk_reads
        w.want_reads = want_reads if want_reads is not None else chunk_reads
        out = StreamParsed()
        _check(self.lib.kg_stream_parse(self.h, lane, C.byref(w), C.byref(out)), "kg_stream_parse")
        return out

    def fetch_reads(self, parsed: StreamParsed, lane: int = 0):
        """the reads of the parsed batch as the seeding stage sees them: list of bytes"""
        off = np.zeros(parsed.n_reads + 1, dtype=np.int64)
        enc = np.zeros(max(1, parsed.n_bases), dtype=np.uint8)
        _check(self.lib.kg_stream_fetch_reads(self.h, lane, _ptr(enc), _ptr(off)), "kg_stream_fetch_reads")
        raw = enc.tobytes()
        return [raw[off[i]:off[i + 1]] for i in range(parsed.n_reads)]

    def map(self, est_distance: int = 1500, max_insert: int = 1500, max_gaps: int = 5, multi_hit: bool = False, unset_flag: int = 0, lane: int = 0):
        """seeding .. SAM text for the parsed batch: (text per read, indices of the reads handed back)"""
        prm = StreamParams(est_distance, max_insert, max_gaps, 1 if multi_hit else 0, unset_flag, 1)
        res = StreamResult()
        _check(self.lib.kg_stream_map(self.h, lane, C.byref(prm), C.byref(res)), "kg_stream_map")
        n = res.n_reads
        off = np.ctypeslib.as_array(res.sam_off, shape=(n + 1,)).copy()
        sam = C.string_at(res.sam, res.sam_bytes)
        host = [int(res.host_reads[i]) for i in range(res.n_host_reads)]
        return [sam[off[i]:off[i + 1]] for i in range(n)], host

    def timing(self, reset: bool = False) -> dict:
        t = StreamTiming()
        _check(self.lib.kg_stream_timing(self.h, C.byref(t), 1 if reset else 0), "kg_stream_timing")
        return t.as_dict()


def apply_ops(s1: bytes, s2: bytes, ops: np.ndarray):
    """Re-insert the gaps described by an op string (what nw_alignment does in place)."""
    a, b = bytearray(), bytearray()
    i = j = 0
    for op in ops:
        if op == KG_OP_DIAG:
            a.append(s1[i]); b.append(s2[j]); i += 1; j += 1
        elif op == KG_OP_GAP1:
            a.append(0x2D); b.append(s2[j]); j += 1
        else:
            a.append(s1[i]); b.append(0x2D); i += 1
    assert i == len(s1) and j == len(s2), "op string does not consume both fragments"
    return bytes(a), bytes(b)


# ---- the host pipeline as a library (include/kart_host.h): bwa_idx_load once, Mapping() many times -------------------------
class HostStats(C.Structure):
    _fields_ = [("total_reads", C.c_int64), ("unmapped", C.c_int64), ("unique", C.c_int64), ("paired", C.c_int64), ("distance", C.c_int64),
                ("respeculated", C.c_int64), ("map_seconds", C.c_double), ("sharded", C.c_int32), ("pad", C.c_int32),
                ("stream_reads", C.c_int64), ("stream_batches", C.c_int64), ("stage_ms", C.c_double * 6), ("search_kernel_ms", C.c_double),
                ("search_kernel_launches", C.c_int64), ("search_useful_bytes", C.c_double), ("text_in_bytes", C.c_double), ("text_out_bytes", C.c_double),
                ("candidates", C.c_double), ("candidate_seeds", C.c_double), ("kernel_ms", C.c_double * 16), ("kernel_launches", C.c_int64 * 16), ("aln_counts", C.c_double * 8),
                ("lane_seconds", C.c_double * 6), ("lanes", C.c_int32), ("pad2", C.c_int32), ("text_checksum", C.c_double * 2)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n.startswith(("kernel_", "stage_", "aln_counts")) else getattr(self, n)) for n, _ in self._fields_}


_host_lib = None


def load_host_library() -> C.CDLL:
    global _host_lib
    if _host_lib is not None:
        return _host_lib
    if not os.path.exists(HOST_LIB_PATH):
        raise KartAmdError(f"{HOST_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    load_library()                                     # libkart_amd.so first (the host library links it)
    L = C.CDLL(HOST_LIB_PATH)
    L.kh_last_error.restype = C.c_char_p
    L.kh_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.kh_map.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(HostStats)]
    L.kh_close.argtypes = [C.c_void_p]
    L.kh_close.restype = None
    _host_lib = L
    return L


class HostSession:
    """Index resident on one device; map(args) = one run of the reference's Mapping() with the reference's own flags."""

    def __init__(self, prefix: str, device: int = 0, threads: int = 4):
        self.lib = load_host_library()
        h = C.c_void_p()
        if self.lib.kh_open(prefix.encode(), device, threads, C.byref(h)) != 0:
            raise KartAmdError("kh_open failed: " + self.lib.kh_last_error().decode())
        self.h = h

    def map(self, args) -> HostStats:
        argv = (C.c_char_p * len(args))(*[str(a).encode() for a in args])
        st = HostStats()
        if self.lib.kh_map(self.h, len(args), argv, C.byref(st)) != 0:
            raise KartAmdError("kh_map failed: " + self.lib.kh_last_error().decode())
        return st

    def close(self):
        if self.h:
            self.lib.kh_close(self.h)
            self.h = None
