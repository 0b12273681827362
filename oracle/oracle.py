"""ctypes bindings of the CPU oracle (oracle/liboracle.so) and, where it was built,
of the unmodified reference (oracle/_ref/libkartref_shim.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by kart_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

SEED_DT = np.dtype([("gPos", "<i8"), ("rPos", "<i4"), ("len", "<i4")])
PAIR_DT = np.dtype([("gPos", "<i8"), ("PosDiff", "<i8"), ("rPos", "<i4"), ("rLen", "<i4"),
                    ("gLen", "<i4"), ("bSimple", "<i4")])


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("searches", "lf1", "lf2", "inv", "sa", "seeds", "bases")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "kart_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(os.path.join(REF_DIR, "kart"))):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def _ptr(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


class Oracle:
    """One loaded index + the restated hot-path functions."""

    def __init__(self, prefix: str):
        self.lib = C.CDLL(build())
        L = self.lib
        L.ko_index_load.restype = C.c_void_p
        L.ko_index_load.argtypes = [C.c_char_p]
        L.ko_genome_size.restype = C.c_int64
        L.ko_seq_len.restype = C.c_uint64
        L.ko_primary.restype = C.c_uint64
        L.ko_occ.restype = C.c_uint64
        L.ko_sa.restype = C.c_uint64
        L.ko_seed_batch.restype = C.c_int64
        L.ko_alignment_boundary.restype = C.c_int64
        L.ko_ref_sequence.restype = C.c_void_p
        for fn in ("ko_genome_size", "ko_seq_len", "ko_primary", "ko_n_contigs", "ko_min_seed_len",
                   "ko_ref_sequence", "ko_index_free"):
            getattr(L, fn).argtypes = [C.c_void_p]
        L.ko_occ.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.ko_occ4.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.ko_sa.argtypes = [C.c_void_p, C.c_uint64]
        L.ko_bwt_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ko_seed_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.ko_seed_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.ko_nw.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
        L.ko_normal_pair_alignment.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
        L.ko_alignment_boundary.argtypes = [C.c_void_p, C.c_int64]
        L.ko_candidates_illumina.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int]
        L.ko_candidates_pacbio.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int]
        L.ko_identify_normal_pairs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.ko_counters_get.argtypes = [C.c_void_p]
        self.h = L.ko_index_load(prefix.encode())
        if not self.h:
            raise FileNotFoundError(f"cannot load index {prefix}")
        self.genome_size = L.ko_genome_size(self.h)
        self.seq_len = L.ko_seq_len(self.h)
        self.primary = L.ko_primary(self.h)
        self.min_seed_len = L.ko_min_seed_len(self.h)
        self.n_contigs = L.ko_n_contigs(self.h)

    def close(self):
        if self.h:
            self.lib.ko_index_free(self.h)
            self.h = None

    def ref_sequence(self) -> np.ndarray:
        p = self.lib.ko_ref_sequence(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(2 * self.genome_size,)).copy()

    def occ(self, k, c):
        return int(self.lib.ko_occ(self.h, C.c_uint64(k & (2**64 - 1)), c))

    def occ4(self, k):
        out = np.zeros(4, dtype=np.uint64)
        self.lib.ko_occ4(self.h, C.c_uint64(k & (2**64 - 1)), _ptr(out))
        return out

    def sa(self, k):
        return int(self.lib.ko_sa(self.h, k))

    def bwt_search(self, enc: np.ndarray, start: int, stop: int, min_seed_len=None):
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        locs = np.zeros(64, dtype=np.uint64)
        ln = C.c_int(0)
        f = self.lib.ko_bwt_search(self.h, _ptr(enc), start, stop, min_seed_len or self.min_seed_len, C.byref(ln), _ptr(locs))
        return ln.value, f, locs[:f].copy()

    def seed_read(self, enc: np.ndarray, mode: int = 0, min_seed_len=None) -> np.ndarray:
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        cap = 4096
        while True:
            out = np.zeros(cap, dtype=SEED_DT)
            n = self.lib.ko_seed_read(self.h, mode, min_seed_len or self.min_seed_len, _ptr(enc), len(enc), _ptr(out), cap)
            if n >= 0:
                return out[:n]
            cap = -n

    def seed_batch(self, enc: np.ndarray, offsets: np.ndarray, mode: int = 0, min_seed_len=None, threads: int = 1):
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        so = np.zeros(n + 1, dtype=np.int64)
        cap = max(1024, 16 * n)
        while True:
            out = np.zeros(cap, dtype=SEED_DT)
            t = self.lib.ko_seed_batch(self.h, mode, min_seed_len or self.min_seed_len, _ptr(enc), _ptr(offsets), n,
                                       _ptr(so), _ptr(out), cap, threads)
            if t >= 0:
                return so, out[:t]
            cap = -t

    def counters(self, reset=False):
        c = Counters()
        self.lib.ko_counters_get(C.byref(c))
        if reset:
            self.lib.ko_counters_reset()
        return c.as_dict()

    def nw(self, s1: bytes, s2: bytes):
        o1 = C.create_string_buffer(len(s1) + len(s2) + 2)
        o2 = C.create_string_buffer(len(s1) + len(s2) + 2)
        self.lib.ko_nw(s1, len(s1), s2, len(s2), o1, o2)
        return o1.value, o2.value

    def normal_pair_alignment(self, s1: bytes, s2: bytes, pacbio=True, max_gaps=5):
        """GenerateNormalPairAlignment(rLen, frag1, gLen, frag2): the two aligned strings"""
        o1 = C.create_string_buffer(len(s1) + len(s2) + 2)
        o2 = C.create_string_buffer(len(s1) + len(s2) + 2)
        self.lib.ko_normal_pair_alignment(int(pacbio), max_gaps, s1, len(s1), s2, len(s2), o1, o2)
        return o1.value, o2.value

    def boundary(self, g):
        return int(self.lib.ko_alignment_boundary(self.h, g))

    def _cands(self, fn, args, n_seeds):
        ccap, pcap = n_seeds + 1, n_seeds + 1
        off = np.zeros(ccap + 1, dtype=np.int32)
        score = np.zeros(ccap, dtype=np.int32)
        pd = np.zeros(ccap, dtype=np.int64)
        pairs = np.zeros(pcap, dtype=PAIR_DT)
        n = fn(*args, _ptr(off), _ptr(score), _ptr(pd), _ptr(pairs), ccap, pcap)
        assert n >= 0
        return [(int(score[i]), int(pd[i]), pairs[off[i]:off[i + 1]].copy()) for i in range(n)]

    def candidates(self, rlen: int, seeds: np.ndarray, pacbio: bool = False, max_gaps: int = 5):
        seeds = np.ascontiguousarray(seeds, dtype=SEED_DT)
        if pacbio:
            return self._cands(self.lib.ko_candidates_pacbio, (self.h, rlen, _ptr(seeds), len(seeds)), len(seeds))
        return self._cands(self.lib.ko_candidates_illumina, (self.h, rlen, max_gaps, _ptr(seeds), len(seeds)), len(seeds))

    def identify_normal_pairs(self, rlen: int, glen: int, pairs: np.ndarray) -> np.ndarray:
        cap = 2 * len(pairs) + 4
        buf = np.zeros(cap, dtype=PAIR_DT)
        buf[:len(pairs)] = pairs
        n = self.lib.ko_identify_normal_pairs(rlen, glen, _ptr(buf), len(pairs), cap)
        assert n >= 0
        return buf[:n].copy()


class RefShim:
    """The unmodified reference's functions (needs oracle/_ref, i.e. a container with /root/reference).
    The reference keeps its index in process globals, so one RefShim per process."""

    def __init__(self, prefix: str, pacbio: bool = False, max_gaps: int = 5, threads: int = 1):
        path = os.path.join(REF_DIR, "libkartref_shim.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = L = C.CDLL(path, mode=os.RTLD_LAZY | os.RTLD_GLOBAL)
        L.shim_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.shim_genome_size.restype = C.c_int64
        L.shim_primary.restype = C.c_uint64
        L.shim_seq_len.restype = C.c_uint64
        L.shim_occ.restype = C.c_uint64
        L.shim_occ.argtypes = [C.c_uint64, C.c_int]
        L.shim_occ4.argtypes = [C.c_uint64, C.c_void_p]
        L.shim_sa.restype = C.c_uint64
        L.shim_sa.argtypes = [C.c_uint64]
        L.shim_ref_sequence.restype = C.c_void_p
        L.shim_bwt_search.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.shim_seed_read.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.shim_nw.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
        if hasattr(L, "shim_normal_pair_alignment"):
            L.shim_normal_pair_alignment.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
        if hasattr(L, "shim_seed_batch"):
            L.shim_seed_batch.restype = C.c_int64
            L.shim_seed_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.shim_candidates.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int]
        L.shim_identify_normal_pairs.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        self.min_seed_len = L.shim_init(prefix.encode(), threads, int(pacbio), max_gaps)
        if self.min_seed_len < 0:
            raise RuntimeError("reference failed to load index")
        self.genome_size = L.shim_genome_size()
        self.primary = L.shim_primary()
        self.seq_len = L.shim_seq_len()

    def set_mode(self, pacbio: bool, max_gaps: int = 5):
        self.lib.shim_set_mode(int(pacbio), max_gaps)

    def seed_batch_count(self, enc: np.ndarray, offsets: np.ndarray, mode: int = 0, threads: int = 1) -> int:
        """IdentifySeedPairs_* of the reference over a batch on `threads` threads; returns the number of seeds (timing aid)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        return int(self.lib.shim_seed_batch(mode, _ptr(enc), _ptr(offsets), len(offsets) - 1, threads))

    def ref_sequence(self) -> np.ndarray:
        p = self.lib.shim_ref_sequence()
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(2 * self.genome_size,)).copy()

    def occ(self, k, c):
        return int(self.lib.shim_occ(C.c_uint64(k & (2**64 - 1)), c))

    def occ4(self, k):
        out = np.zeros(4, dtype=np.uint64)
        self.lib.shim_occ4(C.c_uint64(k & (2**64 - 1)), _ptr(out))
        return out

    def sa(self, k):
        return int(self.lib.shim_sa(k))

    def bwt_search(self, enc, start, stop):
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        locs = np.zeros(64, dtype=np.uint64)
        ln = C.c_int(0)
        f = self.lib.shim_bwt_search(_ptr(enc), start, stop, C.byref(ln), _ptr(locs))
        return ln.value, f, locs[:f].copy()

    def seed_read(self, enc, mode=0):
        # pad: SensitiveMode reads past rlen after an N run
        enc = np.concatenate([np.ascontiguousarray(enc, dtype=np.uint8), np.full(64, 4, dtype=np.uint8)])
        cap = 4096
        while True:
            out = np.zeros(cap, dtype=SEED_DT)
            n = self.lib.shim_seed_read(mode, _ptr(enc), len(enc) - 64, _ptr(out), cap)
            if n >= 0:
                return out[:n]
            cap = -n

    def nw(self, s1: bytes, s2: bytes):
        o1 = C.create_string_buffer(len(s1) + len(s2) + 2)
        o2 = C.create_string_buffer(len(s1) + len(s2) + 2)
        self.lib.shim_nw(s1, len(s1), s2, len(s2), o1, o2)
        return o1.value, o2.value

    def normal_pair_alignment(self, s1: bytes, s2: bytes):
        o1 = C.create_string_buffer(len(s1) + len(s2) + 2)
        o2 = C.create_string_buffer(len(s1) + len(s2) + 2)
        self.lib.shim_normal_pair_alignment(s1, len(s1), s2, len(s2), o1, o2)
        return o1.value, o2.value

    def candidates(self, rlen, seeds, pacbio=False):
        seeds = np.ascontiguousarray(seeds, dtype=SEED_DT)
        n_seeds = len(seeds)
        ccap, pcap = n_seeds + 1, n_seeds + 1
        off = np.zeros(ccap + 1, dtype=np.int32)
        score = np.zeros(ccap, dtype=np.int32)
        pd = np.zeros(ccap, dtype=np.int64)
        pairs = np.zeros(pcap, dtype=PAIR_DT)
        n = self.lib.shim_candidates(int(pacbio), rlen, _ptr(seeds), n_seeds, _ptr(off), _ptr(score), _ptr(pd), _ptr(pairs), ccap, pcap)
        assert n >= 0
        return [(int(score[i]), int(pd[i]), pairs[off[i]:off[i + 1]].copy()) for i in range(n)]

    def identify_normal_pairs(self, rlen, glen, pairs):
        cap = 2 * len(pairs) + 4
        buf = np.zeros(cap, dtype=PAIR_DT)
        buf[:len(pairs)] = pairs
        n = self.lib.shim_identify_normal_pairs(rlen, glen, _ptr(buf), len(pairs), cap)
        assert n >= 0
        return buf[:n].copy()


def ref_available() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "kart")) and os.path.exists(os.path.join(REF_DIR, "libkartref.so"))


def ref_build_index(fasta: str, prefix: str):
    """Run the reference's own index builder (oracle/_ref/bwt_index)."""
    subprocess.check_call([os.path.join(REF_DIR, "bwt_index"), fasta, prefix], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def ref_kart(args, **kw):
    """Run the unmodified reference mapper (oracle/_ref/kart)."""
    return subprocess.run([os.path.join(REF_DIR, "kart")] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, **kw)
