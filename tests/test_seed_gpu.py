"""GPU parity for the seeding path: HIP kernels (through the C ABI) vs the golden vectors of the
unmodified reference and vs the CPU oracle on larger seeded inputs.  Bit-exact (integer work)."""
import numpy as np
import pytest

from conftest import split
from kart_amd import api, synth

pytestmark = pytest.mark.gpu


def _check_reads(ix, reads, want, mode):
    fn = ix.IdentifySeedPairs_FastMode if mode == 0 else ix.IdentifySeedPairs_SensitiveMode
    got = fn(reads)
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w), (i, len(g), len(w))
        assert (g == w.astype(api.SEED_DT)).all(), i


@pytest.mark.parametrize("which", ["gpu_index", "gpu_index_full", "gpu_index_compact", "gpu_index_wide", "gpu_index_dense4", "gpu_index_dense8"])
def test_fast_mode_golden(golden, which, request):
    ix = request.getfixturevalue(which)
    _check_reads(ix, split(golden["fast_enc"], golden["fast_off"]), split(golden["fast_seeds"], golden["fast_seed_off"]), 0)


@pytest.mark.parametrize("which", ["gpu_index", "gpu_index_full", "gpu_index_compact", "gpu_index_wide", "gpu_index_dense4", "gpu_index_dense8"])
def test_sensitive_mode_golden(golden, which, request):
    ix = request.getfixturevalue(which)
    _check_reads(ix, split(golden["sens_enc"], golden["sens_off"]), split(golden["sens_seeds"], golden["sens_seed_off"]), 1)


@pytest.mark.parametrize("which", ["gpu_index", "gpu_index_full", "gpu_index_compact", "gpu_index_wide", "gpu_index_dense8"])
@pytest.mark.parametrize("seg_len", [128, 192, 512])
def test_sensitive_mode_walks_from_segment_starts(golden, which, seg_len, request, oracle_small, monkeypatch):
    """SensitiveMode within a read in parallel (SeedArgs::vr_read, search.inc): the loop of IdentifySeedPairs_SensitiveMode
    (src/AlignmentCandidates.cpp:132-169) is a walk p -> next(p) whose links depend on the read alone; lanes start a walk at every segment
    start, claim positions before they search them, and the walk from 0 is put together from the links afterwards.  Same seeds as the
    reference's golden vectors and as the oracle on long reads with N runs, at three segment lengths, and as one lane per read (KG_NO_SEGMENTS)."""
    import os
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    from conftest import GOLDEN
    ix = request.getfixturevalue(which)
    monkeypatch.setenv("KG_SEG_LEN", str(seg_len))
    reads = split(golden["sens_enc"], golden["sens_off"])
    long_gold = [(r, w) for r, w in zip(reads, split(golden["sens_seeds"], golden["sens_seed_off"])) if len(r) >= 4 * seg_len]
    if long_gold:
        _check_reads(ix, [r for r, _ in long_gold], [w for _, w in long_gold], 1)
    # long reads (7 kb at 15 % error, 3 kb at 2 %, N runs, a read that is ALL N, reads around the segment length) against the oracle
    genome = {n: s_ for n, _, s_ in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    _, r1 = synth.simulate_long_reads(genome, 40, seed=3, read_len=7000, err=0.15, indel_err_frac=0.1)
    _, r2 = synth.simulate_long_reads(genome, 20, seed=4, read_len=3000, err=0.02, indel_err_frac=0.3)
    _, r3 = synth.simulate_long_reads(genome, 12, seed=5, read_len=4 * seg_len + 7, err=0.1)
    rs = [synth.encode(np.array(r, dtype=np.uint8)) for r in list(r1) + list(r2) + list(r3)]
    rs[2][1000:1100] = 4; rs[5][seg_len - 3:seg_len + 40] = 4; rs[9][:] = 4; rs[11][::17] = 4       # (no N run at a read's end: SURVEY App. B-10 is fenced off)
    got = ix.IdentifySeedPairs_SensitiveMode(rs)
    enc, off = api.concat_reads(rs)
    so, seeds = oracle_small.seed_batch(enc, off, 1)
    for i in range(len(rs)):
        w = seeds[so[i]:so[i + 1]].astype(api.SEED_DT)
        assert len(got[i]) == len(w) and (got[i] == w).all(), (i, len(got[i]), len(w))
    monkeypatch.setenv("KG_NO_SEGMENTS", "1")
    again = ix.IdentifySeedPairs_SensitiveMode(rs)
    assert all(len(a) == len(b) and (a == b).all() for a, b in zip(got, again))


def test_exact_long_reads_fall_back_to_one_walk_per_read(gpu_index_full, oracle_small, monkeypatch):
    """Error-free 7 kb reads: every SensitiveMode search returns its 30 bases, the walks from the segment starts never merge
    (512 k mod 30 differs for all k) and leave ~0.25 hits per base where the workspace's hit list holds one walk's worth (~0.1):
    the batch is re-seeded with one walk per read instead of failing (kgi_seed_resident), the seeds are the oracle's."""
    import os
    from kart_amd.index_build import read_fasta
    from conftest import GOLDEN
    ix = gpu_index_full
    genome = np.concatenate([synth.encode(np.asarray(s_, dtype=np.uint8)) for _, _, s_ in read_fasta(os.path.join(GOLDEN, "small.fa"))])
    rng = np.random.default_rng(11)
    n, rlen = 6000, 7000
    starts = rng.integers(0, len(genome) - rlen, n)
    enc = np.concatenate([genome[s_:s_ + rlen] for s_ in starts])
    enc[enc > 3] = 0
    off = np.arange(n + 1, dtype=np.int64) * rlen
    ws = ix.workspace(n, len(enc))
    so, seeds = ws.seed_batch(enc, off, api.KG_MODE_SENSITIVE)
    assert ws.segment_fallbacks() == 1, "the segment walks of exact reads were expected to outgrow the hit list"
    sub = 40
    so_o, seeds_o = oracle_small.seed_batch(enc[:sub * rlen], off[:sub + 1], 1)
    assert (so[:sub + 1] == so_o).all() and (seeds[:so[sub]] == seeds_o.astype(api.SEED_DT)).all()
    # the next batch on this workspace goes straight to one walk per read; noisy reads afterwards get their segments back
    so2, seeds2 = ws.seed_batch(enc, off, api.KG_MODE_SENSITIVE)
    assert ws.segment_fallbacks() == 1 and (so2 == so).all() and (seeds2 == seeds).all()
    # (the workspace is the index fixture's own, cached for the tests behind this one: not closed here)


def test_counters_match_oracle(golden, gpu_index, oracle_small):
    """the kernel's work counters are the ones the roofline figure is computed from"""
    oracle_small.counters(reset=True)
    oracle_small.seed_batch(golden["fast_enc"], golden["fast_off"], 0)
    want = oracle_small.counters(reset=True)
    ws = gpu_index.workspace(len(golden["fast_off"]) - 1, len(golden["fast_enc"]))
    ws.set_single_steps(True)         # (double steps keep lf1 + lf2, not the split: second test below)
    ws.seed_batch(golden["fast_enc"], golden["fast_off"], 0)
    got = ws.counters().as_dict()
    # the kernel walks LF from the ranks of the reverse-complement interval (x[1]+i) while the
    # reference walks from x[0]+i: same text positions, different ranks, so the number of invPsi
    # steps agrees only in expectation (geometric, mean 31 per hit)
    inv_got, inv_want = got.pop("inv"), want.pop("inv")
    assert got == want
    assert abs(inv_got - inv_want) < 0.1 * inv_want


def test_counters_with_text_comparison(golden, gpu_index_full, oracle_small):
    """with the full SA resident, single-suffix searches finish by comparing against the text; the
    kernel still reports the reference's step count (each compared base = one LF step of the reference).
    Only the one-or-two-blocks split of those steps is not reconstructed (they are booked as one-block
    steps; a single-suffix step touches two blocks when its rank is a multiple of 128)."""
    oracle_small.counters(reset=True)
    oracle_small.seed_batch(golden["fast_enc"], golden["fast_off"], 0)
    want = oracle_small.counters(reset=True)
    ws = gpu_index_full.workspace(len(golden["fast_off"]) - 1, len(golden["fast_enc"]))
    ws.set_single_steps(True)
    ws.seed_batch(golden["fast_enc"], golden["fast_off"], 0)
    got = ws.counters().as_dict()
    assert got["searches"] == want["searches"] and got["seeds"] == want["seeds"] and got["bases"] == want["bases"]
    assert got["lf1"] + got["lf2"] == want["lf1"] + want["lf2"]
    assert got["lf2"] <= want["lf2"] and want["lf2"] - got["lf2"] < 0.02 * (want["lf1"] + want["lf2"])


@pytest.mark.parametrize("which", ["gpu_index", "gpu_index_full"])
@pytest.mark.parametrize("mode", [0, 1])
def test_double_steps_keep_the_step_count_and_the_seeds(golden, which, mode, request):
    """two extension steps at once on the pair planes (the default): same seeds and the same total of the reference's steps as
    with single steps (whose counters test_counters_match_oracle pins to the oracle), and the kernel reports double steps"""
    ix = request.getfixturevalue(which)
    enc, off = (golden["fast_enc"], golden["fast_off"]) if mode == 0 else (golden["sens_enc"], golden["sens_off"])
    res = []
    for single in (False, True):
        ws = ix.workspace(len(off) - 1, len(enc))
        ws.set_single_steps(single)
        o, seeds = ws.seed_batch(enc, off, mode)
        res.append((o.copy(), seeds.copy(), ws.counters().as_dict(), ws.traffic().as_dict()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    a, b = res[0][2], res[1][2]
    assert a["lf1"] + a["lf2"] == b["lf1"] + b["lf2"] and a["searches"] == b["searches"] and a["seeds"] == b["seeds"]
    assert res[0][3]["double_steps"] > 0 and res[1][3]["double_steps"] == 0
    assert res[0][3]["rank_steps"] < res[1][3]["rank_steps"]


@pytest.mark.parametrize("which", ["gpu_index", "gpu_index_full"])
def test_pair_planes_known_answers(which, request):
    """device KAT of the two-step rank structure: 4 M pseudo-random intervals of every width x every pair of bases, one double
    step == two single BWT_Search steps (src/bwt_search.cpp:157-168) on the plain rank structure"""
    ix = request.getfixturevalue(which)
    for seed in (1, 2, 3):
        assert ix.selfcheck(1 << 22, seed) == 0


def test_empty_and_tiny_batches(gpu_index):
    assert gpu_index.IdentifySeedPairs_FastMode([]) == []
    out = gpu_index.IdentifySeedPairs_FastMode([np.zeros(1, np.uint8), np.zeros(0, np.uint8), np.full(200, 4, np.uint8)])
    assert [len(o) for o in out] == [0, 0, 0]


@pytest.mark.parametrize("which", ["gpu_index", "gpu_index_full", "gpu_index_compact", "gpu_index_wide", "gpu_index_dense4", "gpu_index_dense8"])
def test_random_pairs_vs_oracle(which, request, oracle_small):
    """20k reads of 150 bp with 2 % errors + N's, ragged lengths mixed in; both modes"""
    ix = request.getfixturevalue(which)
    g = {}
    cur = None
    for line in open(request.config.rootpath / "tests" / "golden" / "small.fa", "rb"):
        if line.startswith(b">"):
            cur = line[1:].strip().decode(); g[cur] = []
        else:
            g[cur].append(line.strip())
    g = {k: np.frombuffer(b"".join(v), dtype=np.uint8) for k, v in g.items()}
    _, r1, r2 = synth.simulate_pairs(g, 10000, seed=11, err=0.02, n_frac=0.001)
    reads = [synth.encode(r) for r in r1] + [synth.encode(r) for r in r2]
    rng = np.random.default_rng(5)
    for i in rng.integers(0, len(reads), size=500):
        reads[i] = reads[i][: int(rng.integers(0, 151))]
    enc, off = api.concat_reads(reads)
    for mode in (0, 1):
        so_o, s_o = oracle_small.seed_batch(enc, off, mode, threads=8)
        ws = ix.workspace(len(reads), len(enc))
        so_g, s_g = ws.seed_batch(enc, off, mode)
        assert (so_g == so_o).all()
        assert (s_g == s_o.astype(api.SEED_DT)).all()


def test_occ_threshold_and_min_seed_len_parameters(golden, gpu_index, oracle_small):
    enc, off = golden["fast_enc"], golden["fast_off"]
    for msl in (13, 16):
        so_o, s_o = oracle_small.seed_batch(enc, off, 0, min_seed_len=msl)
        ws = gpu_index.workspace(len(off) - 1, len(enc))
        so_g, s_g = ws.seed_batch(enc, off, 0, min_seed_len=msl)
        assert (so_g == so_o).all() and (s_g == s_o.astype(api.SEED_DT)).all()


def test_wide_index_instantiation(golden, built_lib, gpu_index_full, tmp_path, request):
    """search_kernel<uint64_t>, the 8-byte q-mer entries and the u64 full SA -- the variants an hg38-sized index
    (2L > 2^32) selects -- forced onto the small index (KG_FORCE_U64) in a child process; results must not change
    (golden reads, plus reads at the ends of the text whose expected seeds come from the narrow kernels)."""
    import os, subprocess, sys, textwrap
    from conftest import ROOT
    chunks = [l.strip() for l in open(request.config.rootpath / "tests" / "golden" / "small.fa", "rb") if not l.startswith(b">")]
    fwd = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    raw = [fwd[:150], fwd[-150:], fwd[1:40], fwd[-41:-1], np.concatenate([fwd[-100:], fwd[:50]]), synth.revcomp(fwd[:150]), synth.revcomp(fwd[-150:])]
    reads = [synth.encode(r) for r in raw]
    enc, off = api.concat_reads(reads)
    exp = {}
    for mode in (0, 1):
        so, seeds = gpu_index_full.workspace(len(reads), len(enc)).seed_batch(enc, off, mode)
        exp["so%d" % mode], exp["s%d" % mode] = so, seeds
    edge = str(tmp_path / "edge.npz")
    np.savez(edge, enc=enc, off=off, **exp)
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r)
        from kart_amd import api
        g = np.load(%r, allow_pickle=True)
        e = np.load(%r)
        for sa_mode in (api.KG_SA_SAMPLED, api.KG_SA_FULL, api.KG_SA_FULL40, api.KG_SA_FULL40_WIDE, api.KG_SA_DENSE4, api.KG_SA_DENSE8):
            ix = api.Index(%r, 0, sa_mode)
            for mode, key in ((0, "fast"), (1, "sens")):
                ws = ix.workspace(len(g[key + "_off"]) - 1, len(g[key + "_enc"]))
                so, seeds = ws.seed_batch(g[key + "_enc"], g[key + "_off"], mode)
                assert (so == g[key + "_seed_off"]).all() and (seeds == g[key + "_seeds"].astype(api.SEED_DT)).all(), (sa_mode, key)
                so, seeds = ix.workspace(len(e["off"]) - 1, len(e["enc"])).seed_batch(e["enc"], e["off"], mode)
                assert (so == e["so%%d" %% mode]).all() and (seeds == e["s%%d" %% mode]).all(), ("edge", sa_mode, mode)
            ix.close()
        print("wide ok")
    ''') % (ROOT, os.path.join(ROOT, "tests", "golden", "hotpath_small.npz"), edge, os.path.join(ROOT, "tests", "golden", "idx", "small"))
    # ... and the same checks for the variant that reads the raw read codes in the search kernel (no pack pre-pass)
    for env in ({"KG_FORCE_U64": "1"}, {"KG_FUSED_PACK": "1"}, {"KG_FORCE_U64": "1", "KG_FUSED_PACK": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0 and b"wide ok" in r.stdout, (env, r.stdout.decode()[-800:])


def _check_candidates(got, want_iter):
    for ri, lst in enumerate(got):
        want = want_iter(ri)
        assert len(lst) == len(want), (ri, len(lst), len(want))
        for (gs, gp, gv), (ws_, wp, wv) in zip(lst, want):
            assert (gs, gp) == (ws_, wp), ri
            assert len(gv) == len(wv), ri
            assert (gv["gPos"] == wv["gPos"]).all() and (gv["rPos"] == wv["rPos"]).all() and (gv["len"] == wv["rLen"]).all(), ri


@pytest.mark.parametrize("pacbio", [0, 1])
def test_candidates_golden(golden, gpu_index, pacbio):
    """kg_candidates_batch == the candidates the unmodified reference produced (golden cand_rows/cand_pairs)"""
    key = "sens" if pacbio else "fast"
    enc, off = golden[key + "_enc"], golden[key + "_off"]
    rows = golden["cand_rows"]
    cp_off = np.concatenate([[0], np.cumsum(rows[:, 4])])
    ws = gpu_index.workspace(len(off) - 1, len(enc))
    so, _ = ws.seed_batch(enc, off, pacbio)
    got = ws.candidates_batch(so, bool(pacbio))
    by_read = {}
    for gi, row in enumerate(rows):
        if row[0] == pacbio:
            by_read.setdefault(int(row[1]), []).append((int(row[2]), int(row[3]), golden["cand_pairs"][cp_off[gi]:cp_off[gi + 1]]))
    _check_candidates(got, lambda ri: by_read.get(ri, []))
    assert sum(len(x) for x in got) == sum(1 for r in rows if r[0] == pacbio)


def test_candidates_vs_oracle(gpu_index, oracle_small, request):
    """both chaining modes on 6k simulated reads (Illumina: incl. ragged lengths and max_gaps variants)"""
    g = {}
    cur = None
    for line in open(request.config.rootpath / "tests" / "golden" / "small.fa", "rb"):
        if line.startswith(b">"):
            cur = line[1:].strip().decode(); g[cur] = []
        else:
            g[cur].append(line.strip())
    g = {k: np.frombuffer(b"".join(v), dtype=np.uint8) for k, v in g.items()}
    _, r1, r2 = synth.simulate_pairs(g, 3000, seed=23, err=0.03, n_frac=0.001)
    reads = [synth.encode(r) for r in r1] + [synth.encode(r) for r in r2]
    rng = np.random.default_rng(9)
    for i in rng.integers(0, len(reads), size=300):
        reads[i] = reads[i][: int(rng.integers(0, 151))]
    enc, off = api.concat_reads(reads)
    ws = gpu_index.workspace(len(reads), len(enc))
    for pacbio, max_gaps in ((0, 5), (0, 0), (0, 30), (1, 5)):
        so, seeds = ws.seed_batch(enc, off, pacbio)
        got = ws.candidates_batch(so, bool(pacbio), max_gaps)
        _check_candidates(got, lambda ri: oracle_small.candidates(len(reads[ri]), seeds[so[ri]:so[ri + 1]], bool(pacbio), max_gaps))


def test_candidates_need_the_seeded_batch(golden, gpu_index):
    ws = api.Workspace(gpu_index, 1024, 1 << 16)
    with pytest.raises(RuntimeError, match="kg_seed_batch first"):
        ws.candidates_batch(np.zeros(2, np.int64))
    so, _ = ws.seed_batch(golden["fast_enc"][: golden["fast_off"][10]], golden["fast_off"][:11], 0)
    with pytest.raises(RuntimeError, match="batch shape"):
        ws.candidates_batch(so[:5])          # not the batch that was seeded
    assert len(ws.candidates_batch(so)) == 10
    ws.close()


def test_long_reads_cover_every_sort_class(gpu_index_full, gpu_index, gpu_index_dense4, oracle_small, request):
    """reads of 200 bp .. 60 kb in SensitiveMode give seed lists of a handful up to several thousand per read: the register
    network, the wave bitonic, both LDS classes and the in-place Shell pass of the sort, and multi-round expansion in locate"""
    g = {}
    cur = None
    for line in open(request.config.rootpath / "tests" / "golden" / "small.fa", "rb"):
        if line.startswith(b">"):
            cur = line[1:].strip().decode(); g[cur] = []
        else:
            g[cur].append(line.strip())
    g = {k: np.frombuffer(b"".join(v), dtype=np.uint8) for k, v in g.items()}
    big = max(g.values(), key=len)
    rng = np.random.default_rng(31)
    reads = []
    for ln in (200, 900, 2500, 7000, 20000, min(60000, len(big) - 10)):
        for rep in range(3):
            p = int(rng.integers(0, len(big) - ln))
            r = big[p:p + ln].copy()
            e = rng.random(ln) < 0.04
            r[e] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(e.sum()))]
            reads.append(synth.encode(r if rep != 1 else synth.revcomp(r)))
    enc, off = api.concat_reads(reads)
    so_o, s_o = oracle_small.seed_batch(enc, off, 1, threads=4)
    sizes = np.diff(so_o)
    assert sizes.max() > 2048 and ((sizes > 256) & (sizes <= 2048)).any() and ((sizes > 64) & (sizes <= 256)).any() and (sizes <= 64).any()
    for ix in (gpu_index_full, gpu_index, gpu_index_dense4):
        ws = ix.workspace(len(reads), len(enc))
        so_g, s_g = ws.seed_batch(enc, off, 1)
        assert (so_g == so_o).all()
        assert (s_g == s_o.astype(api.SEED_DT)).all()


def test_reads_at_the_ends_of_the_text(gpu_index_full, gpu_index, gpu_index_dense4, gpu_index_dense8, oracle_small, request):
    """matches that run into the first / last base of the forward strand (= the last / first base of the indexed text
    on the other strand), across contig junctions and across the forward/reverse seam, with and without tails that
    cannot match: the text-comparison path must stop exactly where the reference's interval empties"""
    chunks = []
    for line in open(request.config.rootpath / "tests" / "golden" / "small.fa", "rb"):
        if not line.startswith(b">"):
            chunks.append(line.strip())
    fwd = np.frombuffer(b"".join(chunks), dtype=np.uint8)          # all contigs, as the index concatenates them
    L = len(fwd)
    rng = np.random.default_rng(77)
    rnd = lambda n: np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)]
    raw = []
    for ln in (20, 40, 150, 151, 300):
        for a in (0, 1, 2, 3, 17):
            raw.append(fwd[a:a + ln])                               # starts at / near the first base
            raw.append(fwd[L - ln - a:L - a])                       # ends at / near the last base
            raw.append(np.concatenate([rnd(30), fwd[a:a + ln]]))    # unmatchable head, then the text start
            raw.append(np.concatenate([fwd[L - ln - a:L - a], rnd(30)]))   # runs off the end of the strand
            raw.append(np.concatenate([synth.revcomp(fwd[:ln]), fwd[:ln]]))   # the seam: revcomp(end of text) + start
    for edge in (3000, 63000, 93000):                               # contig junctions of small.fa
        for ln in (60, 150):
            raw.append(fwd[edge - ln // 2:edge + ln // 2])
    reads = []
    for r in raw:
        reads.append(synth.encode(r))
        reads.append(synth.encode(synth.revcomp(r)))
    enc, off = api.concat_reads(reads)
    for mode in (0, 1):
        so_o, s_o = oracle_small.seed_batch(enc, off, mode)
        assert so_o[-1] > 0
        for ix in (gpu_index_full, gpu_index, gpu_index_dense4, gpu_index_dense8):    # (dense: walks through the primary row wrap, bwt_sa :128-138)
            ws = ix.workspace(len(reads), len(enc))
            so_g, s_g = ws.seed_batch(enc, off, mode)
            assert (so_g == so_o).all() and (s_g == s_o.astype(api.SEED_DT)).all(), mode


def test_character_input_is_encoded_on_the_device(gpu_index_full, gpu_index, oracle_small, request):
    """KG_INPUT_ASCII: the same reads given as characters (upper and lower case, N, n, IUPAC codes, other bytes) must seed
    exactly like their nst_nt4_table codes"""
    chunks = [l.strip() for l in open(request.config.rootpath / "tests" / "golden" / "small.fa", "rb") if not l.startswith(b">")]
    fwd = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    rng = np.random.default_rng(123)
    odd = np.frombuffer(b"NnRYKMSWacgtACGT.-*@\x00\xff", dtype=np.uint8)
    texts = []
    for i in range(3000):
        ln = int(rng.integers(0, 400))
        p = int(rng.integers(0, len(fwd) - ln - 1))
        r = fwd[p:p + ln].copy()
        if i % 3 == 0:
            r = np.frombuffer(r.tobytes().lower(), np.uint8).copy()
        m = rng.random(ln) < 0.03
        r[m] = odd[rng.integers(0, len(odd), int(m.sum()))]
        texts.append(r)
    txt, off = api.concat_reads(texts)
    enc, _ = api.concat_reads([synth.encode(t) for t in texts])
    assert (enc <= 4).all()
    for mode in (0, 1):
        for ix in (gpu_index_full, gpu_index):
            ws = ix.workspace(len(texts), max(1, len(txt)))
            so_c, s_c = ws.seed_batch(enc, off, mode)
            so_a, s_a = ws.seed_batch(txt, off, mode | api.KG_INPUT_ASCII)
            assert (so_a == so_c).all() and (s_a == s_c).all(), mode
    so_o, s_o = oracle_small.seed_batch(enc, off, 0)
    assert (so_a[-1] >= 0) and (gpu_index_full.workspace(len(texts), len(txt)).seed_batch(txt, off, api.KG_INPUT_ASCII)[1] == s_o.astype(api.SEED_DT)).all()


@pytest.mark.parametrize("which", ["gpu_index", "gpu_index_full"])
def test_rank_and_sa_kats_on_the_device(golden, which, request):
    """the reference's own bwt_occ4 / bwt_sa answers (tests/golden/hotpath_small.npz: kat_k, kat_occ4, kat_sa, written by
    oracle/pin_against_ref.py from the reference's object code; reference src/bwt_search.cpp:35-138) against the device's
    rank_plane / lf_step_plane walk / expanded suffix array, through kg_rank_sa_batch"""
    ix = request.getfixturevalue(which)
    ks = golden["kat_k"]
    occ4, walk, full = ix.rank_sa(ks)
    assert (occ4 == golden["kat_occ4"]).all()
    assert (walk == golden["kat_sa"]).all()
    if which == "gpu_index_full":
        # SA[0] is the sentinel suffix: 2L in the expanded array, sa[0] = -1 (wrapping) in the reference's samples
        want = np.where(ks == 0, np.uint64(ix.seq_len), golden["kat_sa"])
        assert (full == want).all()
    else:
        assert (full == np.uint64(2**64 - 1)).all()
    # k = (bwtint_t)-1 is legal for the rank part (src/bwt_search.cpp:72-75)
    occ4, _, _ = ix.rank_sa(np.array([2**64 - 1], dtype=np.uint64))
    assert (occ4 == 0).all()


def test_rank_and_sa_every_rank_of_the_small_index(gpu_index_full, oracle_small):
    """all 2L+1 ranks: device occ4 against the oracle's, and walk == expanded suffix array (a permutation of 0..2L)"""
    n = gpu_index_full.seq_len
    ks = np.arange(n + 1, dtype=np.uint64)
    occ4, walk, full = gpu_index_full.rank_sa(ks)
    assert (walk[1:] == full[1:]).all() and full[0] == n and walk[0] == 2**64 - 1
    assert (np.sort(full) == ks).all()
    for k in range(0, n + 1, 997):
        assert (oracle_small.occ4(k) == occ4[k]).all(), k
    # ranks are monotone and sum to the position
    tot = occ4.sum(axis=1)
    kk = ks - (ks >= np.uint64(gpu_index_full.info.primary))
    assert (tot == kk + 1).all()
