"""The read generator behind bench.py and the hg38-size tests (benchkit/reads.py: wgsim's model restated with torch) on the CPU:
the materialised haplotypes against a base-by-base replay of their own event lists, the coordinate map, the rates, the record files."""
import numpy as np
import pytest
import torch

from benchkit import reads as R


def _replay(codes, hp, h):
    """haplotype h, one reference position at a time, from the event list and the substituted positions alone"""
    ep, el, ed = hp.ev_pos[h].tolist(), hp.ev_len[h].tolist(), hp.ev_del[h].tolist()
    ev = {p: (l, d) for p, l, d in zip(ep, el, ed)}
    out, to_hap = [], {}
    i, L = 0, len(codes)
    while i < L:
        if i in ev and ev[i][1]:
            i += ev[i][0]
            continue
        to_hap[i] = len(out)
        out.append(None)                        # the base at i (possibly substituted): compared through the map below
        if i in ev:
            out.extend([-1] * ev[i][0])         # inserted bases
        i += 1
    return out, to_hap


@pytest.mark.parametrize("seed", [1, 2])
def test_haplotypes_are_the_reference_with_their_events_applied(seed):
    dev = torch.device("cpu")
    g = torch.Generator().manual_seed(seed)
    codes = torch.randint(0, 4, (60_000,), generator=g, dtype=torch.uint8)
    hp = R.Haplotypes(codes, dev, seed=seed, mut_rate=0.02, indel_frac=0.4)
    assert hp.n_indel_sites > 100
    for h in (0, 1):
        shape, to_hap = _replay(codes.tolist(), hp, h)
        seq = hp.seq[hp.base[h]:hp.base[h] + hp.hap_len[h]]
        assert len(shape) == hp.hap_len[h]
        kept = np.array(sorted(to_hap), dtype=np.int64)
        want = np.array([to_hap[p] for p in kept], dtype=np.int64)
        got = hp.to_hap(h, torch.from_numpy(kept)).numpy()
        assert (got == want).all()
        # a kept position holds the reference base unless it is one of the (few) substituted ones
        same = (seq[torch.from_numpy(want)] == codes[torch.from_numpy(kept)]).float().mean().item()
        assert same > 0.97
        # a deleted position maps to the first kept one behind it
        ep, el, ed = hp.ev_pos[h], hp.ev_len[h], hp.ev_del[h]
        dels = ep[ed]
        if dels.numel():
            behind = dels + el[ed]
            assert (hp.to_hap(h, dels) == hp.to_hap(h, behind)).all()


def test_rates_are_wgsims():
    dev = torch.device("cpu")
    g = torch.Generator().manual_seed(3)
    codes = torch.randint(0, 4, (4_000_000,), generator=g, dtype=torch.uint8)
    hp = R.Haplotypes(codes, dev)
    assert abs(hp.n_sites / 4e6 - R.MUT_RATE) < 1e-4
    assert abs(hp.n_indel_sites / hp.n_sites - R.INDEL_FRAC) < 0.02
    # a haplotype carries a hom site always and a het site half the time: 2/3 of the sites
    assert abs(hp.ev_pos[0].numel() / hp.n_indel_sites - 2 / 3) < 0.05
    dl = hp.ev_len[0][hp.ev_del[0]].float()
    assert abs(dl.mean().item() - 1 / 0.7) < 0.15          # 1 + geometric extensions at 0.3
    il = hp.ev_len[0][~hp.ev_del[0]]
    assert int(il.max()) <= 4 and int(il.min()) >= 1
    enc, off, starts, st = R.gen_reads_device(codes, 40_000, seed=5, err=0.01, dev=dev, want_meta=True)
    R.release_haplotypes()
    # ~ 2 * 150 * 0.001 * 0.15 * (2/3): the share of pairs whose windows hold an indel of their haplotype
    assert 0.02 < st["pairs_with_indel"] / st["pairs"] < 0.045
    assert enc.numel() == 2 * 40_000 * 150 and int(enc.max()) <= 4


def test_records_are_wgsims_and_prefixes_are_whole_records(tmp_path):
    dev = torch.device("cpu")
    g = torch.Generator().manual_seed(4)
    codes = torch.randint(0, 4, (300_000,), generator=g, dtype=torch.uint8)
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    st = R.write_fastq_pairs(codes, 3000, 9, f1, f2, dev, err=0.02)
    l1, l2 = open(f1, "rb").read().split(b"\n"), open(f2, "rb").read().split(b"\n")
    assert len(l1) == len(l2) == 4 * 3000 + 1 and l1[-1] == b""
    assert l1[0].startswith(b"@0_Pos=") and l1[0].endswith(b"\t/1") and l2[0].endswith(b"\t/2") and l1[4 * 2999].startswith(b"@2999_Pos=")
    assert all(len(x) == 150 for x in l1[1::4]) and set(l1[3]) == {ord("2")}           # -e 0.02 -> Q 17 -> '2'
    assert len({len(x) for x in l1[0::4] if x}) > 1                                       # names of varying width
    assert st["fastq_bytes"] == sum(len(x) + 1 for x in l1[:-1]) + sum(len(x) + 1 for x in l2[:-1])
    # error-free mates are pieces of a haplotype: mate 1 as written, or its reverse complement, occurs in hap 0 or hap 1
    st0 = R.write_fastq_pairs(codes, 200, 9, f1, f2, dev, err=0.0)
    hp = R.haplotypes_of(codes, dev)
    text = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[hp.seq.numpy()])
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for r in open(f1, "rb").read().split(b"\n")[1::4]:
        assert r in text or r.translate(comp)[::-1] in text
    g1 = str(tmp_path / "p.fq")
    assert R.copy_records(f1, g1, 50) == 50
    assert open(g1, "rb").read() == b"\n".join(open(f1, "rb").read().split(b"\n")[:200]) + b"\n"
    assert R.copy_records(f1, g1, 30, skip_records=170) == 30
    assert open(g1, "rb").read() == b"\n".join(open(f1, "rb").read().split(b"\n")[680:800]) + b"\n"
    fl = str(tmp_path / "long.fq")
    stl = R.write_long_reads(codes, 50, 7000, 31, fl, dev, err=0.15)
    ll = open(fl, "rb").read().split(b"\n")
    assert len(ll) == 201 and all(len(x) == 7000 for x in ll[1::4]) and set(ll[3]) == {ord(")")}
    assert stl["reads_with_indel"] > 10                                                    # ~ 1 - exp(-7000 * 0.001 * 0.15 * 2/3) = 50 %
    ss, sl = R.record_starts(fl)
    assert len(ss) == 50 and (sl == 7000).all()
    R.release_haplotypes()
