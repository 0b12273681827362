"""FM-index writer compatible with the reference's `bwt_index`.

Produces the five files the reference's loader reads (.bwt .sa .pac .ann .amb) byte-for-byte as
reference src/BWT_Index would (bns_fasta2bntseq bntseq.c:158-211, bwt_bwtupdate_core
bwtindex.c:53-75, bwt_cal_sa bwt.c:101-123, dumps bwt.c:174-196, bntseq.c:59-89), but the suffix
array is built MI355X-style: prefix doubling with device-wide radix sorts (torch.sort on the GPU
when one is present, the same code on the CPU otherwise) instead of BWA's incremental BWT-SW.
The GPU box receives no reference binaries, so every benchmark/test index is made here.
"""
from __future__ import annotations

import numpy as np
import torch

_NT4 = np.full(256, 4, dtype=np.uint8)
for _i, _ch in enumerate(b"ACGT"):
    _NT4[_ch] = _i
    _NT4[_ch + 32] = _i


def read_fasta(path):
    """[(name, comment, ASCII uint8 array)] -- name/comment split like kseq (first whitespace)."""
    out, name, comment, chunks = [], None, "", []
    with open(path, "rb") as fh:
        for line in fh:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    out.append((name, comment, np.frombuffer(b"".join(chunks), dtype=np.uint8)))
                head = line[1:].decode()
                parts = head.split(None, 1)
                name = parts[0] if parts else ""
                comment = parts[1] if len(parts) > 1 else ""
                chunks = []
            elif name is not None:
                chunks.append(line)
    if name is not None:
        out.append((name, comment, np.frombuffer(b"".join(chunks), dtype=np.uint8)))
    return out


def _lrand48_bits(n: int) -> np.ndarray:
    """n successive values of lrand48() & 3 after srand48(11) (glibc: X <- (0x5DEECE66D X + 0xB) mod 2^48,
    X0 = (seed << 16) | 0x330E, result X >> 17); bntseq.c:144,173-174."""
    out = np.empty(n, dtype=np.uint8)
    x = (11 << 16) | 0x330E
    a, c, m = 0x5DEECE66D, 0xB, (1 << 48) - 1
    for i in range(n):
        x = (a * x + c) & m
        out[i] = (x >> 17) & 3
    return out


def pack_contigs(contigs):
    """-> (codes uint8[L] with N replaced like the reference, ann records, amb records)."""
    codes, anns, ambs = [], [], []
    offset = 0
    n_amb_total = sum(int((_NT4[s] > 3).sum()) for _, _, s in contigs)
    rnd = _lrand48_bits(n_amb_total)
    used = 0
    for name, comment, seq in contigs:
        c = _NT4[seq].copy()
        bad = np.nonzero(c > 3)[0]
        n_ambs = 0
        if len(bad):
            # holes = maximal runs of the SAME ambiguous character (bntseq.c:125-141)
            start = 0
            for i in range(1, len(bad) + 1):
                if i == len(bad) or bad[i] != bad[i - 1] + 1 or seq[bad[i]] != seq[bad[i - 1]]:
                    ambs.append((offset + int(bad[start]), int(bad[i - 1] - bad[start] + 1), chr(seq[bad[start]])))
                    n_ambs += 1
                    start = i
            c[bad] = rnd[used:used + len(bad)]
            used += len(bad)
        anns.append((name, comment if comment else "(null)", offset, len(seq), n_ambs))
        codes.append(c)
        offset += len(seq)
    return (np.concatenate(codes) if codes else np.zeros(0, np.uint8)), anns, ambs


def suffix_array(text: torch.Tensor) -> torch.Tensor:
    """Suffix array of text+$ (codes 0..3, $ smallest) by prefix doubling; returns int64[N+1], SA[0]=N."""
    dev = text.device
    n = text.numel()
    t = torch.zeros(n + 1 + 32, dtype=torch.int64, device=dev)
    t[:n] = text.to(torch.int64) + 1                      # 0 = past the end ($ and beyond)
    # initial key: first 24 symbols as a base-5 number
    h = 24
    key = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    for d in range(h):
        key = key * 5 + t[d:d + n + 1]
    rank = _dense_rank(key)
    del key
    big = n + 2
    if big * big >= (1 << 62):
        raise NotImplementedError("text too long for single-key doubling; split-key sort not implemented yet")
    while int(rank.max()) < n:
        nxt = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        if h <= n:
            nxt[: n + 1 - h] = rank[h:] + 1                # 0 for suffixes shorter than h
        rank = _dense_rank(rank * big + nxt)
        h *= 2
    sa = torch.empty(n + 1, dtype=torch.int64, device=dev)
    sa[rank] = torch.arange(n + 1, dtype=torch.int64, device=dev)
    return sa


def suffix_array_bucketed(text: torch.Tensor, k_bucket: int = 3, verbose: bool = False) -> torch.Tensor:
    """Suffix array of text+$ for texts too long for `suffix_array` ((2L)^2 overflowing one 63-bit key, or
    whole-array sorts not fitting in HBM): hg38 has 2L = 6.2 G.  Suffixes are bucketed by their first
    `k_bucket` symbols -- the most significant part of every sort key, so buckets are ordered among
    themselves and sort independently -- and ranks are refined IN PLACE by prefix doubling, one packed key
    (local rank << 34 | rank[i+h]) and one radix sort per unresolved bucket and round.  Reading a rank that a
    previous bucket of the same round already refined is harmless (it only orders by more symbols).
    Everything that touches all 2L positions runs in pieces of <= 2^30 elements (torch.nonzero's limit)."""
    dev = text.device
    n = text.numel()
    n1 = n + 1
    pad = 64
    t8 = torch.zeros(n1 + pad, dtype=torch.uint8, device=dev)
    t8[:n] = text + 1                                     # digit 0 = past the end
    piece = 1 << 30
    nb = 5 ** k_bucket
    # bucket id per suffix, its histogram, and the positions of every bucket
    bucket = torch.empty(n1, dtype=torch.uint8 if nb <= 255 else torch.int16, device=dev)
    for a in range(0, n1, piece):
        b = min(n1, a + piece)
        acc = torch.zeros(b - a, dtype=torch.int16, device=dev)
        for d in range(k_bucket):
            acc = acc * 5 + t8[a + d:b + d]
        bucket[a:b] = acc.to(bucket.dtype)
    counts = _histogram(bucket, nb)
    counts_h = counts.cpu().tolist()
    base = [0] * (nb + 1)
    for b in range(nb):
        base[b + 1] = base[b] + counts_h[b]
    pos_of = [None] * nb
    for b in range(nb):
        if counts_h[b] == 0:
            continue
        parts = []
        for a in range(0, n1, piece):
            e = min(n1, a + piece)
            parts.append(torch.nonzero(bucket[a:e] == b).view(-1) + a)
        pos_of[b] = torch.cat(parts) if len(parts) > 1 else parts[0]
    del bucket
    rank = torch.empty(n1, dtype=torch.int64, device=dev)
    # initial ranks: first 27 symbols as a base-5 number, dense ranks inside the bucket
    h = 27
    resolved = [True] * nb
    for b in range(nb):
        pos = pos_of[b]
        if pos is None:
            continue
        if counts_h[b] == 1:
            rank[pos] = base[b]
            continue
        key = torch.zeros_like(pos)
        for d in range(h):
            key = key * 5 + t8[pos + d]
        srt, order = torch.sort(key)
        step = torch.zeros_like(srt)
        step[1:] = (srt[1:] != srt[:-1]).to(torch.int64)
        dense = torch.cumsum(step, 0)
        rank[pos[order]] = base[b] + dense
        resolved[b] = int(dense[-1]) + 1 == counts_h[b]
        del key, srt, order, step, dense
    del t8
    rounds = 0
    while not all(resolved):
        for b in range(nb):
            if resolved[b]:
                continue
            pos = pos_of[b]
            local = rank[pos] - base[b]
            nx = pos + h
            inside = nx < n1
            nxt = torch.zeros_like(pos)
            nxt[inside] = rank[nx[inside]] + 1            # 0 = suffix shorter than h
            srt, order = torch.sort((local << 34) | nxt)
            step = torch.zeros_like(srt)
            step[1:] = (srt[1:] != srt[:-1]).to(torch.int64)
            dense = torch.cumsum(step, 0)
            rank[pos[order]] = base[b] + dense
            resolved[b] = int(dense[-1]) + 1 == counts_h[b]
            del local, nx, inside, nxt, srt, order, step, dense
        h *= 2
        rounds += 1
        if verbose:
            print(f"[index_build] doubling round {rounds}: h={h}, unresolved buckets={sum(not r for r in resolved)}", flush=True)
    del pos_of
    sa = torch.empty(n1, dtype=torch.int64, device=dev)
    for a in range(0, n1, piece):
        e = min(n1, a + piece)
        sa[rank[a:e]] = torch.arange(a, e, dtype=torch.int64, device=dev)
    return sa


def _histogram(x: torch.Tensor, nbins: int) -> torch.Tensor:
    """bincount in pieces of 2^27 elements (torch.bincount raised SIGFPE on a 2^30-element input on ROCm)."""
    out = torch.zeros(nbins, dtype=torch.int64, device=x.device)
    step = 1 << 27
    for a in range(0, x.numel(), step):
        out += torch.bincount(x[a:a + step].to(torch.int64), minlength=nbins)
    return out


def _dense_rank(key: torch.Tensor) -> torch.Tensor:
    srt, idx = torch.sort(key)
    step = torch.zeros_like(srt)
    step[1:] = (srt[1:] != srt[:-1]).to(torch.int64)
    r = torch.cumsum(step, 0)
    out = torch.empty_like(r)
    out[idx] = r
    return out


def build_index(fasta: str, prefix: str, device: str | None = None, bucketed: bool | None = None) -> dict:
    """Write <prefix>.{bwt,sa,pac,ann,amb}; returns {'l_pac','seq_len','primary'}."""
    contigs = read_fasta(fasta)
    fwd, anns, ambs = pack_contigs(contigs)
    return build_index_from_codes(fwd, anns, ambs, prefix, device, bucketed)


def build_index_from_codes(fwd: np.ndarray, anns, ambs, prefix: str, device: str | None = None, bucketed: bool | None = None,
                           verbose: bool = False) -> dict:
    """fwd: forward-strand codes 0..3 (uint8); anns/ambs as pack_contigs returns them."""
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    L = len(fwd)
    N = 2 * L
    text = torch.from_numpy(np.concatenate([fwd, (3 - fwd[::-1])]).astype(np.uint8)).to(device)
    if bucketed is None:
        bucketed = (N + 2) * (N + 2) >= (1 << 62)
    sa = suffix_array_bucketed(text, verbose=verbose) if bucketed else suffix_array(text)
    piece = 1 << 30
    primary = -1
    for a in range(0, N + 1, piece):
        hit = torch.nonzero(sa[a:min(N + 1, a + piece)] == 0)
        if hit.numel():
            primary = a + int(hit[0, 0])
    # BWT[i] = text[SA[i] - 1] with the $ row (SA = 0) dropped: N symbols
    bwt = torch.empty(N, dtype=torch.uint8, device=text.device)
    for a in range(0, N + 1, piece):
        e = min(N + 1, a + piece)
        s_ = sa[a:e]
        lo = a - (1 if a > primary else 0)                   # rows after the primary shift down by one
        rows = s_ if not (a <= primary < e) else torch.cat([s_[:primary - a], s_[primary - a + 1:]])
        bwt[lo:lo + rows.numel()] = text[rows - 1]
    n_sa = (N + 32) // 32
    samples = sa[torch.arange(1, n_sa, device=sa.device) * 32].cpu().numpy().astype(np.uint64)
    del sa
    counts = _histogram(text, 4)
    del text
    L2 = np.zeros(5, dtype=np.uint64)
    L2[1:] = np.cumsum(counts.cpu().numpy()).astype(np.uint64)

    # 128-symbol blocks: running counts before the block, then 8 words of 16 symbols (MSB first)
    n_blocks = (N + 127) // 128
    pad = n_blocks * 128 - N
    bwt = torch.cat([bwt, torch.full((pad,), 255, dtype=bwt.dtype, device=bwt.device)])    # 255 = "no symbol"
    per_block = torch.empty(n_blocks, 4, dtype=torch.int64, device=bwt.device)
    words = torch.empty(n_blocks, 8, dtype=torch.int64, device=bwt.device)
    shifts = (30 - 2 * torch.arange(16, device=bwt.device)).to(torch.int64)
    bpiece = 1 << 22                                          # blocks per piece (2^29 symbols)
    for a in range(0, n_blocks, bpiece):
        e = min(n_blocks, a + bpiece)
        bw = bwt[a * 128:e * 128].view(e - a, 128)
        for c in range(4):
            per_block[a:e, c] = (bw == c).sum(1)
        words[a:e] = ((bw.view(e - a, 8, 16).to(torch.int64) & 3) * (bw.view(e - a, 8, 16) != 255) << shifts).sum(2)
    before = torch.cumsum(per_block, 0) - per_block
    inter = torch.zeros(n_blocks, 16, dtype=torch.int64, device=bwt.device)
    inter[:, 0:8:2] = before & 0xFFFFFFFF
    inter[:, 1:8:2] = before >> 32
    inter[:, 8:] = words
    inter = inter.view(-1).cpu().numpy().astype(np.uint32)
    raw_words = (N + 15) // 16
    last_syms = raw_words - (n_blocks - 1) * 8              # symbol words of the last (maybe partial) block
    body = inter[: (n_blocks - 1) * 16 + 8 + last_syms]
    total = per_block.sum(0).cpu().numpy().astype(np.uint64)
    with open(prefix + ".bwt", "wb") as fh:
        fh.write(np.uint64(primary).tobytes())
        fh.write(L2[1:].tobytes())
        fh.write(body.tobytes())
        fh.write(total.tobytes())
    with open(prefix + ".sa", "wb") as fh:
        fh.write(np.uint64(primary).tobytes())
        fh.write(L2[1:].tobytes())
        fh.write(np.uint64(32).tobytes())
        fh.write(np.uint64(N).tobytes())
        fh.write(samples.tobytes())
    # forward-only .pac (bntseq.c:192-205)
    padded = np.concatenate([fwd, np.zeros((-L) % 4, dtype=np.uint8)]).reshape(-1, 4)
    pac = (padded[:, 0] << 6 | padded[:, 1] << 4 | padded[:, 2] << 2 | padded[:, 3]).astype(np.uint8)
    with open(prefix + ".pac", "wb") as fh:
        fh.write(pac.tobytes())
        if L % 4 == 0:
            fh.write(b"\0")
        fh.write(bytes([L % 4]))
    with open(prefix + ".ann", "w") as fh:
        fh.write(f"{L} {len(anns)} 11\n")
        for name, anno, off, ln, n_ambs in anns:
            fh.write(f"0 {name} {anno}\n{off} {ln} {n_ambs}\n")
    with open(prefix + ".amb", "w") as fh:
        fh.write(f"{L} {len(anns)} {len(ambs)}\n")
        for off, ln, ch in ambs:
            fh.write(f"{off} {ln} {ch}\n")
    return {"l_pac": L, "seq_len": N, "primary": primary}
