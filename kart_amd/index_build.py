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


def _dense_rank(key: torch.Tensor) -> torch.Tensor:
    srt, idx = torch.sort(key)
    step = torch.zeros_like(srt)
    step[1:] = (srt[1:] != srt[:-1]).to(torch.int64)
    r = torch.cumsum(step, 0)
    out = torch.empty_like(r)
    out[idx] = r
    return out


def build_index(fasta: str, prefix: str, device: str | None = None) -> dict:
    """Write <prefix>.{bwt,sa,pac,ann,amb}; returns {'l_pac','seq_len','primary'}."""
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    contigs = read_fasta(fasta)
    fwd, anns, ambs = pack_contigs(contigs)
    L = len(fwd)
    N = 2 * L
    text = torch.from_numpy(np.concatenate([fwd, (3 - fwd[::-1])]).astype(np.uint8)).to(device)
    sa = suffix_array(text)
    primary = int(torch.nonzero(sa == 0)[0, 0])
    keep = sa != 0
    bwt = text[(sa[keep] - 1)]                              # N symbols, the $ row dropped
    counts = torch.bincount(text.to(torch.int64), minlength=4)
    L2 = np.zeros(5, dtype=np.uint64)
    L2[1:] = np.cumsum(counts.cpu().numpy()).astype(np.uint64)

    # 128-symbol blocks: running counts before the block, then 8 words of 16 symbols (MSB first)
    n_blocks = (N + 127) // 128
    pad = n_blocks * 128 - N
    bw = torch.cat([bwt, torch.zeros(pad, dtype=bwt.dtype, device=bwt.device)]).view(n_blocks, 128)
    valid = (torch.arange(n_blocks * 128, device=bwt.device) < N).view(n_blocks, 128)
    per_block = torch.stack([((bw == c) & valid).sum(1) for c in range(4)], 1).to(torch.int64)   # (n_blocks, 4)
    before = torch.cumsum(per_block, 0) - per_block
    shifts = (30 - 2 * torch.arange(16, device=bwt.device)).to(torch.int64)
    words = (bw.view(n_blocks, 8, 16).to(torch.int64) << shifts).sum(2)                            # (n_blocks, 8)
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
    n_sa = (N + 32) // 32
    samples = sa[torch.arange(1, n_sa, device=sa.device) * 32].cpu().numpy().astype(np.uint64)
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
