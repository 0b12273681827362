"""Error behaviour of the C ABI on a real device: status codes + kg_last_error instead of the reference's exit(1);
a failed call must leave the handles usable."""
import os
import shutil

import numpy as np
import pytest
import torch

from conftest import ROOT
from kart_amd import api

pytestmark = pytest.mark.gpu
SMALL = os.path.join(ROOT, "tests", "golden", "idx", "small")


def test_index_load_errors(tmp_path):
    with pytest.raises(api.KartAmdError, match=r"status 2"):                    # KG_ERR_IO: no such index
        api.Index(str(tmp_path / "nothing"), 0, api.KG_SA_FULL)
    with pytest.raises(api.KartAmdError, match=r"status 3.*sa_mode"):           # KG_ERR_ARG
        api.Index(SMALL, 0, 7)
    with pytest.raises(api.KartAmdError, match=r"status 3.*out of range"):
        api.Index(SMALL, 99, api.KG_SA_FULL)
    for ext in (".bwt", ".sa", ".pac", ".ann", ".amb"):
        shutil.copy(SMALL + ext, str(tmp_path / ("cut" + ext)))
    with open(tmp_path / "cut.bwt", "r+b") as fh:
        fh.truncate(1000)
    with pytest.raises(api.KartAmdError, match=r"status 2"):                    # truncated .bwt
        api.Index(str(tmp_path / "cut"), 0, api.KG_SA_FULL)
    ix = api.Index(SMALL, 0, api.KG_SA_FULL)                                    # and the library still works afterwards
    assert ix.n_contigs == 4
    ix.close()


def test_seed_batch_capacity_errors(golden, gpu_index):
    ws = api.Workspace(gpu_index, 1024, 1 << 16)
    enc, off = golden["fast_enc"], golden["fast_off"]
    big_off = np.arange(0, 150 * 2001, 150, dtype=np.int64)
    with pytest.raises(api.KartAmdError, match=r"status 4.*reads exceed"):      # KG_ERR_CAPACITY
        ws.seed_batch(np.zeros(150 * 2000, np.uint8), big_off, 0)
    with pytest.raises(api.KartAmdError, match=r"status 4.*bases exceed"):
        ws.seed_batch(np.zeros(1 << 17, np.uint8), np.array([0, 1 << 17], dtype=np.int64), 0)
    # device form with a seed buffer that is too small: the call reports how many seeds it needs, nothing is written past the buffer
    n = 200
    d_enc = torch.from_numpy(enc[: off[n]].copy()).cuda()
    d_off = torch.from_numpy(off[: n + 1].copy()).cuda()
    need = int(golden["fast_seed_off"][n])
    cap = need // 2
    d_so = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    d_seeds = torch.full(((cap + 8) * 16,), 0xAB, dtype=torch.uint8, device="cuda")
    ws.seed_batch_device(d_enc.data_ptr(), d_off.data_ptr(), n, int(off[n]), d_so.data_ptr(), d_seeds.data_ptr(), cap, 0)
    torch.cuda.synchronize()
    assert ws.overflow() == need
    assert (d_seeds[cap * 16:] == 0xAB).all()
    # the host form grows its own buffer and is unaffected by the failed calls above
    so, seeds = ws.seed_batch(enc[: off[n]], off[: n + 1], 0)
    assert int(so[n]) == need and (seeds == golden["fast_seeds"][:need].astype(api.SEED_DT)).all()
    ws.close()


def test_nw_fragment_beyond_the_lds_limit_is_not_an_error(gpu_index, oracle_small):
    """fragments longer than 7000 bases used to be rejected (and the host pipeline then exited); the reference's
    nw_alignment has no limit, and neither has the kernel path with the boundary column in HBM"""
    a, b = b"A" * 7001, b"A" * 10
    assert gpu_index.nw_alignment([(a, b)]) == [oracle_small.nw(a, b)]
    assert gpu_index.nw_alignment([(b"ACGT", b"ACGT")]) == [(b"ACGT", b"ACGT")]
