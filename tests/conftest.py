import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SMALL_PREFIX = os.path.join(GOLDEN, "idx", "small")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "hotpath_small.npz"), allow_pickle=True)


@pytest.fixture(scope="session")
def oracle_small():
    from oracle import oracle as O
    o = O.Oracle(SMALL_PREFIX)
    yield o
    o.close()


@pytest.fixture(scope="session")
def built_lib():
    """libkart_amd.so, built in-tree if it is not there yet (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from kart_amd import api
    return api.load_library()


def _have_gpu():
    try:
        from kart_amd import api
        return api.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_index(built_lib):
    from kart_amd import api
    if api.device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (there is no CPU fallback)")
    ix = api.Index(SMALL_PREFIX, 0, api.KG_SA_SAMPLED)
    yield ix
    ix.close()


@pytest.fixture(scope="session")
def gpu_index_full(built_lib):
    from kart_amd import api
    if api.device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (there is no CPU fallback)")
    ix = api.Index(SMALL_PREFIX, 0, api.KG_SA_FULL)
    yield ix
    ix.close()


def _dense_index(built_lib, mode_name):
    from kart_amd import api
    if api.device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (there is no CPU fallback)")
    return api.Index(SMALL_PREFIX, 0, getattr(api, mode_name))


@pytest.fixture(scope="session")
def gpu_index_dense4(built_lib):
    ix = _dense_index(built_lib, "KG_SA_DENSE4")       # the smaller index: every 4th suffix-array entry resident
    yield ix
    ix.close()


@pytest.fixture(scope="session")
def gpu_index_compact(built_lib):
    ix = _dense_index(built_lib, "KG_SA_FULL40")       # the compact index: whole suffix array, a quarter of the q-mer table, no triple planes
    yield ix
    ix.close()


@pytest.fixture(scope="session")
def gpu_index_wide(built_lib):
    ix = _dense_index(built_lib, "KG_SA_FULL40_WIDE")  # 5-byte suffix-array entries where the text needs them, the full q-mer table, triple planes: what KG_SA_AUTO takes for a human-sized text
    yield ix
    ix.close()


@pytest.fixture(scope="session")
def gpu_index_dense8(built_lib):
    ix = _dense_index(built_lib, "KG_SA_DENSE8")
    yield ix
    ix.close()


def split(arr, off):
    return [arr[off[i]:off[i + 1]] for i in range(len(off) - 1)]
