"""-m gpu: t4_kmer_count_* of the hipcc build against the C oracle (same checks as tests/test_kmer_count_emu.py, larger)."""
import os

import pytest

from test_kmer_count_emu import check_engine_kmer_counts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    os.environ.pop("T4_LIB", None)
    import trust4_amd
    import trust4_amd.build
    trust4_amd.build.build()
    return trust4_amd.Engine(0)


def test_kmer_counts_and_stats_gpu(eng):
    check_engine_kmer_counts(eng, 21, 4000)


def test_kmer_count_k31_gpu(eng):
    check_engine_kmer_counts(eng, 22, 600, k=31)
