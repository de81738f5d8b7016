"""-m gpu, last in the suite: what was written after the GPU budget of round 1 was spent and has only run in the emulator so far
(plus the t4_kmer_count_* checks that did run on the MI355X) -- t4_kmer_count_* of the hipcc build against the C oracle (same checks as
tests/test_kmer_count_emu.py, larger), the driver with the opt-in device host phases, bam-extractor-hip against the reference binary."""
import os

import pytest

from test_kmer_count_emu import check_engine_kmer_counts, check_kmer_count_merge, check_per_barcode_counts

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


def test_per_barcode_counts_gpu(eng):
    check_per_barcode_counts(eng, 23, 2000, n_barcodes=40)


def test_kmer_count_export_merge_gpu(eng):
    check_kmer_count_merge(eng, 24, 3000, parts=4)


def test_driver_with_device_host_phases_gpu(tmp_path):
    """trust4-hip with T4_GPU_KMERCOUNT=1 T4_GPU_MATEOVERLAP=1 (opt-in this round): same files as without the switches and, when
    the reference binary travelled, as the reference's."""
    import filecmp
    import subprocess
    from test_stage1_e2e import REF_BIN, _driver, _gunzip
    from t4libs import REF_FA, ROOT
    import t4libs
    t4libs.build_checkers()
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    pre = str(tmp_path / "c5")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, "3000", "0", "4", pre, "--cells", "60"], check=True, stdout=subprocess.DEVNULL)
    args = ["-t", "4", "-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"]
    outs = {}
    for tag, env in (("host", {}), ("dev", {"T4_GPU_KMERCOUNT": "1", "T4_GPU_MATEOVERLAP": "1"})):
        outs[tag] = str(tmp_path / tag)
        subprocess.run([_driver()] + args + ["-o", outs[tag]], check=True, env=dict(os.environ, **env), stderr=subprocess.DEVNULL)
    if os.path.exists(REF_BIN):
        outs["ref"] = str(tmp_path / "ref")
        subprocess.run([REF_BIN] + args + ["-o", outs["ref"]], check=True, stderr=subprocess.DEVNULL)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        for tag in outs:
            assert filecmp.cmp(outs["host"] + suffix, outs[tag] + suffix, shallow=False), (tag, suffix)


def test_bam_extractor_synthetic_gpu(tmp_path):
    from test_bam_extractor import REF_BAMX, synthetic_cases
    if not os.path.exists(REF_BAMX):
        pytest.skip("oracle/_ref/bam-extractor not shipped")
    import trust4_amd.build as b
    from t4libs import ROOT
    b.build()
    synthetic_cases(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "bam-extractor-hip"), 400)
