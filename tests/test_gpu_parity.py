"""-m gpu: the hipcc-built library (through the C ABI) against the golden vectors, the C oracle and,
when it travelled with the repo, the compiled reference itself."""
import os

import numpy as np
import pytest

import t4check
from t4libs import REF_FA, ROOT, Oracle, Ref, Synth, rows_to_strs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    os.environ.pop("T4_LIB", None)
    import trust4_amd
    import trust4_amd.build
    trust4_amd.build.build()
    assert trust4_amd.lib_path().endswith("trust4_amd/libt4hip.so")
    return trust4_amd.Engine(0)


@pytest.fixture(scope="module")
def ref_index(eng):
    return eng.index(9).set_params(17, 10, 0.9).load_ref_fasta(REF_FA).commit()


@pytest.fixture(scope="module")
def oracle():
    return Oracle(9, REF_FA, 17)


def test_golden_all(eng, ref_index):
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_query_k9.npz"))
    reads = [str(x) for x in g["reads"]]
    b = eng.upload(reads)
    ann = ref_index.annotate_rough(b)
    cnt, ov = ref_index.overlaps(b, 0, 0, 128)
    off, hits = ref_index.hits(b)
    for i in range(len(reads)):
        for t in range(4):
            assert t4check.ann_equal(ann[i, t], tuple(g["annotate"][i, t].tolist())), (i, t)
        exp = g["overlaps"][g["overlap_off"][i]:g["overlap_off"][i + 1]]
        assert cnt[i] == len(exp) or (cnt[i] == -1 and len(exp) == 0)
        assert [tuple(x) for x in ov[i, :len(exp)].tolist()] == [tuple(x) for x in exp.tolist()], i
        if i % 8 == 0:
            m = t4check.hits_as_sorted_rows(off, hits, i)
            assert (m == g["hits"][g["hit_off"][i]:g["hit_off"][i + 1]]).all(), i


def test_synthetic_20k_vs_oracle(eng, ref_index, oracle):
    arr = Synth(2000, 21).next_reads(10000)
    b = eng.upload(arr)
    ann = ref_index.annotate_rough(b)
    exp, hp, tot = oracle.annotate_batch(arr, arr.shape[1], arr.shape[0])
    assert eng.stats()["total_hits"] == tot           # H_r of SURVEY 8(d) is a parity quantity
    ok = (ann["seqIdx"] == exp["seqIdx"])
    assert ok.all()
    m = exp["seqIdx"] != -1
    for f in ("readStart", "readEnd", "seqStart", "seqEnd", "strand", "matchCnt", "indelCnt", "similarity"):
        assert (ann[f][m] == exp[f][m]).all(), f


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")
def test_vs_compiled_reference(eng, ref_index):
    r = Ref(9, REF_FA, 17)
    reads = rows_to_strs(Synth(500, 33).next_reads(500))
    b = eng.upload(reads)
    ann = ref_index.annotate_rough(b)
    assert t4check.check_annotate(ann, reads, r) == []


def test_hits_overlaps_skip_repeats(eng, ref_index, oracle):
    reads = rows_to_strs(Synth(400, 5).next_reads(300)) + ["A" * 150, "ACGTACGTA", "", "N" * 30]
    b = eng.upload(reads)
    for sk in (0, 1):
        off, hits = ref_index.hits(b, 0, sk)
        assert t4check.check_hits(off, hits, reads, oracle, allow_total_skip=sk) == []
        cnt, ov = ref_index.overlaps(b, 0, sk, 128)
        assert t4check.check_overlaps(cnt, ov, reads, oracle, skip_repeats=sk) == []
    for strand in (1, -1):
        off, hits = ref_index.hits(b, strand, 0)
        assert t4check.check_hits(off, hits, reads, oracle, strand=strand) == []


@pytest.mark.parametrize("k", [9, 11, 17])
def test_novel_sets(eng, k):
    from test_engine_emu import run_novel_case
    run_novel_case(eng, 100 + k, k)


def test_novel_sets_barcoded(eng):
    from test_engine_emu import run_novel_case
    run_novel_case(eng, 77, 9, barcodes=True, hit_len=13)


def test_full_size_properties(eng, ref_index):
    """C2-scale (200k pairs here to bound time) size-independent properties: determinism across runs and
    invariance to batch composition (a read's result does not depend on its neighbours)."""
    arr = Synth(20000, 1).next_reads(200000)
    b = eng.upload(arr)
    a1 = ref_index.annotate_rough(b)
    h1 = eng.stats()["total_hits"]
    a2 = ref_index.annotate_rough(b)
    assert eng.stats()["total_hits"] == h1
    assert (a1["seqIdx"] == a2["seqIdx"]).all() and (a1["matchCnt"] == a2["matchCnt"]).all()
    perm = np.random.RandomState(0).permutation(arr.shape[0])[:50000]
    b2 = eng.upload(arr[perm])
    a3 = ref_index.annotate_rough(b2)
    assert (a3["seqIdx"] == a1["seqIdx"][perm]).all()
    m = a3["seqIdx"] != -1
    assert (a3["matchCnt"][m] == a1["matchCnt"][perm][m]).all()
    assert (a3["readStart"][m] == a1["readStart"][perm][m]).all()
    # work distribution (next read from a per-launch counter vs a static stride) is not observable either
    os.environ["T4_STATIC_STRIDE"] = "1"
    try:
        a4 = ref_index.annotate_rough(b)
    finally:
        os.environ.pop("T4_STATIC_STRIDE", None)
    assert (a4["seqIdx"] == a1["seqIdx"]).all() and (a4["matchCnt"] == a1["matchCnt"]).all() and (a4["readEnd"] == a1["readEnd"]).all()


def test_gap_dp_vs_oracle(eng):
    from test_engine_emu import check_equal_length_shortcut, check_gap_dp
    check_gap_dp(eng, 5, 20000)
    check_equal_length_shortcut(eng, 50000)


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")
@pytest.mark.parametrize("seed,n_pairs,n_clones", [(1, 150, 12), (3, 500, 30), (4, 400, 6)])
def test_add_path_matches_reference(eng, tmp_path, seed, n_pairs, n_clones):
    """Ordered contig builder: every AddRead / RepeatAddRead / InputNovelRead return value and the final _raw.out
    records equal the unmodified reference SeqSet driven with the same calls."""
    from test_assembler_emu import run_case
    run_case(eng, tmp_path, seed, n_pairs, n_clones)


def test_process_pairs_vs_oracle(eng):
    from test_engine_emu import check_process_pairs, check_process_pairs_edges
    check_process_pairs(eng, 5, 20000)
    check_process_pairs_edges(eng)


def test_mate_overlap_vs_oracle(eng):
    from test_engine_emu import check_mate_overlap
    check_mate_overlap(eng, 3, 20000)


def test_has_hit_vs_oracle(eng):
    from test_engine_emu import check_has_hit
    check_has_hit(eng, 5)
    check_has_hit(eng, 9, hit_lens=(31,))


def test_novel_min_statistics(eng):
    from test_engine_emu import check_novel_min_statistics
    check_novel_min_statistics(eng)
    check_novel_min_statistics(eng, seed=8, n_contigs=260)
    check_novel_min_statistics(eng, seed=9, n_contigs=170, repeats=True)    # the repeat-skip rule over tandem repeats (bit-mask replay in seedPositions)
    check_novel_min_statistics(eng, seed=10, n_contigs=300, repeats=True)   # up to 200 overlaps per read


def test_long_reads_and_limits(eng):
    from test_engine_emu import check_long_reads_and_limits
    check_long_reads_and_limits(eng, n=400)


def test_read_with_more_than_32768_hits(eng):
    """A gene segment that thousands of contigs share (50 036 hits for one k = 17 AssignRead query at only 100 k
    pairs; the C gene of a chain sits in every contig of that chain): one read pass emits far more than the 32 768 hits /
    4 096 overlaps round 1 refused. 2 400 contigs = a shared 110-bp stretch with private flanks; reads from the shared stretch.
    hits, overlaps (> 128 per read), ExtendOverlap and AssignRead against the oracle."""
    import random
    rnd = random.Random(21)
    rand = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    k = 9
    shared = rand(110)
    o = Oracle(k)
    ix = eng.index(k)
    for i in range(2400):
        seq = rand(rnd.randint(20, 60)) + shared + rand(rnd.randint(20, 60))
        a = o.add_novel("IGHC%d" % i, seq, 1, -1, None)
        assert a == ix.add_contig("IGHC%d" % i, seq, -1, None)
    o.set_hit_len_required(31)
    ix.set_params(31, 10, 0.9).commit()
    reads = [shared[5:105], shared, rand(25) + shared[:90], shared[30:] + rand(30)]
    b = eng.upload(reads)
    off, hits = ix.hits(b, 0, 0)
    assert int(off[-1]) > 4 * 32768, int(off[-1])
    assert t4check.check_hits(off, hits, reads, o) == []
    cnt, ov = ix.overlaps(b, 0, 0, 4096)
    assert int(cnt.max()) > 2000, cnt
    assert t4check.check_overlaps(cnt, ov, reads, o) == []
    aret, aout = ix.assign(b, 0)
    for i, rd in enumerate(reads):
        eret, eout = o.assign_read(rd, 0, -1)
        assert int(aret[i]) == eret, (i, eret, int(aret[i]))
        if eret != -1:
            assert tuple(aout[i].tolist()) == tuple(eout)


def test_apply_delta_through_the_abi(eng):
    from test_engine_emu import check_apply_delta
    check_apply_delta(eng)
    check_apply_delta(eng, seed=32, k=11)


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not shipped")
def test_kmer_length_growth_in_lock_step(eng, tmp_path):
    from test_assembler_emu import run_case
    run_case(eng, tmp_path, 7, 400, 25, window=48, grow_at=(420, 11))

