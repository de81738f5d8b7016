"""Kernel logic on the CPU: the SAME kernel sources (trust4_amd/csrc) compiled against the fiber
emulator in tests/hipemu, compared with the golden vectors and the oracle. This is test
infrastructure for a container without a GPU -- the `-m gpu` suite runs the same checks on the real
hipcc build."""
import os
import random

import numpy as np
import pytest

import t4check
from t4libs import REF_FA, ROOT, Oracle, Synth, rows_to_strs


@pytest.fixture(scope="module")
def emu_engine():
    os.environ["T4_LIB"] = t4check.build_emulator_lib()
    import trust4_amd
    eng = trust4_amd.Engine(0)
    yield eng
    os.environ.pop("T4_LIB", None)


@pytest.fixture(scope="module")
def ref_index(emu_engine):
    return emu_engine.index(9).set_params(17, 10, 0.9).load_ref_fasta(REF_FA).commit()


@pytest.fixture(scope="module")
def oracle():
    return Oracle(9, REF_FA, 17)


def test_ref_set_matches_oracle(ref_index, oracle):
    assert ref_index.size() == oracle.size() == 615
    for i in range(oracle.size()):
        assert ref_index.name(i) == oracle.name(i)
        assert ref_index.consensus(i) == oracle.consensus(i)


def test_golden_subset(emu_engine, ref_index):
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_query_k9.npz"))
    reads = [str(x) for x in g["reads"]]
    sel = list(range(0, 400, 7)) + list(range(800, len(reads)))  # synthetic sample + every edge case
    b = emu_engine.upload([reads[i] for i in sel])
    ann = ref_index.annotate_rough(b)
    cnt, ov = ref_index.overlaps(b, 0, 0, 128)
    for k, i in enumerate(sel):
        for t in range(4):
            assert t4check.ann_equal(ann[k, t], tuple(g["annotate"][i, t].tolist())), (i, t)
        exp = g["overlaps"][g["overlap_off"][i]:g["overlap_off"][i + 1]]
        assert cnt[k] == len(exp) or (cnt[k] == -1 and len(exp) == 0)
        assert [tuple(x) for x in ov[k, :len(exp)].tolist()] == [tuple(x) for x in exp.tolist()], i


def test_hits_and_skip_repeats(emu_engine, ref_index, oracle):
    reads = rows_to_strs(Synth(200, 5).next_reads(20)) + ["A" * 150, "ACGTACGTA", "", "N" * 30]
    b = emu_engine.upload(reads)
    for sk in (0, 1):
        off, hits = ref_index.hits(b, 0, sk)
        assert t4check.check_hits(off, hits, reads, oracle, allow_total_skip=sk) == []
        cnt, ov = ref_index.overlaps(b, 0, sk, 128)
        assert t4check.check_overlaps(cnt, ov, reads, oracle, skip_repeats=sk) == []
    for strand in (1, -1):
        off, hits = ref_index.hits(b, strand, 0)
        assert t4check.check_hits(off, hits, reads, oracle, strand=strand) == []


def make_novel_case(seed, k, n_contigs=30, barcodes=False):
    """Contigs cut from synthetic transcripts with random per-base weights + reads drawn from them."""
    rnd = random.Random(seed)
    frags = rows_to_strs(Synth(60, seed).next_reads(n_contigs))[:n_contigs]
    contigs = []
    for i, f in enumerate(frags):
        w = np.zeros((len(f), 4), dtype=np.int32)
        for j, c in enumerate(f):
            bidx = "ACGT".index(c)
            w[j, bidx] = rnd.randint(1, 20)
            if rnd.random() < 0.1:
                w[j, (bidx + 1) % 4] = rnd.randint(0, 12)
        contigs.append(("c%d" % i, f, (i % 3) if barcodes else -1, w))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    reads, bcs = [], []
    for ci, (_, c, bc, _) in enumerate(contigs):
        for _ in range(3):
            st = rnd.randint(0, 60)
            rd = list(c[st: st + rnd.randint(60, 110)])
            for _ in range(rnd.randint(0, 3)):
                rd[rnd.randrange(len(rd))] = rnd.choice("ACGTN")
            if rnd.random() < 0.3:
                rd = rd + [rnd.choice("ACGT") for _ in range(rnd.randint(1, 30))]
            if rnd.random() < 0.3:
                rd = [rnd.choice("ACGT") for _ in range(rnd.randint(1, 30))] + rd
            rd = "".join(rd)
            if rnd.random() < 0.5:
                rd = "".join(comp[x] for x in reversed(rd))
            reads.append(rd)
            bcs.append(bc if rnd.random() < 0.8 else (bc + 1) % 3 if barcodes else -1)
    return contigs, reads, (np.array(bcs, dtype=np.int32) if barcodes else None)


def run_novel_case(eng, seed, k, barcodes=False, hit_len=31):
    contigs, reads, bcs = make_novel_case(seed, k, barcodes=barcodes)
    o = Oracle(k)
    ix = eng.index(k, consider_barcode=barcodes)
    if barcodes:
        o.lib.t4o_set_consider_barcode(o.h, 1)
    for name, seq, bc, w in contigs:
        a = o.add_novel(name, seq, 1, bc, w)
        b = ix.add_contig(name, seq, bc, w)
        assert a == b
    o.set_hit_len_required(hit_len)
    ix.set_params(hit_len, 10, 0.9).commit()
    b = eng.upload(reads, bcs)
    for sk in (0, 1):
        off, hits = ix.hits(b, 0, sk)
        assert t4check.check_hits(off, hits, reads, o, allow_total_skip=sk, barcodes=bcs) == []
        cnt, ov = ix.overlaps(b, 0, sk, 128)
        assert t4check.check_overlaps(cnt, ov, reads, o, skip_repeats=sk, barcodes=bcs) == []
        assert (cnt > 0).sum() > len(reads) // 3


@pytest.mark.parametrize("k", [9, 11, 17])
def test_novel_sets(emu_engine, k):
    run_novel_case(emu_engine, 100 + k, k)


def test_novel_sets_barcoded(emu_engine):
    run_novel_case(emu_engine, 77, 9, barcodes=True, hit_len=13)
