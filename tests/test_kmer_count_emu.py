"""KmerCount on the engine (t4_kmer_count_*): the emulator build of the kernels against the C oracle, and the oracle
against the compiled reference (KmerCount.hpp through oracle/_ref). The `-m gpu` counterpart is tests/test_zz_kmer_count_gpu.py."""
import math
import os
import random

import pytest

import t4check
from t4libs import KmerCountChecker, Ref, Synth, rows_to_strs


def kmer_count_case(seed, n_reads, n_clones=40):
    """reads with a few N, the reference's corner cases, and qualities with low-quality tails (what the trimming looks at)"""
    rnd = random.Random(seed)
    reads = rows_to_strs(Synth(n_clones, seed).next_reads(n_reads // 2))

    def with_n(r):
        r = list(r)
        if rnd.random() < 0.15:
            for _ in range(rnd.randint(1, 3)):
                r[rnd.randrange(len(r))] = "N"
        return "".join(r)

    reads = [with_n(r) for r in reads]
    reads += ["ACGT" * 5, "A" * 21, "N" * 30, "ACGTACGTACGTACGTACGTA", "ACGTNACGTACGTACGTACGTACGTAACCGGTT", "ACGTACGTACGTACGTACGTAN", "",
              reads[0][:60], reads[1][20:], "ACGGTCA" * 40]
    quals = ["".join(rnd.choice("#+5AFI") if rnd.random() < 0.5 else "I" for _ in r) for r in reads]
    for i in range(0, len(quals), 3):
        h = len(quals[i]) // 2
        quals[i] = quals[i][:h] + "#" * (len(quals[i]) - h)
    return reads, quals


def same(a, b):
    return a == b or (isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b))


def check_engine_kmer_counts(eng, seed, n_reads, k=21):
    reads, quals = kmer_count_case(seed, n_reads)
    oracle = KmerCountChecker(k)
    for r in reads:
        oracle.add(r)
    kc = eng.kmer_counter(k, max_kmers=sum(max(0, len(r) - k + 1) for r in reads) + 8)
    half = len(reads) // 2
    kc.add(eng.upload(reads[:half])).add(eng.upload(reads[half:]))   # counts accumulate over batches
    batch = eng.upload(reads)
    for use_q in (False, True):
        mn, md, av, ln = kc.stats(batch, quals if use_q else None)
        for i, r in enumerate(reads):
            ret, omn, omd, oav, r_after, _ = oracle.stats(r, quals[i] if use_q else None)
            got = (int(mn[i]), int(md[i]), float(av[i]), int(ln[i]))
            assert got[0] == omn and got[1] == omd and same(got[2], float(oav)) and got[3] == len(r_after), (i, use_q, r, got, (omn, omd, oav, len(r_after)))
    return kc, reads


@pytest.fixture(scope="module")
def emu_engine():
    os.environ["T4_LIB"] = t4check.build_emulator_lib()
    import trust4_amd
    eng = trust4_amd.Engine(0)
    yield eng
    os.environ.pop("T4_LIB", None)


def test_kmer_counts_and_stats_emulated(emu_engine):
    kc, reads = check_engine_kmer_counts(emu_engine, 5, 120)
    k = 21
    # distinct canonical k-mers, counted independently in Python
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    seen = set()
    for r in reads:
        for p in range(len(r) - k + 1):
            w = r[p:p + k]
            if "N" in w:
                continue
            rc = "".join(comp[c] for c in reversed(w))
            code = lambda s: int("".join(str("ACGT".index(c)) for c in s), 4)
            seen.add(min(code(w), code(rc)))
    assert kc.distinct() == len(seen)


def test_kmer_count_other_k_emulated(emu_engine):
    check_engine_kmer_counts(emu_engine, 9, 60, k=31)
    check_engine_kmer_counts(emu_engine, 10, 60, k=9)


def check_per_barcode_counts(eng, seed, n_reads, n_barcodes=7):
    """per_barcode: one KmerCount per barcode in one table == a separate oracle counter per barcode (main.cpp:1126-1160)"""
    rnd = random.Random(seed)
    reads, _ = kmer_count_case(seed, n_reads, n_clones=6)   # few clones: the same k-mers in several barcodes
    bcs = [rnd.randrange(-1, n_barcodes) for _ in reads]     # -1: reads without a barcode under --keepNoBarcode
    oracles = {}
    for r, b in zip(reads, bcs):
        oracles.setdefault(b, KmerCountChecker(21)).add(r)
    kc = eng.kmer_counter(21, max_kmers=sum(max(0, len(r) - 20) for r in reads) + 8, per_barcode=True)
    batch = eng.upload(reads, bcs)
    kc.add(batch)
    mn, md, av, ln = kc.stats(batch)
    for i, (r, b) in enumerate(zip(reads, bcs)):
        _, omn, omd, oav, r_after, _ = oracles[b].stats(r, None)
        assert (int(mn[i]), int(md[i])) == (omn, omd) and same(float(av[i]), float(oav)) and int(ln[i]) == len(r), (i, b, r)
    with pytest.raises(Exception):
        kc.add(eng.upload(reads))            # a batch without barcodes is refused
    with pytest.raises(Exception):
        eng.kmer_counter(31, 100, per_barcode=True)


def test_per_barcode_counts_emulated(emu_engine):
    check_per_barcode_counts(emu_engine, 14, 120)


def test_kmer_count_set_emulated(emu_engine):
    """AddCountFromFile semantics: counts keyed by the code as given, the later record of a k-mer stays"""
    import ctypes as C
    import numpy as np
    kc = emu_engine.kmer_counter(21, max_kmers=64)
    code = lambda s: int("".join(str("ACGT".index(c)) for c in s), 4)
    canon, noncanon = "A" * 21, "T" * 20 + "G"            # AAAA..A is canonical (its reverse complement is TTTT..T)
    codes = np.array([code(canon), code(noncanon), code(canon)], dtype=np.uint64)
    vals = np.array([5, 9, 7], dtype=np.int32)
    emu_engine.check(emu_engine.lib.t4_kmer_count_set(kc.h, codes.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), C.c_int64(3)))
    assert kc.distinct() == 2
    mn, md, av, ln = kc.stats(emu_engine.upload([canon, "T" * 21, noncanon, "C" * 20 + "A" ]))
    # A*21 and T*21 both look up the canonical A*21 (7: the later record); T..TG looks up its canonical form C A..A, which was never set
    assert list(mn) == [7, 7, 1, 1]


def test_kmer_count_table_grows_emulated(emu_engine, monkeypatch):
    """The table starts at a fraction of what `max_kmers` stands for and is rehashed on the device into one four times as large as it
    fills (round 5: a table for every k-mer POSITION of a million pairs was 12 GB). T4_KC_SLOTS starts it at 1024 slots: the counts of
    a few hundred reads then go through several rehashes, and must equal the oracle's all the same (global and per-barcode tables)."""
    monkeypatch.setenv("T4_KC_SLOTS", "1024")
    kc, reads = check_engine_kmer_counts(emu_engine, 21, 240)
    assert kc.distinct() > 4096
    check_per_barcode_counts(emu_engine, 22, 160)


def check_kmer_count_merge(eng, seed, n_reads, parts=3, k=21):
    """The table of a read set = the sum of the tables of its parts (what the ranks of a run with its input dealt out by cells do:
    t4_kmer_count_export / _merge): a counter per part; part 0's counter takes the others' pairs (a) whole -- it then IS the table
    of the whole set: same pairs as a counter fed every read -- and (b) only where it holds the k-mer already: the statistics of
    part 0's reads equal the oracle's over the whole set, and the table has not grown."""
    import numpy as np
    reads, quals = kmer_count_case(seed, n_reads)
    cut = [len(reads) * p // parts for p in range(parts + 1)]
    cap = sum(max(0, len(r) - k + 1) for r in reads) + 8
    whole = eng.kmer_counter(k, max_kmers=cap).add(eng.upload(reads))
    part = [eng.kmer_counter(k, max_kmers=cap).add(eng.upload(reads[cut[p]:cut[p + 1]])) for p in range(parts)]
    only = eng.kmer_counter(k, max_kmers=cap).add(eng.upload(reads[cut[0]:cut[1]]))
    d0 = only.distinct()
    for p in range(1, parts):
        codes, vals = part[p].export()
        assert len(codes) == part[p].distinct() and len(set(codes.tolist())) == len(codes)
        part[0].merge(codes, vals)
        only.merge(codes, vals, only_present=True)
    as_dict = lambda kc: dict(zip(*[x.tolist() for x in kc.export()]))
    assert as_dict(part[0]) == as_dict(whole)
    assert only.distinct() == d0
    w = as_dict(whole)
    assert as_dict(only) == {c: w[c] for c in as_dict(only)}
    oracle = KmerCountChecker(k)
    for r in reads:
        oracle.add(r)
    mine = reads[cut[0]:cut[1]]
    mn, md, av, ln = only.stats(eng.upload(mine), quals[cut[0]:cut[1]])
    for i, r in enumerate(mine):
        ret, omn, omd, oav, r_after, _ = oracle.stats(r, quals[cut[0] + i])
        assert (int(mn[i]), int(md[i]), int(ln[i])) == (omn, omd, len(r_after)) and same(float(av[i]), float(oav)), (i, r)
    empty = eng.kmer_counter(k, max_kmers=64)
    assert len(empty.export()[0]) == 0
    empty.merge(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.int32))


def test_kmer_count_export_merge_emulated(emu_engine, monkeypatch):
    check_kmer_count_merge(emu_engine, 31, 180)
    monkeypatch.setenv("T4_KC_SLOTS", "1024")   # the merged table outgrows its first size
    check_kmer_count_merge(emu_engine, 32, 120, parts=2)


def test_kmer_count_table_full_is_loud(emu_engine):
    reads = rows_to_strs(Synth(40, 3).next_reads(40))
    kc = emu_engine.kmer_counter(21, max_kmers=16)   # 1024 slots: far too few
    with pytest.raises(Exception):
        kc.add(emu_engine.upload(reads))


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built")
def test_oracle_kmer_count_vs_reference():
    reads, quals = kmer_count_case(3, 300, n_clones=50)
    o, r = KmerCountChecker(21, False), KmerCountChecker(21, True)
    for x in reads:
        assert o.add(x) == r.add(x)
    for x, q in zip(reads, quals):
        for qq in (None, q):
            a, b = o.stats(x, qq), r.stats(x, qq)
            if "N" in x and qq is not None and a[4] != x:
                continue   # the reference sorts stale entries of its reused buffer here (a read with N AND a trimmed tail)
            assert a[:2] == b[:2] and same(a[2], b[2]) and same(a[3], b[3]) and a[4:] == b[4:], (x, qq, a, b)
