"""Pins the C oracle (oracle/t4_oracle.c) against the UNMODIFIED reference compiled into
oracle/_ref/libt4ref.so. Skipped when that build is absent (it needs /root/reference)."""
import random

import numpy as np
import pytest

from t4libs import REF_FA, Oracle, Ref, Synth, rows_to_strs

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref/libt4ref.so not built")


@pytest.fixture(scope="module")
def sets():
    return Oracle(9, REF_FA, 17), Ref(9, REF_FA, 17)


def _reads(seed, n_pairs=150):
    rnd = random.Random(seed)
    reads = rows_to_strs(Synth(300, seed).next_reads(n_pairs))

    def mut(x):
        x = list(x)
        for _ in range(rnd.randint(1, 12)):
            x[rnd.randrange(len(x))] = "N"
        return "".join(x)

    reads += [mut(x) for x in reads[:60]]
    reads += ["".join(rnd.choice("ACGT") for _ in range(150)) for _ in range(30)]
    reads += [x[: rnd.randint(5, 149)] for x in reads[:40]]
    reads += ["A" * 150, "ACGT" * 30, "N" * 40, "ACGTACGTA", "ACGTACGT", "TG" * 60 + "N" * 9 + "CA" * 30]
    # long N gaps split the read into contigs (GetContigIntervals)
    reads += [x[:70] + "N" * rnd.randint(5, 12) + x[80:] for x in reads[:40]]
    return reads


def test_ref_set_identical(sets):
    o, r = sets
    assert o.size() == r.size() == 615
    for i in range(o.size()):
        assert o.name(i) == r.name(i)
        assert o.consensus(i) == r.consensus(i)


@pytest.mark.parametrize("seed", [1, 2])
def test_query_path(sets, seed):
    o, r = sets
    for rd in _reads(seed):
        for sk in (0, 1):
            ho, hr = o.hits(rd, allow_total_skip=sk), r.hits(rd, allow_total_skip=sk)
            assert ho.shape == hr.shape and (ho == hr).all()
        for filt in (0, 1):
            assert o.overlaps_from_hits(rd, filt=filt) == r.overlaps_from_hits(rd, filt=filt)
        assert o.overlaps_from_read(rd) == r.overlaps_from_read(rd)
        assert o.overlaps_from_read(rd, skip_repeats=1) == r.overlaps_from_read(rd, skip_repeats=1)
        a, b = o.annotate_read0(rd), r.annotate_read0(rd)
        assert a[0] == b[0]
        for x, y in zip(a[1], b[1]):
            assert x[0] == y[0]
            if x[0] != -1:
                assert x == y


def test_global_alignment():
    o, r = Oracle(9), Ref(9)
    rnd = random.Random(3)
    for it in range(4000):
        lt, lp = rnd.randint(0, 40), rnd.randint(0, 40)
        if it % 50 == 0:
            lt, lp = rnd.randint(0, 300), rnd.randint(0, 12)
        if it % 50 == 1:
            lt, lp = rnd.randint(0, 12), rnd.randint(0, 300)
        t = "".join(rnd.choice("ACGTN" if it % 7 == 0 else "ACGT") for _ in range(lt))
        if rnd.random() < 0.7 and lt > 0:  # related sequences
            p = list(t)
            for _ in range(rnd.randint(0, 4)):
                if p and rnd.random() < 0.5:
                    del p[rnd.randrange(len(p))]
                else:
                    p.insert(rnd.randint(0, len(p)), rnd.choice("ACGT"))
            for _ in range(rnd.randint(0, 3)):
                if p:
                    p[rnd.randrange(len(p))] = rnd.choice("ACGT")
            p = "".join(p)
        else:
            p = "".join(rnd.choice("ACGT") for _ in range(lp))
        assert o.global_alignment(t, p) == r.global_alignment(t, p), (t, p)


def test_global_alignment_posweight():
    o, r = Oracle(9), Ref(9)
    rnd = random.Random(4)
    nprnd = np.random.RandomState(4)
    for it in range(4000):
        lt = rnd.randint(0, 40)
        if it % 50 == 0:
            lt = rnd.randint(100, 200)
        base = [rnd.randrange(4) for _ in range(lt)]
        w = np.zeros((lt, 4), dtype=np.int32)
        for i, b in enumerate(base):
            mode = rnd.random()
            if mode < 0.6:
                w[i, b] = rnd.randint(1, 30)
            elif mode < 0.8:
                w[i] = nprnd.randint(0, 10, 4)
            elif mode < 0.9:
                w[i, b] = 3
                w[i, (b + 1) % 4] = 2
        p = ["ACGT"[b] for b in base]
        for _ in range(rnd.randint(0, 4)):
            c = rnd.random()
            if p and c < 0.3:
                del p[rnd.randrange(len(p))]
            elif c < 0.6:
                p.insert(rnd.randint(0, len(p)), rnd.choice("ACGT"))
            elif p:
                p[rnd.randrange(len(p))] = rnd.choice("ACGTN")
        p = "".join(p)
        assert o.global_alignment_posweight(w, p) == r.global_alignment_posweight(w, p), (w.tolist(), p)


def test_is_mate_overlap():
    o, r = Oracle(9), Ref(9)
    rnd = random.Random(6)
    for it in range(1500):
        L = rnd.randint(40, 150)
        frag = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(L, 2 * L + 20)))
        fr, sr = frag[:L], frag[len(frag) - L:]
        if it % 5 == 0:
            sr = "".join(rnd.choice("ACGT") for _ in range(L))
        if it % 7 == 0:
            fr = ("ACG" * 60)[:L]
            sr = ("ACG" * 60)[1:L + 1]
        sr = list(sr)
        for _ in range(rnd.randint(0, 3)):
            sr[rnd.randrange(L)] = rnd.choice("ACGT")
        sr = "".join(sr)
        for mo, ct in ((min(31, (2 * L) // 10), 0), (min(31, (2 * L) // 20), 1)):
            assert o.is_mate_overlap(fr, sr, mo, ct) == r.is_mate_overlap(fr, sr, mo, ct)


def test_process_read_vs_the_reference_main():
    """The oracle's restatement of ProcessRead (main.cpp:224-449) against the function itself (the reference's main.cpp compiled as
    a library, oracle/_ref/libt4refmain.so): which records are pushed, read 1 as pushed, its qualities, the weight. With qualities
    on both mates (without them the reference's merge dereferences a null pointer, main.cpp:304-309)."""
    from t4libs import RefMain
    from test_engine_emu import pair_cases
    if not RefMain.available():
        pytest.skip("oracle/_ref/libt4refmain.so not built")
    o, r = Oracle(9), RefMain()
    R1, Q1, R2, Q2 = pair_cases(11, 1500)
    kinds = [0, 0, 0, 0]
    for i in range(len(R1)):
        kind, rd, ql, fl = o.process_read(R1[i], Q1[i], R2[i], Q2[i])
        cnt, rrd, rql, rfl = r.process_read(R1[i], Q1[i], R2[i], Q2[i])
        kinds[kind] += 1
        assert (fl & 3) == (rfl & 3), (i, R1[i], R2[i], fl, rfl)
        assert cnt == (1 if fl & 1 else 0) * (2 if fl & 4 else 1) + (1 if fl & 2 else 0), (i, cnt, fl)
        if fl & 1:
            assert (rd, ql, fl & 12) == (rrd, rql, rfl & 12), (i, kind, R1[i], R2[i], rd, rrd)
    assert min(kinds) > 20, kinds
    rnd = random.Random(2)
    # IsLowComplexity through a pair that cannot overlap: read 2 random, read 1 the string under test
    for _ in range(300):
        s = "".join(rnd.choice(rnd.choice(["ACGT", "AAAC", "ACGTN", "AC"])) for _ in range(rnd.randint(30, 120)))
        other = "".join(rnd.choice("ACGT") for _ in range(100))
        kind, rd, ql, fl = o.process_read(s, None, other, None)
        if kind == 0:
            assert (fl & 1) == (0 if r.is_low_complexity(s) else 1), s


def test_lis():
    o, r = Oracle(9), Ref(9)
    rnd = random.Random(7)
    for it in range(3000):
        n = rnd.randint(1, 60)
        diag = rnd.randint(-50, 50)
        pairs = []
        for _ in range(n):
            b = rnd.randint(0, 200)
            a = b + diag + rnd.choice([0, 0, 0, 1, -1, 2, -3, 5, rnd.randint(-10, 10)])
            pairs.append((a, b))
        pairs.sort(key=lambda x: (x[1], x[0]))
        assert o.lis(pairs) == r.lis(pairs), pairs


def _novel_sets(seed, k=9, n_contigs=40):
    """Contigs cut from synthetic transcripts with random per-base weights, loaded identically in both."""
    o, r = Oracle(k), Ref(k)
    rnd = random.Random(seed)
    nprnd = np.random.RandomState(seed)
    frags = rows_to_strs(Synth(60, seed).next_reads(n_contigs))
    contigs = []
    for i, f in enumerate(frags[:n_contigs]):
        L = len(f)
        w = np.zeros((L, 4), dtype=np.int32)
        for j, c in enumerate(f):
            b = "ACGT".index(c)
            w[j, b] = rnd.randint(1, 20)
            if rnd.random() < 0.1:
                w[j, (b + 1) % 4] = rnd.randint(0, 12)
        a = o.add_novel("c%d" % i, f, 1, -1, w)
        b = r.add_novel("c%d" % i, f, 1, -1, w)
        assert a == b
        contigs.append(f)
    return o, r, contigs


@pytest.mark.parametrize("k", [9, 11, 17])
def test_novel_set_queries(k):
    o, r, contigs = _novel_sets(11 + k, k)
    for s in (o, r):
        s.set_hit_len_required(31)
    rnd = random.Random(k)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    reads = []
    for c in contigs:
        for _ in range(4):
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
    n_ext = 0
    for rd in reads:
        for sk in (0, 1):
            ho, hr = o.hits(rd, allow_total_skip=sk), r.hits(rd, allow_total_skip=sk)
            assert ho.shape == hr.shape and (ho == hr).all()
        assert o.overlaps_from_hits(rd, hit_len_required=31) == r.overlaps_from_hits(rd, hit_len_required=31)
        a, b = o.overlaps_from_read(rd), r.overlaps_from_read(rd)
        assert a == b
        assert o.overlaps_from_read(rd, skip_repeats=1) == r.overlaps_from_read(rd, skip_repeats=1)
        assert o.assign_read(rd) == r.assign_read(rd)
        rc = "".join(comp[x] for x in reversed(rd))
        for ov in a[1]:
            rr = rd if ov[5] == 1 else rc
            for f in (1.0, 2.0):
                assert o.extend_overlap(rr, f, ov) == r.extend_overlap(rr, f, ov)
                n_ext += 1
    assert n_ext > 50


def has_hit_reads(seed):
    """stage-0 style input: receptor reads, mutated ones, short ones, random genomic-like reads, chimeras of two genes on
    opposite strands (the ambiguous-strand branch of HasHitInSet)"""
    rnd = random.Random(seed)
    reads = _reads(seed, 120)
    rc = lambda s: s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))
    base = rows_to_strs(Synth(300, seed + 7).next_reads(60))
    reads += [base[2 * i][:75] + rc(base[2 * i + 1])[:75] for i in range(30)]          # plus-strand half + minus-strand half
    reads += ["".join(rnd.choice("ACGT") for _ in range(rnd.randint(20, 150))) for _ in range(80)]
    reads += [x[rnd.randint(0, 100):][: rnd.randint(9, 60)] for x in base]
    return reads


@pytest.mark.parametrize("hit_len", [17, 27, 31])
def test_has_hit_in_set(hit_len):
    o, r = Oracle(9, REF_FA, hit_len), Ref(9, REF_FA, hit_len)
    reads = has_hit_reads(5)
    seen = {0: 0, 1: 0, -1: 0}
    for mode in (0, 1):
        for rd in reads:
            a, b = o.has_hit_in_set(rd, mode), r.has_hit_in_set(rd, mode)
            assert a == b, (mode, rd, a, b)
            seen[a] += 1
    assert seen[1] > 50 and seen[-1] > 50 and seen[0] > 50
