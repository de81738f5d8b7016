"""Kernel logic on the CPU: the SAME kernel sources (trust4_amd/csrc) compiled against the fiber
emulator in tests/hipemu, compared with the golden vectors and the oracle. This is test
infrastructure for a container without a GPU -- the `-m gpu` suite runs the same checks on the real
hipcc build."""
import ctypes
import os
import random

import numpy as np
import pytest

import t4check
from t4libs import REF_FA, ROOT, Oracle, Ref, Synth, rows_to_strs


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
    # ExtendOverlap of every returned overlap, and AssignRead
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    cnt, ov = ix.overlaps(b, 0, 0, 32)
    for factor in (1.0, 2.0):
        ret, ext = ix.extend(b, cnt, ov, factor)
        n_ext = 0
        for i, rd in enumerate(reads):
            rc = "".join(comp[x] for x in reversed(rd))
            for t in range(max(int(cnt[i]), 0)):
                o_in = tuple(ov[i, t].tolist())
                eret, eout = o.extend_overlap(rd if o_in[5] == 1 else rc, factor, o_in)
                assert int(ret[i, t]) == eret and tuple(ext[i, t].tolist()) == tuple(eout), (i, t, o_in, eout, tuple(ext[i, t].tolist()))
                n_ext += 1
        assert n_ext > 30
    aret, aout = ix.assign(b, 0)
    n_asg = 0
    for i, rd in enumerate(reads):
        eret, eout = o.assign_read(rd, 0, -1 if bcs is None else int(bcs[i]))
        assert int(aret[i]) == eret, (i, eret, int(aret[i]))
        if eret != -1:
            assert tuple(aout[i].tolist()) == tuple(eout), (i, eout, tuple(aout[i].tolist()))
            n_asg += 1
    assert n_asg > 5
    if not barcodes:
        run_final_tail_case(eng, o, ix, b, reads, contigs, hit_len)


def run_final_tail_case(eng, o, ix, b, reads, contigs, hit_len):
    """The read-level part of the bulk `_final.out` tail (main.cpp:2075-2118): AssignRead of every read with its own strand
    argument at novelSeqSimilarity 0.95, then RecomputePosWeight from the assignments -- engine vs oracle, and the oracle vs the
    compiled reference when it is there."""
    rnd = np.random.RandomState(len(reads))
    strands = rnd.choice([-1, 0, 1], size=len(reads)).astype(np.int32)
    o.set_novel_similarity(0.95)
    ix.set_params(hit_len, 10, 0.95)
    aret, aout = ix.assign_strands(b, strands)
    exp = []
    for i, rd in enumerate(reads):
        eret, eout = o.assign_read(rd, int(strands[i]), -1)
        assert int(aret[i]) == eret, (i, eret, int(aret[i]))
        if eret != -1:
            assert tuple(aout[i].tolist()) == tuple(eout), (i, eout, tuple(aout[i].tolist()))
        exp.append(eout if eret != -1 else (-1, -1, -1, -1, -1, 1, 0, 0, 0.0))
    assert sum(1 for e in exp if e[0] != -1) > 5
    mult = rnd.randint(1, 4, size=len(reads)).astype(np.int32)
    lens = [len(c[1]) for c in contigs]
    asg = np.zeros(len(reads), dtype=aout.dtype)
    for i, e in enumerate(exp):
        asg[i] = e
    got = ix.posweight_recompute(b, asg, sum(lens), mult)
    rep_reads, rep_asg = [], []
    for i, rd in enumerate(reads):
        rep_reads += [rd] * int(mult[i])
        rep_asg += [exp[i]] * int(mult[i])
    o.recompute_posweight(rep_reads, rep_asg)
    at = 0
    for c, ln in enumerate(lens):
        assert (got[at:at + ln] == o.posweight(c)).all(), c
        at += ln
    assert got.sum() > sum(lens)
    # ... followed by UpdateConsensus of every contig (SeqSet.hpp:4537-4588): t4_consensus_recompute; the reads carry substitutions,
    # so a column covered by such a read alone changes its base
    got2, cons2, changed2 = ix.consensus_recompute(b, asg, sum(lens), mult)
    assert (got2 == got).all()
    r = None
    if Ref.available():
        r = Ref(o_k(o))
        for name, seq, bc, w in contigs:
            r.add_novel(name, seq, 1, bc, w)
        r.recompute_posweight(rep_reads, rep_asg)
        for c in range(len(lens)):
            assert (r.posweight(c) == o.posweight(c)).all(), c
    exp_changed = o.update_all_consensus_chars()
    exp_cons = "".join(o.consensus(c) for c in range(len(lens)))
    assert cons2.decode() == exp_cons and changed2 == exp_changed
    if r is not None:
        r.update_all_consensus_chars()
        assert "".join(r.consensus(c) for c in range(len(lens))) == exp_cons
    print("UpdateConsensus after RecomputePosWeight: %d of %d bases changed" % (changed2, sum(lens)))
    assert changed2 > 0
    o.set_novel_similarity(0.9)
    return changed2


def o_k(o):
    o.lib.t4o_kmer_length.restype = ctypes.c_int
    o.lib.t4o_kmer_length.argtypes = [ctypes.c_void_p]
    return o.lib.t4o_kmer_length(o.h)


@pytest.mark.parametrize("k", [9, 11, 17])
def test_novel_sets(emu_engine, k):
    run_novel_case(emu_engine, 100 + k, k)


def test_novel_sets_barcoded(emu_engine):
    run_novel_case(emu_engine, 77, 9, barcodes=True, hit_len=13)


def make_dp_cases(seed, n, posweight):
    rnd = random.Random(seed)
    nprnd = np.random.RandomState(seed)
    T, P = [], []
    for it in range(n):
        lt = rnd.randint(0, 40)
        if it % 25 == 0:
            lt = rnd.randint(60, 300)
        base = [rnd.randrange(4) for _ in range(lt)]
        p = ["ACGT"[b] for b in base]
        nedit = rnd.randint(0, 5) if it % 3 else rnd.randint(3, 14)
        for _ in range(nedit):
            c = rnd.random()
            if p and c < 0.35:
                del p[rnd.randrange(len(p))]
            elif c < 0.7:
                p.insert(rnd.randint(0, len(p)), rnd.choice("ACGT"))
            elif p:
                p[rnd.randrange(len(p))] = rnd.choice("ACGTN")
        if it % 11 == 0:
            p = [rnd.choice("ACGT") for _ in range(rnd.randint(0, 40))]
        p = "".join(p)[:300]
        if posweight:
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
            T.append(w)
        else:
            t = "".join("ACGT"[b] for b in base)
            if it % 7 == 0 and lt:
                t = list(t)
                t[rnd.randrange(lt)] = "N"
                t = "".join(t)
            T.append(t)
        P.append(p)
    return T, P


def check_gap_dp(eng, seed, n):
    """The scoring kernels return what GetOverlapsFromRead reads of an alignment, the GetAlignStats counts (matches, mismatches,
    indels) -- compared here for four formulations of both aligners. The EDIT STRING itself, which only ExtendOverlap reads (of the posWeight aligner, SeqSet.hpp:1203-1235), is
    compared through impl 4 (the scratch-row aligner with its traceback). The affine aligner's string is never read on this path."""
    o = Oracle(9)
    for kind in (0, 1):
        T, P = make_dp_cases(seed + kind, n, kind == 1)
        exp, exp_al = [], []
        for t, p in zip(T, P):
            sc, al = o.global_alignment(t, p) if kind == 0 else o.global_alignment_posweight(t, p)
            exp.append((al.count(0), al.count(1), al.count(2) + al.count(3)))
            exp_al.append(al)
        if kind == 1:
            got4, strings = eng.gap_dp(1, T, P, 4)
            bad = [i for i in range(n) if got4[i, 3] == 0 and (strings[i] != exp_al[i] or tuple(got4[i, :3]) != exp[i])]
            assert not bad, (bad[:5], [(exp_al[i], strings[i]) for i in bad[:2]])
            assert (got4[:, 3] == 0).sum() > n * 9 // 10
        for impl in (0, 1, 2, 3):
            got = eng.gap_dp(kind, T, P, impl)
            wide = got[:, 3] == 2     # impl 2 / 3 only: band wider than one wavefront / one 16-lane row (the scorer falls back)
            assert impl >= 2 or not wide.any()
            assert ((got[:, 3] == 0) | wide).all() and wide.sum() < (n // 10 if impl == 2 else n // 2)
            if impl == 3:
                assert (~wide).sum() > n // 3
            bad = [i for i in range(n) if not wide[i] and tuple(got[i, :3]) != exp[i]]
            assert not bad, (kind, impl, bad[:5], [(T[i] if kind == 0 else T[i].tolist(), P[i], exp[i], got[i].tolist()) for i in bad[:2]])


def test_gap_dp_vs_oracle(emu_engine):
    check_gap_dp(emu_engine, 5, 600)


def equal_length_cases(seed, n_random):
    """Equal-length targets with 0..4 substitutions at every placement: the affine shortcut's domain."""
    import itertools
    rnd = random.Random(seed)
    T, P = [], []
    for L in range(2, 11):
        for mmc in range(0, 5):
            for pos in itertools.islice(itertools.combinations(range(L), mmc), 40):
                for rep in range(3):
                    t = [rnd.choice("ACGT") for _ in range(L)]
                    if rep == 1:
                        t = [rnd.choice("AC") for _ in range(L)]
                    if rep == 2:
                        t = ["A"] * L
                    p = list(t)
                    for q in pos:
                        p[q] = rnd.choice([c for c in "ACGT" if c != t[q]])
                    T.append("".join(t))
                    P.append("".join(p))
    for _ in range(n_random):
        L = rnd.randint(2, 40)
        t = [rnd.choice("ACGT" if rnd.random() < .7 else "AC") for _ in range(L)]
        p = list(t)
        for q in rnd.sample(range(L), min(L, rnd.randint(0, 5))):
            p[q] = rnd.choice("ACGTN")
        if rnd.random() < .2:
            t[rnd.randrange(L)] = "N"
        T.append("".join(t))
        P.append("".join(p))
    return T, P


def check_equal_length_shortcut(eng, n_random):
    o = Oracle(9)
    T, P = equal_length_cases(3, n_random)
    got = eng.gap_dp(0, T, P, 0)
    for i, (t, p) in enumerate(zip(T, P)):
        sc, al = o.global_alignment(t, p)
        assert tuple(got[i, :3]) == (al.count(0), al.count(1), al.count(2) + al.count(3)), (t, p)


def test_affine_equal_length_shortcut(emu_engine):
    check_equal_length_shortcut(emu_engine, 3000)


def mate_cases(seed, n):
    """pairs with a true suffix/prefix overlap (with a few mismatches / N), unrelated pairs, tandem repeats, length extremes"""
    rnd = random.Random(seed)
    F, S, MO = [], [], []
    for it in range(n):
        kind = it % 6
        L = rnd.randint(30, 150)
        base = "".join(rnd.choice("ACGT") for _ in range(2 * L))
        if kind in (0, 1, 2):
            ov = rnd.randint(5, L)
            f = base[:L]
            s = list(base[L - ov: L - ov + rnd.randint(ov, L)])
            for _ in range(rnd.choice([0, 0, 1, 2, 6])):
                if s:
                    s[rnd.randrange(min(len(s), ov))] = rnd.choice("ACGTN")
            s = "".join(s)
        elif kind == 3:
            f, s = base[:L], "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 150)))
        elif kind == 4:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 5)))
            f = base[: L // 2] + unit * 12
            s = unit * rnd.randint(3, 20)
        else:
            f, s = base[: rnd.randint(1, 40)], base[: rnd.randint(1, 40)]
        F.append(f); S.append(s)
        MO.append(min(31, (len(f) + len(s)) // rnd.choice([10, 20])))
    return F, S, MO


def check_mate_overlap(eng, seed, n):
    o = Oracle(9)
    F, S, MO = mate_cases(seed, n)
    for tandem in (0, 1):
        got = eng.mate_overlap(F, S, MO, bool(tandem))
        n_pos = 0
        for i in range(n):
            ret, off, best = o.is_mate_overlap(F[i], S[i], MO[i], tandem)
            assert got[i, 0] == ret, (i, tandem, F[i], S[i], MO[i], got[i].tolist(), (ret, off, best))
            if ret >= 0:
                n_pos += 1
                assert (got[i, 1], got[i, 2]) == (off, best), (i, got[i].tolist(), (ret, off, best))
        assert n_pos > n // 8


def test_mate_overlap_vs_oracle(emu_engine):
    check_mate_overlap(emu_engine, 3, 400)


def pair_cases(seed, n):
    """mate pairs as a sequencer gives them (read 2 from the other strand): fragments shorter than a read (read-through), mates
    that overlap at their ends with few or many disagreements, tandem-repeat overlaps, mates apart, low-complexity reads"""
    rnd = random.Random(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rc = lambda x: "".join(comp[c] for c in reversed(x))
    R1, R2, Q1, Q2 = [], [], [], []
    for _ in range(n):
        L = rnd.choice([60, 100, 150])
        kind = rnd.random()
        if kind < 0.25:
            frag = rnd.randint(25, L - 1)             # fragment shorter than the reads: read-through
        elif kind < 0.65:
            frag = rnd.randint(L + 1, 2 * L - 8)      # ends overlap
        else:
            frag = rnd.randint(2 * L + 1, 3 * L)      # apart
        if rnd.random() < 0.08:
            unit = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(1, 6)))
            t = (unit * 400)[:3 * L + 40]
        elif rnd.random() < 0.08:
            t = "".join(rnd.choice("AAAAAAAC") for _ in range(3 * L + 40))
        else:
            t = "".join(rnd.choice("ACGT") for _ in range(3 * L + 40))
        adapter = "".join(rnd.choice("ACGT") for _ in range(L))
        a = (t[:frag] + adapter)[:L] if frag < L else t[:L]
        b = (rc(t[:frag]) + adapter)[:L] if frag < L else rc(t[frag - L:frag])
        err = rnd.choice([0.0, 0.01, 0.04, 0.12])
        mut = lambda x: "".join((rnd.choice("ACGTN") if rnd.random() < err else c) for c in x)
        a, b = mut(a), mut(b)
        R1.append(a); R2.append(b)
        Q1.append("".join(chr(rnd.choice([35, 40, 50, 60, 70, 73])) for _ in a))
        Q2.append("".join(chr(rnd.choice([35, 45, 55, 66, 73])) for _ in b))
    return R1, Q1, R2, Q2


def check_process_pairs(eng, seed, n):
    """t4_process_pairs (ProcessRead + IsLowComplexity, main.cpp:224-449 / 183-205) against the oracle's restatement, with
    qualities and without; every branch must be taken"""
    o = Oracle(9)
    R1, Q1, R2, Q2 = pair_cases(seed, n)
    for with_q in (True, False):
        got = eng.process_pairs(R1, Q1 if with_q else None, R2, Q2 if with_q else None)
        kinds = [0, 0, 0, 0]
        dropped = 0
        for i in range(n):
            exp = o.process_read(R1[i], Q1[i] if with_q else None, R2[i], Q2[i] if with_q else None)
            assert got[i] == exp, (i, with_q, R1[i], R2[i], got[i], exp)
            kinds[exp[0]] += 1
            dropped += 0 if exp[3] & 1 else 1
        assert min(kinds) > n // 40 and dropped > 0, (kinds, dropped)


def test_process_pairs_vs_oracle(emu_engine):
    check_process_pairs(emu_engine, 5, 500)


def check_process_pairs_edges(eng):
    """empty and very short mates, a mate of the maximum length, one beyond it (refused loudly)"""
    o = Oracle(9)
    rnd = random.Random(9)
    long_a = "".join(rnd.choice("ACGT") for _ in range(384))
    R1 = ["", "A", "ACGTACGTAC", long_a, "ACGTTGCATGCATGCAAGGT"]
    R2 = ["", "T", "", long_a[::-1], ""]
    got = eng.process_pairs(R1, None, R2, None)
    for i in range(len(R1)):
        assert got[i] == o.process_read(R1[i], None, R2[i], None), (i, got[i])
    with pytest.raises(Exception) as e:
        eng.process_pairs([long_a + "A"], None, ["ACGT"], None)
    assert "longer than" in str(e.value)


def test_process_pairs_edges(emu_engine):
    check_process_pairs_edges(emu_engine)


def check_has_hit(eng, seed, hit_lens=(17, 27)):
    from test_oracle_vs_ref import has_hit_reads
    reads = [r for r in has_hit_reads(seed) if set(r) <= set("ACGTN")]
    for hl in hit_lens:
        o = Oracle(9, REF_FA, hl)
        ix = eng.index(9).set_params(hl, 10, 0.9).load_ref_fasta(REF_FA).commit()
        got = ix.has_hit(eng.upload(reads))
        exp = [o.has_hit_in_set(r, 0) for r in reads]
        bad = [i for i in range(len(reads)) if got[i] != exp[i]]
        assert not bad, (hl, len(bad), [(reads[i], int(got[i]), exp[i]) for i in bad[:3]])
        assert sum(1 for e in exp if e == 1) > 20 and sum(1 for e in exp if e == -1) > 20 and sum(1 for e in exp if e == 0) > 20


def test_has_hit_vs_oracle(emu_engine):
    check_has_hit(emu_engine, 5, hit_lens=(17,))


def check_novel_min_statistics(eng, seed=4, n_contigs=180, k=9, repeats=False):
    """More than 100 novel groups of more than 3 hits on a strand: GetOverlapsFromHits raises its minimum run size from the
    group statistics (SeqSet.hpp:784-821), including the `i = j; ++i` stepping that skips the first hit of every other group.
    The set: many near-identical contigs plus one-k-mer decoys interleaved in the id order, so that one-hit groups sit between
    the big ones."""
    rnd = random.Random(seed)
    core = "".join(rnd.choice("ACGT") for _ in range(260))
    if repeats:   # tandem repeats of period 1, 2, 3 and 5: k-mers that equal the one 1 .. 5 positions back, between k-mers of 100+ postings
        rn = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
        core = rn(40) + "AC" * 11 + rn(25) + "A" * 19 + rn(30) + "ACG" * 8 + rn(22) + "GATTC" * 5 + rn(40) + "T" * 12 + rn(30)
    o = Oracle(k)
    ix = eng.index(k)
    for i in range(n_contigs):
        s = list(core)
        for _ in range(rnd.randint(0, 3)):
            s[rnd.randrange(len(s))] = rnd.choice("ACGT")
        s = "".join(s)
        if i % 3 == 1:   # decoy: shares a single k-mer with the core
            st = rnd.randint(0, 200)
            s = "".join(rnd.choice("ACGT") for _ in range(60)) + core[st: st + k] + "".join(rnd.choice("ACGT") for _ in range(60))
        w = np.zeros((len(s), 4), dtype=np.int32)
        for j, ch in enumerate(s):
            w[j, "ACGT".index(ch)] = rnd.randint(1, 9)
        assert o.add_novel("c%d" % i, s, 1, -1, w) == ix.add_contig("c%d" % i, s, -1, w)
    o.set_hit_len_required(17)
    ix.set_params(17, 10, 0.9).commit()
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    reads = []
    comp["N"] = "N"
    for _ in range(40 if repeats else 12):
        st = rnd.randint(0, len(core) - 110)
        rd = core[st: st + rnd.randint(40, 110)]
        if repeats and rnd.random() < 0.3:
            at = rnd.randrange(len(rd))
            rd = rd[:at] + "N" + rd[at + 1:]
        reads.append(rd if rnd.random() < 0.5 else "".join(comp[x] for x in reversed(rd)))
    b = eng.upload(reads)
    room = 128 if n_contigs * 2 // 3 <= 128 else 512   # a read may overlap every copy of the core
    for strand in ((0, 1, -1) if repeats else (0,)):
        cnt, ov = ix.overlaps(b, strand, 0, room)
        assert cnt.max() <= room
        assert t4check.check_overlaps(cnt, ov, reads, o, strand=strand) == []
    if repeats:   # skipRepeats (allowTotalSkip) keeps the one-lane replay of the rule
        cnt1, ov1 = ix.overlaps(b, 0, 1, room)
        assert t4check.check_overlaps(cnt1, ov1, reads, o, skip_repeats=1) == []
    off, hits = ix.hits(b, 0, 0)
    assert t4check.check_hits(off, hits, reads, o) == []
    if not repeats:
        assert (np.diff(off) > 100 * 4 * 2).all() and (cnt > 0).all()   # > 100 groups of > 3 hits per read


def test_novel_min_statistics(emu_engine):
    check_novel_min_statistics(emu_engine)


def test_repeat_skip_rule_on_tandem_repeats(emu_engine):
    """GetHitsFromRead's repeat-skip rule (SeqSet.hpp:1381-1391) where it bites: lists of 100+ postings, k-mers equal to the one
    1 - 5 positions back, N in the reads; the query kernel replays the rule over bit masks (seedPositions), the oracle walks it"""
    check_novel_min_statistics(emu_engine, seed=9, n_contigs=170, repeats=True)


def test_group_stepping_closed_form():
    """The kernel's closed form of the reference's `i = j; ++i` group walk (SeqSet.hpp:784-811): which groups are measured and
    with which size, against a literal simulation of the loop, over random group layouts."""
    def serial(sizes):
        elems = [g for g, n in enumerate(sizes) for _ in range(n)]
        out, i = [], 0
        while i < len(elems):
            j = i + 1
            while j < len(elems) and elems[j] == elems[i]:
                j += 1
            out.append((elems[i], j - i))
            i = j + 1
        return out

    def closed(sizes):
        out = []
        for t, n in enumerate(sizes):
            skip = False
            if t >= 1:
                r = max(x for x in range(t) if x == 0 or sizes[x] >= 2)
                skip = (t - r - 1) % 2 == 0
            if n - skip > 0:
                out.append((t, n - skip))
        return out

    rnd = random.Random(1)
    for _ in range(5000):
        sizes = [rnd.choice([1, 1, 1, 2, 3, 5]) for _ in range(rnd.randint(1, 14))]
        assert serial(sizes) == closed(sizes), sizes


def check_long_reads_and_limits(eng, n=40):
    """reads up to the engine's maximum (T4_MAXL = 384 bp: merged mates reach ~290 bp in stage 1), ragged lengths in one batch,
    and the loud failure one base beyond the limit"""
    import trust4_amd
    rnd = random.Random(12)
    o = Oracle(9, REF_FA, 17)
    ix = eng.index(9).set_params(17, 10, 0.9).load_ref_fasta(REF_FA).commit()
    # long reads = stretches of reference genes (V ... C) glued with random linkers
    names = [o.name(i) for i in range(o.size())]
    seqs = [o.consensus(i) for i in range(o.size())]
    reads = []
    for _ in range(n):
        parts = []
        while sum(len(x) for x in parts) < 384:
            s = seqs[rnd.randrange(len(seqs))]
            st = rnd.randint(0, max(0, len(s) - 60))
            parts.append(s[st: st + rnd.randint(40, 200)])
            parts.append("".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 12))))
        rd = "".join(parts).replace("N", "A")
        reads.append(rd[: rnd.choice([384, 384, 383, 300, 257, 200])])
    reads += ["ACGT", "", "A" * 384, reads[0][:9], reads[1][:8]]
    ann = ix.annotate_rough(eng.upload(reads))
    assert t4check.check_annotate(ann, reads, o) == []
    assert (ann["seqIdx"][:, 0] != -1).sum() + (ann["seqIdx"][:, 2] != -1).sum() + (ann["seqIdx"][:, 3] != -1).sum() > n // 2
    with pytest.raises(trust4_amd.T4Error):
        eng.upload([("ACGT" * 97)[:385]])       # 385 bp
    with pytest.raises(trust4_amd.T4Error):
        eng.upload(["ACGTRYACGT" * 5])          # IUPAC letters are refused, not guessed


def test_long_reads_and_limits(emu_engine):
    check_long_reads_and_limits(emu_engine, n=12)


def test_work_distribution_is_not_observable(emu_engine, ref_index, monkeypatch):
    """Blocks fetch their next read from a per-launch counter (or stride statically under T4_STATIC_STRIDE); the tier lists are
    appended 64 reads at a time in whatever order the blocks get there. Neither may change a result."""
    reads = rows_to_strs(Synth(300, 11).next_reads(150)) + ["A" * 150, "ACGTACGTA", "", "N" * 40]
    b = emu_engine.upload(reads)
    monkeypatch.delenv("T4_STATIC_STRIDE", raising=False)
    a1 = ref_index.annotate_rough(b).copy()
    tiers = emu_engine.stats()["tier_reads"]
    assert sum(tiers) == len(reads) and sum(1 for x in tiers if x) >= 2   # more than one tier list was filled
    monkeypatch.setenv("T4_STATIC_STRIDE", "1")
    a2 = ref_index.annotate_rough(b).copy()
    assert a1.tobytes() == a2.tobytes()
    rev = emu_engine.upload(reads[::-1])                                # another list order, another batch composition
    a3 = ref_index.annotate_rough(rev).copy()
    assert a1.tobytes() == a3[::-1].tobytes()


class LiveImage:
    """A Python-side replica of the layout t4_index_apply_delta patches (the role HostIndex plays in t4_assembler.cpp): table on
    the k-mer code, posting lists with room to grow, one arena of bases. add() follows KmerIndex::BuildIndexFromRead
    (KmerIndex.hpp:118-141) as the oracle's add_novel does; flush() ships what changed since the last flush."""

    def __init__(self, k, table_slots=64):
        self.k, self.slots = k, table_slots
        self.keys = {}            # code -> [slot, start, cnt, cap]
        self.occ = set()
        self.post = []            # flat arena of (idx, offset) / None
        self.seqs = []            # (base_off, cons, pw bytes, name)
        self.bases_used = 0
        self.dirty_keys, self.dirty_post, self.dirty_seq = set(), set(), set()
        self.rebuilt = True

    def _slot(self, code):
        from trust4_amd.api import mix64
        s = mix64(code) & (self.slots - 1)
        while s in self.occ:
            s = (s + 1) & (self.slots - 1)
        self.occ.add(s)
        return s

    def _insert(self, code, idx, off):
        if code not in self.keys:
            if 2 * (len(self.keys) + 1) > self.slots:   # re-hash: every key moves
                self.slots *= 2
                self.occ = set()
                for c in self.keys:
                    self.keys[c][0] = self._slot(c)
                self.rebuilt = True
            self.keys[code] = [self._slot(code), 0, 0, 0]
        e = self.keys[code]
        if e[2] == e[3]:
            ncap = max(2, 2 * e[3])
            start = len(self.post)
            self.post += [None] * ncap
            for t in range(e[2]):
                self.post[start + t] = self.post[e[1] + t]
                self.dirty_post.add(start + t)
            e[1], e[3] = start, ncap
        self.post[e[1] + e[2]] = (idx, off)
        self.dirty_post.add(e[1] + e[2])
        e[2] += 1
        self.dirty_keys.add(code)

    def add(self, name, cons, weights=None):
        idx = len(self.seqs)
        num = {"A": 0, "C": 1, "G": 2, "T": 3}
        code, prev, invalid = 0, 0, -1
        mask = (1 << (2 * self.k)) - 1
        for i, ch in enumerate(cons):
            if invalid != -1:
                invalid += 1
            code = ((code << 2) & mask) | (num.get(ch, 3) & 3)   # nucToNum['N' - 'A'] & 3 is 0 in the reference; N windows are invalid anyway
            if ch == "N":
                invalid = 0
            if invalid >= self.k:
                invalid = -1
            if i >= self.k - 1 and len(cons) >= self.k:
                if invalid == -1 and (i == self.k or code != prev):
                    self._insert(code, idx, i - self.k + 1)
                prev = code
        pw = bytearray()
        for i, ch in enumerate(cons):
            w = [0, 0, 0, 0]
            if weights is not None:
                w = [int(x) for x in weights[i]]
            elif ch != "N":
                w[num[ch]] = 1
            sm = sum(w)
            pw.append((16 if sm == 0 else 0) | sum((1 << x) for x in range(4) if sm < 3 * w[x]))
        pw.append(16)
        self.seqs.append((self.bases_used, cons, bytes(pw), name))
        self.bases_used += len(cons) + 1 + 17   # room nobody uses: arenas need not be dense
        self.dirty_seq.add(idx)
        return idx

    def flush(self, ix):
        keys = self.keys if self.rebuilt else {c: self.keys[c] for c in self.dirty_keys}
        slots = [(e[0], c, e[1], e[2]) for c, e in keys.items()]
        runs, cur = [], None
        for at in sorted(self.dirty_post):
            if cur is not None and at == cur[0] + len(cur[1]):
                cur[1].append(self.post[at])
            else:
                cur = [at, [self.post[at]]]
                runs.append(cur)
        seqs = [(i, self.seqs[i][0], len(self.seqs[i][1]), -1, self.seqs[i][3]) for i in sorted(self.dirty_seq)]
        bases = [(self.seqs[i][0], self.seqs[i][1].encode() + b"\0", self.seqs[i][2]) for i in sorted(self.dirty_seq)]
        ix.apply_delta(self.slots, 1 if self.rebuilt else 0, len(self.post) + 8, self.bases_used + 8, len(self.seqs) + 4, len(self.seqs),
                       max(len(x[1]) for x in self.seqs), slots, [(r[0], r[1]) for r in runs], seqs, bases)
        self.dirty_keys, self.dirty_post, self.dirty_seq, self.rebuilt = set(), set(), set(), False


def check_apply_delta(eng, seed=31, k=9):
    """t4_index_apply_delta through the C ABI alone (no t4_assembler): the image of a contig set is written by position from a
    Python replica, queried (hits, overlaps, AssignRead vs the oracle holding the same contigs), then GROWN by a second delta
    (more contigs: longer lists that move, new keys, a re-hashed table) and queried again."""
    contigs, reads, _ = make_novel_case(seed, k)
    o = Oracle(k)
    img = LiveImage(k)
    ix = eng.index(k).set_params(31, 10, 0.9)
    half = len(contigs) // 2
    for part in (contigs[:half], contigs[half:]):
        for name, seq, bc, w in part:
            assert o.add_novel(name, seq, 1, bc, w) == img.add(name, seq, w)
        o.set_hit_len_required(31)
        img.flush(ix)
        b = eng.upload(reads)
        off, hits = ix.hits(b, 0, 0)
        assert t4check.check_hits(off, hits, reads, o) == []
        cnt, ov = ix.overlaps(b, 0, 0, 128)
        assert t4check.check_overlaps(cnt, ov, reads, o) == []
        aret, aout = ix.assign(b, 0)
        for i, rd in enumerate(reads):
            eret, eout = o.assign_read(rd, 0, -1)
            assert int(aret[i]) == eret and (eret == -1 or tuple(aout[i].tolist()) == tuple(eout)), i
    assert (cnt > 0).sum() > len(reads) // 3


def test_apply_delta_through_the_abi(emu_engine):
    check_apply_delta(emu_engine)


def test_threaded_read_packing(emu_engine):
    """t4_reads_upload packs large batches on several host threads: the same batch packed by one and by five threads gives the
    same hits; a base outside the alphabet is reported for the right read either way"""
    import trust4_amd
    reads = rows_to_strs(Synth(50, 4).next_reads(150)) + ["ACGTNNACGT" * 9, "", "A"]
    ref = emu_engine.index(9).set_params(17, 10, 0.9).load_ref_fasta(REF_FA).commit()
    off1, h1 = ref.hits(emu_engine.upload(reads), 0, 0)
    os.environ.update(T4_PACK_MIN="1", T4_PACK_THREADS="5")
    try:
        off2, h2 = ref.hits(emu_engine.upload(reads), 0, 0)
        assert (off1 == off2).all() and (h1 == h2).all()
        bad = list(reads)
        bad[207] = bad[207][:10] + "R" + bad[207][11:]
        with pytest.raises(trust4_amd.T4Error) as e:
            emu_engine.upload(bad)
        assert "read 207" in str(e.value) and "'R'" in str(e.value)
    finally:
        os.environ.pop("T4_PACK_MIN", None)
        os.environ.pop("T4_PACK_THREADS", None)
