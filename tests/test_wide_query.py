"""The wide AddRead query (trust4_amd/csrc/t4_wide.h): a read whose k-mer hits outgrow one workgroup is spread over the chip --
hits scattered into contig-range partitions, every partition sorted / chained / scored by a workgroup of its own, the steps of
GetOverlapsFromRead that look across contigs (group statistics incl. the `i = j; ++i` stepping, sort, strand of the best overlap,
order-dependent pre-filters, similarity cut) replayed over all partitions' records. Held here against the oracle
(GetOverlapsFromRead + ExtendOverlap of every returned overlap) on contig sets where a read meets hundreds to thousands of contigs;
small capacities (T4_AQ_CAP_LIMIT, T4_WIDE_PCAP, T4_WIDE_PARTS, T4_WIDE_GROUPS) push small inputs through many partitions and
through every grow-and-repeat path. The dependency records the wide query returns (hits and hull per (strand, contig) group) are
checked against the oracle's hit lists."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import t4check
from t4libs import Oracle

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def rc(s):
    return "".join(COMP[x] for x in reversed(s))


def shared_core_set(seed, n_contigs, k, core_len=260, decoys=True, flank=0):
    """many near-identical contigs (a gene segment every clone carries) + one-k-mer decoys between them in the id order"""
    rnd = random.Random(seed)
    core = "".join(rnd.choice("ACGT") for _ in range(core_len))
    contigs = []
    for i in range(n_contigs):
        s = list(core)
        for _ in range(rnd.randint(0, 3)):
            s[rnd.randrange(len(s))] = rnd.choice("ACGT")
        s = "".join(s)
        if flank:
            s = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, flank))) + s + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, flank)))
        if decoys and i % 3 == 1:
            st = rnd.randint(0, core_len - k - 1)
            s = "".join(rnd.choice("ACGT") for _ in range(60)) + core[st: st + k] + "".join(rnd.choice("ACGT") for _ in range(60))
        w = np.zeros((len(s), 4), dtype=np.int32)
        for j, ch in enumerate(s):
            w[j, "ACGT".index(ch)] = rnd.randint(1, 9)
            if rnd.random() < 0.05:
                w[j, rnd.randrange(4)] += rnd.randint(1, 9)
        contigs.append(("c%d" % i, s, w))
    return core, contigs


def reads_of_core(rnd, core, n, lo=60, hi=150, errors=2):
    reads = []
    for _ in range(n):
        st = rnd.randint(0, len(core) - lo)
        rd = list(core[st: st + rnd.randint(lo, hi)])
        for _ in range(rnd.randint(0, errors)):
            rd[rnd.randrange(len(rd))] = rnd.choice("ACGTN")
        if rnd.random() < 0.3:
            rd = rd + [rnd.choice("ACGT") for _ in range(rnd.randint(1, 25))]
        rd = "".join(rd)
        reads.append(rd if rnd.random() < 0.5 else rc(rd))
    return reads


def add_query(eng, ix, reads, strands, factors, room):
    from trust4_amd.api import OV_DTYPE
    n = len(reads)
    P = C.c_void_p
    bases = "".join(reads).encode()
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    st = np.asarray(strands, dtype=np.int32)
    fac = np.asarray(factors, dtype=np.float64)
    cnt = np.zeros(n, dtype=np.int32)
    ov, ex, ret = np.zeros((n, room), dtype=OV_DTYPE), np.zeros((n, room), dtype=OV_DTYPE), np.zeros((n, room), dtype=np.int32)
    rcode = eng.lib.t4_add_query(ix.h, n, bases, offs.ctypes.data_as(P), None, st.ctypes.data_as(P), 0, fac.ctypes.data_as(P), room, cnt.ctypes.data_as(P),
                                 ov.ctypes.data_as(P), ex.ctypes.data_as(P), ret.ctypes.data_as(P))
    eng.check(rcode)
    return cnt, ov, ex, ret


def add_query_pool(eng, ix, reads, strands, factors, hints):
    """t4_add_query_pool (lean records, variable-size results) with per-read hints: a hinted read starts on the second stream
    (wideSeedKernel) and its wide query runs beside the round's query kernel -> (counts, list of (ov, ext, ret) arrays per read)"""
    from trust4_amd.api import OV_DTYPE
    n = len(reads)
    P = C.c_void_p
    bases = "".join(reads).encode()
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    st = np.asarray(strands, dtype=np.int32)
    fac = np.asarray(factors, dtype=np.float64)
    hint = np.asarray(hints, dtype=np.uint8).copy()
    pc, pb, po, pe, pr = P(), P(), P(), P(), P()
    eng.check(eng.lib.t4_add_query_pool(ix.h, n, bases, offs.ctypes.data_as(P), None, st.ctypes.data_as(P), 0, fac.ctypes.data_as(P), C.byref(pc), C.byref(pb), C.byref(po),
                                        C.byref(pe), C.byref(pr), hint.ctypes.data_as(P)))
    cnt = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_int32)), (n,)).copy()
    base = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_int32)), (n,)).copy()
    out = []
    for i in range(n):
        k = max(int(cnt[i]), 0)
        if not k:
            out.append(None)
            continue
        ov = np.frombuffer((C.c_char * (40 * k)).from_address(po.value + 40 * int(base[i])), dtype=OV_DTYPE, count=k).copy()
        ex = np.frombuffer((C.c_char * (40 * k)).from_address(pe.value + 40 * int(base[i])), dtype=OV_DTYPE, count=k).copy()
        ret = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_int32)), (int(base[i]) + k,))[int(base[i]):].copy()
        out.append((ov, ex, ret))
    return cnt, out, hint


class Grp(C.Structure):
    _fields_ = [("key", C.c_uint32), ("cnt", C.c_uint32), ("lo", C.c_int32), ("hi", C.c_int32)]


def groups_of(eng, i):
    g, n, huge, n4 = C.POINTER(Grp)(), C.c_int(0), C.c_int(0), C.c_int(0)
    eng.lib.t4_add_query_groups.restype = C.c_int
    got = eng.lib.t4_add_query_groups(eng.h, i, C.byref(g), C.byref(n), C.byref(huge), C.byref(n4))
    if got != 1:
        return None
    # (bits 24-27 of a record's count, reads with lists beyond 10000 postings: hits of shorter lists capped at 4, first hit of a shorter list)
    return [(g[t].key, g[t].cnt & 0xFFFFFF, g[t].lo, g[t].hi) for t in range(n.value)], huge.value, n4.value, [g[t].cnt >> 24 for t in range(n.value)]


def expected_groups(o, read, strand):
    h = o.hits(read, strand=strand, cap=1 << 22)
    tab = {}
    for idx, off, roff, st, _rep in h.tolist():
        tab.setdefault((idx * 2 + (1 if st == 1 else 0)), {}).setdefault(off - roff, 0)
        tab[idx * 2 + (1 if st == 1 else 0)][off - roff] += 1
    out = []
    for key in sorted(tab, key=lambda x: (x & 1, x >> 1)):
        d = tab[key]
        ats = [at for at, c in d.items() if c >= 3]
        lo, hi = (min(ats), max(ats) + len(read) - 1) if ats else (0x7FFFFFFF, -0x7FFFFFFF)
        out.append((key, sum(d.values()), lo, hi))
    return out


def expected_group_info(o, read, strand):
    """per group, in the records' order: hits of lists of at most 10000 postings (capped at 4) | 8 when its hit lowest on the read is one"""
    h = o.hits(read, strand=strand, cap=1 << 22)
    tab = {}
    for idx, off, roff, st, rep in h.tolist():
        tab.setdefault(idx * 2 + (1 if st == 1 else 0), []).append((roff, rep))
    out = []
    for key in sorted(tab, key=lambda x: (x & 1, x >> 1)):
        hits = tab[key]
        small = sum(1 for _, rep in hits if rep <= 10000)
        first = min(hits)[1] <= 10000
        out.append(min(small, 4) | (8 if first else 0))
    return out


def check_wide(eng, o, ix, reads, strands, factors, room, expect_wide=True, stats_before=None):
    cnt, ov, ex, ret = add_query(eng, ix, reads, strands, factors, room)
    n_wide = 0
    for i, rd in enumerate(reads):
        eret, lst = o.overlaps_from_read(rd, strand=int(strands[i]), skip_repeats=0, cap=room + 8)
        assert eret == cnt[i], (i, eret, int(cnt[i]))
        mine = [tuple(x) for x in ov[i, :max(eret, 0)].tolist()]
        assert mine == [tuple(x) for x in lst], (i, "overlap list")
        rcs = rc(rd)
        for t in range(max(eret, 0)):
            o_in = tuple(ov[i, t].tolist())
            xret, xout = o.extend_overlap(rd if o_in[5] == 1 else rcs, float(factors[i]), o_in)
            assert int(ret[i, t]) == xret and tuple(ex[i, t].tolist()) == tuple(xout), (i, t, o_in, xout, tuple(ex[i, t].tolist()))
        g = groups_of(eng, i)
        if g is not None:
            n_wide += 1
            assert g[0] == expected_groups(o, rd, int(strands[i])), (i, "dependency records")
            if g[1]:   # a read with lists beyond 10000 postings: what the statistics loop reads of every group for removeOnlyRepeats
                assert g[3] == expected_group_info(o, rd, int(strands[i])), (i, "hits of shorter lists per group")
            else:
                assert not any(g[3]), (i, "info bits on a read without long lists")
    if expect_wide:
        assert n_wide > 0
    return cnt, n_wide


def build_set(eng, contigs, k, hit_len):
    o = Oracle(k)
    ix = eng.index(k)
    for name, s, w in contigs:
        assert o.add_novel(name, s, 1, -1, w) == ix.add_contig(name, s, -1, w)
    o.set_hit_len_required(hit_len)
    ix.set_params(hit_len, 10, 0.9).commit()
    return o, ix


def wide_stats(eng):
    out = (C.c_int64 * 4)()
    eng.lib.t4_add_query_wide_stats(eng.h, out)
    return list(out)


def run_wide_cases(monkeypatch, make_engine, n_contigs=240, n_reads=14, pcap=256, cap_limit=400, seed=3):
    """reads of a segment that two thirds of the contigs carry: hundreds of overlaps per read, the group statistics live"""
    monkeypatch.setenv("T4_WIDE_PCAP", str(pcap))
    monkeypatch.setenv("T4_AQ_CAP_LIMIT", str(cap_limit))
    monkeypatch.setenv("T4_WIDE_PARTS", "8")        # the partition pool grows on demand
    monkeypatch.setenv("T4_WIDE_GROUPS", "64")      # and so does the pool of dependency records
    eng = make_engine()
    rnd = random.Random(seed)
    for k, hit_len in ((9, 17), (11, 31)):
        core, contigs = shared_core_set(seed + k, n_contigs, k, flank=40)
        o, ix = build_set(eng, contigs, k, hit_len)
        reads = reads_of_core(rnd, core, n_reads) + [core[:150], rc(core[40:190]), "ACGT" * 20, core[:k + 3]]
        strands = [rnd.choice([0, 0, 1, -1]) for _ in reads]
        factors = [rnd.choice([1.0, 2.0]) for _ in reads]
        cnt, n_wide = check_wide(eng, o, ix, reads, strands, factors, room=n_contigs + 16)
        assert cnt.max() > 50      # the order-dependent pre-filters ran
        assert n_wide >= n_reads // 2
        # the same reads on the single-workgroup global-scratch tier give the same records
        monkeypatch.setenv("T4_WIDE_OFF", "1")
        cnt2, ov2, ex2, ret2 = add_query(eng, ix, reads, strands, factors, n_contigs + 16)
        monkeypatch.delenv("T4_WIDE_OFF")
        cnt1, ov1, ex1, ret1 = add_query(eng, ix, reads, strands, factors, n_contigs + 16)
        assert (cnt1 == cnt2).all() and (ov1 == ov2).all() and (ex1 == ex2).all() and (ret1 == ret2).all()
        # hints: the first call learns which reads are heavy, the second starts those on the second stream -- the same records either way
        c0, r0, h0 = add_query_pool(eng, ix, reads, strands, factors, [0] * len(reads))
        assert h0.sum() >= n_reads // 2          # (the call marks the reads the wide query served)
        before = wide_stats(eng)[0]
        c1, r1, h1 = add_query_pool(eng, ix, reads, strands, factors, h0)
        assert wide_stats(eng)[0] - before == h0.sum() and (h1 == h0).all()
        assert (c0 == c1).all() and (c0 == cnt1).all()
        for a, b in zip(r0, r1):
            assert (a is None) == (b is None)
            if a is not None:
                assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
        for i in range(len(reads)):
            if h0[i]:
                assert groups_of(eng, i)[0] == expected_groups(o, reads[i], int(strands[i])), (i, "dependency records of a hinted read")
    st = wide_stats(eng)
    assert st[0] > 0 and st[1] > st[0] and st[2] > 0, st   # reads, partitions (several per read), calls repeated with larger pools
    return st


def test_wide_query_vs_oracle_emulated(monkeypatch):
    monkeypatch.setenv("T4_LIB", t4check.build_emulator_lib())
    import trust4_amd
    run_wide_cases(monkeypatch, lambda: trust4_amd.Engine(0), n_contigs=150, n_reads=8)


def run_huge_lists(monkeypatch, make_engine, n_contigs=10400, n_reads=5):
    """More than 10000 contigs carry the read's k-mers: posting lists beyond 10000 entries switch on removeOnlyRepeats and the
    run test that indexes hits[] with a group-relative k (SeqSet.hpp:802, 876, 934-940). Part of every read lies outside the shared
    segment, in a stretch a few contigs carry, so that groups with and without `unique` hits exist side by side."""
    monkeypatch.delenv("T4_WIDE_PCAP", raising=False)
    monkeypatch.delenv("T4_AQ_CAP_LIMIT", raising=False)
    eng = make_engine()
    rnd = random.Random(8)
    k, hit_len = 9, 17
    shared = "".join(rnd.choice("ACGT") for _ in range(48))
    private = ["".join(rnd.choice("ACGT") for _ in range(70)) for _ in range(6)]
    contigs = []
    for i in range(n_contigs):
        left = private[i % 6] if i % 1733 < 3 else "".join(rnd.choice("ACGT") for _ in range(rnd.randint(20, 40)))
        s = left + shared + "".join(rnd.choice("ACGT") for _ in range(rnd.randint(10, 30)))
        w = np.zeros((len(s), 4), dtype=np.int32)
        for j, ch in enumerate(s):
            w[j, "ACGT".index(ch)] = rnd.randint(1, 9)
        contigs.append(("c%d" % i, s, w))
    o, ix = build_set(eng, contigs, k, hit_len)
    reads = []
    for t in range(n_reads):
        rd = private[t % 6][rnd.randint(0, 30):] + shared[: rnd.randint(30, 48)]
        reads.append(rd if t % 2 == 0 else rc(rd))
    reads.append(shared)
    strands = [0] * len(reads)
    factors = [1.0] * len(reads)
    cnt, n_wide = check_wide(eng, o, ix, reads, strands, factors, room=n_contigs + 64)
    assert n_wide == len(reads)
    for i in range(len(reads)):
        assert groups_of(eng, i)[1] == 1   # huge: a list beyond 10000 postings
    return cnt


def test_wide_query_lists_beyond_10000_postings_emulated(monkeypatch):
    monkeypatch.setenv("T4_LIB", t4check.build_emulator_lib())
    import trust4_amd
    run_huge_lists(monkeypatch, lambda: trust4_amd.Engine(0))


@pytest.mark.gpu
def test_wide_query_vs_oracle_gpu(monkeypatch):
    monkeypatch.delenv("T4_LIB", raising=False)
    import trust4_amd
    run_wide_cases(monkeypatch, lambda: trust4_amd.Engine(0), n_contigs=700, n_reads=40, pcap=1024, cap_limit=2000)


@pytest.mark.gpu
def test_wide_query_lists_beyond_10000_postings_gpu(monkeypatch):
    monkeypatch.delenv("T4_LIB", raising=False)
    import trust4_amd
    run_huge_lists(monkeypatch, lambda: trust4_amd.Engine(0))


@pytest.mark.gpu
def test_wide_query_beyond_the_old_limits_gpu(monkeypatch):
    """one read with more than 262 144 hits and more than 16 384 overlaps (the single-workgroup tier's limits), against the oracle"""
    monkeypatch.delenv("T4_LIB", raising=False)
    monkeypatch.delenv("T4_WIDE_PCAP", raising=False)
    monkeypatch.delenv("T4_AQ_CAP_LIMIT", raising=False)
    import trust4_amd
    eng = trust4_amd.Engine(0)
    rnd = random.Random(21)
    k, hit_len = 9, 17
    n_contigs = 17000     # every emitted k-mer of the read hits every copy (lists beyond 10000 postings: all of them `repeats`)
    core = "".join(rnd.choice("ACGT") for _ in range(170))
    contigs = []
    for i in range(n_contigs):
        s = core      # identical copies: a copy with a variant of its own that the read shares would hold `unique` hits (lists of a few
        #               postings), switch on removeOnlyRepeats and leave the read with that copy alone (test_wide_query_lists_beyond_10000...)
        w = np.zeros((len(s), 4), dtype=np.int32)
        for j, ch in enumerate(s):
            w[j, "ACGT".index(ch)] = rnd.randint(1, 9)
        contigs.append(("c%d" % i, s, w))
    o, ix = build_set(eng, contigs, k, hit_len)
    rd = list(core[5:155])
    rd[70] = "A" if rd[70] != "A" else "C"      # no perfect match: every copy is scored
    rd2 = "".join(rd)
    reads = [rd2, rc(rd2), core[10:160]]
    for i, r in enumerate(reads):
        h = o.hits(r, strand=0, cap=1 << 23)
        if i < 2:
            assert len(h) > 262144, len(h)
    # both strands (strand argument 0); with it, each copy yields two groups
    cnt, n_wide = check_wide(eng, o, ix, reads, [0, 0, 0], [1.0, 2.0, 1.0], room=2 * n_contigs + 64)
    assert cnt[0] > 16384 and n_wide == 3, (cnt.tolist(), n_wide)
