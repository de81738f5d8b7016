"""The ordered contig builder asks for LEAN ExtendOverlap records (t4_add_query_pool*, extendOverlaps(lean)): for an overlap whose
extension meets an indel, only what SeqSet::AddRead reads of it is derived -- the return value 0 and the coordinates (anchor +-
"good" overhangs, SeqSet.hpp:1224-1266); no traceback is walked for it (the path state rides on the DP, eight alignments per
wavefront). This test holds the lean records against the exact ones (t4_add_query, itself held against the oracle and the compiled
reference elsewhere) on a set of contigs as an assembly leaves them: every return value and coordinate, and every field of the
records that extend. It is what caught the reference traceback's border quirk (the step into the origin from (1, 0) / (0, 1) is
recorded as a match), which an 1 200-pair end-to-end run never met."""
import ctypes as C
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

import t4check
import t4libs
from t4libs import REF_FA, ROOT

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "trust4")


def assembled_contigs(tmp_path, pairs, clones, seed):
    """contigs (name, consensus, weights) of the reference's raw assembly of a synthetic batch + the batch's distinct reads"""
    t4libs.build_checkers()
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = str(tmp_path / "s")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), str(clones), str(seed), pre], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([REF_BIN, "-t", "2", "--skipMateExtension", "-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "-o", str(tmp_path / "ref")], check=True, stderr=subprocess.DEVNULL)
    lines = open(str(tmp_path / "ref_raw.out")).read().split("\n")
    contigs, i = [], 0
    while i + 5 < len(lines):
        if lines[i].startswith(">"):
            w = np.array([[int(x) for x in lines[i + 2 + c].split()] for c in range(4)], dtype=np.int32).T.copy()
            contigs.append((lines[i].split()[1], lines[i + 1], w))
            i += 6
        else:
            i += 1
    reads = []
    for f in (pre + "_1.fq", pre + "_2.fq"):
        ls = open(f).read().split("\n")
        reads += [ls[j] for j in range(1, len(ls), 4) if ls[j]]
    return contigs, sorted(set(reads))


def check_lean_records(eng, tmp_path, pairs=1500, clones=30, seed=31, max_reads=900):
    from trust4_amd.api import OV_DTYPE
    contigs, reads = assembled_contigs(tmp_path, pairs, clones, seed)
    ix = eng.index(9)
    for nm, cons, w in contigs:
        ix.add_contig(nm, cons, -1, w)
    ix.set_params(31, 10, 0.9).commit()
    lib, P = eng.lib, C.c_void_p
    reads = reads[:max_reads]
    compared = gapped = 0
    for b0 in range(0, len(reads), 64):
        rs = reads[b0:b0 + 64]
        n = len(rs)
        bases = "".join(rs).encode()
        offs = np.zeros(n + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(r) for r in rs])
        st, fac, M = np.zeros(n, dtype=np.int32), np.ones(n, dtype=np.float64), 512
        cnt = np.zeros(n, dtype=np.int32)
        ov, ex, ret = np.zeros((n, M), dtype=OV_DTYPE), np.zeros((n, M), dtype=OV_DTYPE), np.zeros((n, M), dtype=np.int32)
        assert lib.t4_add_query(ix.h, n, bases, offs.ctypes.data_as(P), None, st.ctypes.data_as(P), 0, fac.ctypes.data_as(P), M, cnt.ctypes.data_as(P),
                                ov.ctypes.data_as(P), ex.ctypes.data_as(P), ret.ctypes.data_as(P)) == 0
        pc, pb, po, pe, pr = P(), P(), P(), P(), P()
        assert lib.t4_add_query_pool(ix.h, n, bases, offs.ctypes.data_as(P), None, st.ctypes.data_as(P), 0, fac.ctypes.data_as(P), C.byref(pc), C.byref(pb), C.byref(po),
                                     C.byref(pe), C.byref(pr), None) == 0
        lc = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_int32)), (n,))
        lb = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_int32)), (n,))
        for i in range(n):
            assert lc[i] == cnt[i]
            k = max(int(cnt[i]), 0)
            if not k:
                continue
            lex = np.frombuffer((C.c_char * (40 * k)).from_address(pe.value + 40 * int(lb[i])), dtype=OV_DTYPE, count=k)
            lret = np.ctypeslib.as_array(C.cast(pr, C.POINTER(C.c_int32)), (int(lb[i]) + k,))[int(lb[i]):]
            for t in range(k):
                a, b = ex[i, t], lex[t]
                compared += 1
                gapped += int(ret[i, t] == 0)
                assert int(ret[i, t]) == int(lret[t]), (b0 + i, t)
                for fld in ("seqIdx", "readStart", "readEnd", "seqStart", "seqEnd", "strand"):
                    assert a[fld] == b[fld], (b0 + i, t, fld, tuple(a.tolist()), tuple(b.tolist()))
                if ret[i, t] == 1:
                    assert tuple(a.tolist()) == tuple(b.tolist()), (b0 + i, t)
    assert compared > 1000 and gapped > 100, (compared, gapped)


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_lean_records_equal_exact_ones_emulated(tmp_path):
    os.environ["T4_LIB"] = t4check.build_emulator_lib()
    try:
        import trust4_amd
        check_lean_records(trust4_amd.Engine(0), tmp_path)
    finally:
        os.environ.pop("T4_LIB", None)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
def test_lean_records_equal_exact_ones_gpu(tmp_path):
    os.environ.pop("T4_LIB", None)
    import trust4_amd
    check_lean_records(trust4_amd.Engine(0), tmp_path, pairs=12000, clones=240, seed=32, max_reads=6000)
