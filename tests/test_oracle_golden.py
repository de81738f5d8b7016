"""The C oracle against the committed golden vectors (generated from the real reference by
tests/golden/make_golden.py). Runs anywhere (no GPU, no /root/reference)."""
import os

import numpy as np
import pytest

from t4libs import REF_FA, ROOT, Oracle

G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(G, "ref_query_k9.npz"))


@pytest.fixture(scope="module")
def oracle():
    return Oracle(9, REF_FA, 17)


def test_annotate_and_overlaps(golden, oracle):
    reads = [str(x) for x in golden["reads"]]
    ann, ovs, off = golden["annotate"], golden["overlaps"], golden["overlap_off"]
    for i, rd in enumerate(reads):
        _, g = oracle.annotate_read0(rd)
        for t in range(4):
            exp = tuple(ann[i, t].tolist())
            assert g[t][0] == int(exp[0])
            if g[t][0] != -1:
                assert tuple(float(v) for v in g[t]) == exp
        ret, lst = oracle.overlaps_from_read(rd)
        exp = ovs[off[i]:off[i + 1]]
        assert len(lst) == len(exp)
        for a, b in zip(lst, exp.tolist()):
            assert tuple(float(v) for v in a) == tuple(b)


def test_hits(golden, oracle):
    reads = [str(x) for x in golden["reads"]]
    hits, hoff = golden["hits"], golden["hit_off"]
    for i, rd in enumerate(reads):
        if i % 8:
            continue
        h = oracle.hits(rd)
        h = h[np.lexsort((h[:, 1], h[:, 2], h[:, 0], h[:, 3]))] if len(h) else h.reshape(0, 5)
        assert (h == hits[hoff[i]:hoff[i + 1]]).all()


def test_dp_known_answers(oracle):
    d = np.load(os.path.join(G, "ref_dp_affine.npz"))
    for t, p, sc, al in zip(d["t"], d["p"], d["score"], d["align"]):
        s, a = oracle.global_alignment(str(t), str(p))
        assert s == int(sc) and "".join(map(str, a)) == str(al)
