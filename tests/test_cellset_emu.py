"""Barcode mode: t4_cellset (per-barcode sets, many cells queried per launch; here through the emulator build) against
ONE unmodified reference SeqSet with SetConsiderBarcodeInIndexHash(true) walked cell after cell as main.cpp:1583-1880
does. Contig ids of the reference are global; a cell's ids are local, so return values are compared after subtracting the
number of contig slots the earlier cells created."""
import filecmp
import os
import random

import pytest

import t4check
from t4libs import REF_FA, Oracle, Ref, RefSeqSet, Synth, rows_to_strs

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref/libt4ref.so not built")


def make_cells(seed, n_cells, pairs_per_cell):
    """reads of every cell in processing order: 2 clones per cell, duplicates adjacent, a few N reads"""
    rnd = random.Random(seed)
    cells = []
    for c in range(n_cells):
        reads = rows_to_strs(Synth(2, seed * 1000 + c).next_reads(pairs_per_cell))
        extra = []
        for r in reads[: pairs_per_cell // 4]:
            x = list(r)
            x[rnd.randrange(len(x))] = "N"
            extra.append("".join(x))
        reads += extra + reads[: pairs_per_cell // 2]
        reads.sort(key=lambda s: (-len(s), s))
        cells.append(reads)
    return cells


def annotate(cells, seed):
    o = Oracle(9, REF_FA, 17)
    rnd = random.Random(seed)
    names, thr, cache = [], [], {}
    for reads in cells:
        nm_c, th_c = [], []
        for rd in reads:
            if rd not in cache:
                _, g = o.annotate_read0(rd)
                nm = ""
                for t in range(4):
                    if g[t][0] != -1:
                        nm = o.name(g[t][0])[:4]
                cache[rd] = nm
            nm_c.append(cache[rd])
            th_c.append(rnd.choice([0.9, 0.95]))
        names.append(nm_c)
        thr.append(th_c)
    return names, thr


class CellWalk:
    """main.cpp's per-read policy for one cell, resumable: `pending()` is the read the next AddRead will be offered"""

    def __init__(self, bc, reads, names, thr):
        self.bc, self.reads, self.names, self.thr = bc, reads, names, thr
        self.i, self.prev, self.log, self.n_ok = 0, -1, [], 0

    def done(self):
        return self.i >= len(self.reads)

    def upcoming(self, n):
        out, j = [], self.i
        while j < len(self.reads) and len(out) < n:
            if j == 0 or self.reads[j] != self.reads[j - 1]:
                out.append(self.reads[j])
            j += 1
        return out

    def step(self, asm, base):
        """process reads up to and including the next AddRead"""
        while not self.done():
            i, rd = self.i, self.reads[self.i]
            self.i += 1
            if i > 0 and rd == self.reads[i - 1]:
                ret = asm.repeat_add_read(rd) if self.prev not in (-1, -3) else self.prev
                self.log.append(("rep", ret - base if ret >= 0 else ret))
                self.prev = ret
                if ret >= 0:
                    self.n_ok += 1
                continue
            ret, strand = asm.add_read(rd, self.names[i], 0, self.bc, 1 + (i % 3), 0, self.thr[i])
            self.log.append(("add", ret - base if ret >= 0 else ret, strand))
            if ret < 0 and i % 4 != 3:
                ret = asm.input_novel_read(self.names[i] if self.names[i] else "Novel", rd, 1 if i % 5 else -1, self.bc)
                self.log.append(("new", ret - base))
            self.prev = ret
            if ret >= 0:
                self.n_ok += 1
            return


def run_reference(cells, names, thr, k, release):
    ref = RefSeqSet(k, 13, consider_barcode=True)
    logs = []
    for bc, reads in enumerate(cells):
        base = ref.size()
        w = CellWalk(bc, reads, names[bc], thr[bc])
        while not w.done():
            w.step(ref, base)
        if bc in release:
            ref.release_finished_barcode(bc, len(reads))
        logs.append(w.log)
    ref.update_all_consensus()
    return ref, logs


def run_cellset(eng, cells, names, thr, k, release, lanes, window):
    import trust4_amd
    cs = trust4_amd.CellSet(eng, k, 13)
    walks = [CellWalk(bc, reads, names[bc], thr[bc]) for bc, reads in enumerate(cells)]
    todo = list(range(len(cells)))
    active = []
    while todo or active:
        while todo and len(active) < lanes:
            active.append(todo.pop(0))
        bcs, rds = [], []
        for bc in active:
            for rd in walks[bc].upcoming(window):
                bcs.append(bc)
                rds.append(rd)
        cs.prefetch(bcs, rds, [0] * len(rds))
        for bc in list(active):
            w = walks[bc]
            for _ in range(window):
                if not w.done():
                    w.step(cs.cell(bc), 0)
            if w.done():
                if bc in release:
                    cs.cell(bc).release_finished_barcode(bc)
                cs.close_cell(bc)
                active.remove(bc)
    cs.update_all_consensus()
    return cs, [w.log for w in walks]


@pytest.fixture(scope="module")
def emu_engine():
    os.environ["T4_LIB"] = t4check.build_emulator_lib()
    import trust4_amd
    eng = trust4_amd.Engine(0)
    yield eng
    os.environ.pop("T4_LIB", None)


@pytest.mark.parametrize("lanes,window", [(1, 1), (4, 1), (3, 4)])
def test_cells_match_reference(emu_engine, tmp_path, lanes, window):
    k = 9
    cells = make_cells(7, 6, 24)
    names, thr = annotate(cells, 7)
    release = {1, 4}
    ref, log_ref = run_reference(cells, names, thr, k, release)
    cs, log_mine = run_cellset(emu_engine, cells, names, thr, k, release, lanes, window)
    for bc in range(len(cells)):
        first = next((i for i, (a, b) in enumerate(zip(log_ref[bc], log_mine[bc])) if a != b), None)
        assert first is None and len(log_ref[bc]) == len(log_mine[bc]), (bc, first, log_ref[bc][first or 0], log_mine[bc][first or 0])
    assert ref.size() == cs.size()
    bnames = ["BC%03d" % i for i in range(len(cells))]
    pa, pb = str(tmp_path / "ref.out"), str(tmp_path / "mine.out")
    ref.output_barcodes(pa, bnames)
    cs.output(pb, bnames)
    assert filecmp.cmp(pa, pb, shallow=False)
    assert sum(1 for l in log_ref for x in l if x[0] == "add" and x[1] >= 0) > 30
    c = cs.counters()
    if lanes > 1:
        assert c["query_batches"] < sum(1 for l in log_mine for x in l if x[0] == "add")
