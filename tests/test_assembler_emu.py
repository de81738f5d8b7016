"""The ordered contig builder (trust4_amd/csrc/t4_assembler.cpp + GPU queries, here through the emulator build)
against the UNMODIFIED reference SeqSet driven with the same call sequence (oracle/_ref). The driver below
follows the policy of main.cpp:1583-1880: AddRead every distinct read in order, RepeatAddRead duplicates,
seed a new contig with InputNovelRead when AddRead fails, UpdateAllConsensus periodically."""
import filecmp
import os
import random

import pytest

import t4check
from t4libs import REF_FA, Oracle, Ref, RefSeqSet, Synth, rows_to_strs

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref/libt4ref.so not built")


def make_reads(seed, n_pairs, n_clones):
    rnd = random.Random(seed)
    reads = rows_to_strs(Synth(n_clones, seed).next_reads(n_pairs))
    extra = []
    for r in reads[: n_pairs // 3]:
        x = list(r)
        x[rnd.randrange(len(x))] = "N"
        extra.append("".join(x))
    reads += extra + reads[: n_pairs // 2]          # N-containing reads and exact duplicates
    reads.sort(key=lambda s: (-len(s), s))          # duplicates become adjacent, as after main.cpp's sort
    return reads


def drive(asm, reads, names, thresholds, update_every=150, window=0, grow_at=None):
    """grow_at = (i, k): SeqSet::ChangeKmerLength(k) before read i (main.cpp:1874-1879 does it when the set passes 4096 * 4^n
    contigs: the set is compacted, contig ids are renumbered, the whole index is rebuilt with the new k)"""
    log = []
    prev_ret, n_ok = -1, 0
    for i, rd in enumerate(reads):
        if grow_at and i == grow_at[0]:
            asm.change_kmer_length(grow_at[1])   # Clean() forgets prevAddInfo: a RepeatAddRead right after returns -1 (SeqSet.hpp:4619)
        if window and not (i > 0 and rd == reads[i - 1]) and not asm.window_valid():
            nxt = [reads[j] for j in range(i, len(reads)) if j == 0 or reads[j] != reads[j - 1]][:window]
            asm.prefetch(nxt, [0] * len(nxt))
        if i > 0 and rd == reads[i - 1]:
            ret = asm.repeat_add_read(rd) if prev_ret not in (-1, -3) else prev_ret
            log.append(("rep", ret))
        else:
            ret, strand = asm.add_read(rd, names[i], 0, -1, 1 + (i % 7), 0, thresholds[i])
            log.append(("add", ret, strand))
            if ret < 0 and i % 3 != 2:
                ret = asm.input_novel_read(names[i] if names[i] else "Novel", rd, 1 if i % 5 else -1, -1)
                log.append(("new", ret))
        prev_ret = ret
        if ret >= 0:
            n_ok += 1
            if n_ok % update_every == 0:
                asm.update_all_consensus()
    asm.update_all_consensus()
    return log


def run_case(eng, tmp_path, seed, n_pairs, n_clones, k=9, window=0, grow_at=None):
    import trust4_amd
    reads = make_reads(seed, n_pairs, n_clones)
    # gene names come from the rough annotation, as in main.cpp:1609-1620 (first 4 letters of the last annotated gene)
    o = Oracle(9, REF_FA, 17)
    names, thr = [], []
    rnd = random.Random(seed)
    for rd in reads:
        _, g = o.annotate_read0(rd)
        nm = ""
        for t in range(4):
            if g[t][0] != -1:
                nm = o.name(g[t][0])[:4]
        names.append(nm)
        thr.append(rnd.choice([0.9, 0.95, 0.97]))
    ref = RefSeqSet(k)
    mine = trust4_amd.Assembler(eng, k)
    log_ref = drive(ref, reads, names, thr, grow_at=grow_at)
    log_mine = drive(mine, reads, names, thr, window=window, grow_at=grow_at)
    if window:
        c = mine.counters()
        assert c["window_hits"] > 0 and c["queries"] < sum(1 for x in log_mine if x[0] == "add")
    first_diff = next((i for i, (a, b) in enumerate(zip(log_ref, log_mine)) if a != b), None)
    assert first_diff is None, (first_diff, log_ref[first_diff], log_mine[first_diff])
    pa, pb = str(tmp_path / "ref_raw.out"), str(tmp_path / "mine_raw.out")
    ref.output(pa)
    mine.output(pb)
    assert filecmp.cmp(pa, pb, shallow=False)
    n_add = sum(1 for x in log_ref if x[0] == "add" and x[1] >= 0)
    assert n_add > len(reads) // 10
    return log_ref


@pytest.fixture(scope="module")
def emu_engine():
    os.environ["T4_LIB"] = t4check.build_emulator_lib()
    import trust4_amd
    eng = trust4_amd.Engine(0)
    yield eng
    os.environ.pop("T4_LIB", None)


@pytest.mark.parametrize("seed", [1, 2])
def test_add_path_matches_reference(emu_engine, tmp_path, seed):
    run_case(emu_engine, tmp_path, seed, 150, 12)


def test_speculation_window_is_exact(emu_engine, tmp_path):
    run_case(emu_engine, tmp_path, 5, 200, 10, window=32)


def test_kmer_length_growth_in_lock_step(emu_engine, tmp_path):
    """ChangeKmerLength in the middle of an assembly (k 9 -> 11), with a speculation window standing: the live set is compacted,
    renumbered, re-indexed and its device image sent anew; everything after must still equal the reference call for call"""
    run_case(emu_engine, tmp_path, 7, 160, 10, window=24, grow_at=(170, 11))


def _shared_segment_case(n_contigs, seed):
    """contigs that all carry one 130 bp segment (with up to two substitutions each) between random flanks, and reads out of
    that segment: every such read meets every contig -- the shape of a constant-gene read in a set of thousands of contigs"""
    rnd = random.Random(seed)
    rand = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    core = rand(130)
    contigs = []
    for c in range(n_contigs):
        x = list(core)
        for _ in range(rnd.choice([0, 0, 1, 2])):
            p = rnd.randrange(len(x))
            x[p] = rnd.choice([b for b in "ACGT" if b != x[p]])
        contigs.append(rand(60) + "".join(x) + rand(60))
    reads = []
    for r in range(14):
        c = contigs[rnd.randrange(n_contigs)]
        a = rnd.choice([40, 55, 60, 70, 90])
        rd = list(c[a:a + 140])
        for _ in range(rnd.choice([0, 0, 1, 3])):
            p = rnd.randrange(len(rd))
            rd[p] = rnd.choice("ACGT")
        reads.append("".join(rd))
    reads += reads[:3]
    return contigs, reads


def _run_shared_segment(eng, tmp_path, n_contigs, window):
    import trust4_amd
    contigs, reads = _shared_segment_case(n_contigs, 11)

    def play(asm, win):
        log = []
        for c, s in enumerate(contigs):
            log.append(("new", asm.input_novel_read("C%d" % c, s, 1, -1)))
        for i, rd in enumerate(reads):
            if win and not asm.window_valid():
                asm.prefetch(reads[i:i + win], [0] * len(reads[i:i + win]))
            ret, strand = asm.add_read(rd, "", 0, -1, 1, 0, 0.9)
            log.append(("add", ret, strand))
            if ret < 0 and i % 2 == 0:
                log.append(("new", asm.input_novel_read("N%d" % i, rd, 1, -1)))
            if i % 5 == 4:
                asm.update_all_consensus()
        asm.update_all_consensus()
        return log
    ref, mine = RefSeqSet(9), trust4_amd.Assembler(eng, 9)
    log_ref, log_mine = play(ref, 0), play(mine, window)
    first_diff = next((i for i, (a, b) in enumerate(zip(log_ref, log_mine)) if a != b), None)
    assert first_diff is None, (first_diff, log_ref[first_diff], log_mine[first_diff])
    pa, pb = str(tmp_path / "ref_raw.out"), str(tmp_path / "mine_raw.out")
    ref.output(pa)
    mine.output(pb)
    assert filecmp.cmp(pa, pb, shallow=False)
    assert sum(1 for x in log_ref if x[0] == "add" and x[1] >= 0) >= 4
    return log_ref


def test_read_meeting_hundreds_of_contigs(emu_engine, tmp_path):
    """600 overlaps per read: the read outgrows the LDS arrays inside its launch (tens of thousands of hits), its overlaps are
    ordered by the key sort (more than 256 of them), the pre-filters against the best novel overlap are replayed in chunks (more
    than 50 novel overlaps) and their ExtendOverlaps run in the extension kernel (more than 64)"""
    _run_shared_segment(emu_engine, tmp_path, 600, 4)


def test_read_meeting_hundreds_of_contigs_small_blocks(tmp_path):
    """the same with the testing aid that shrinks the LDS staging block to 64 keys (T4_AQ_CAP_LIMIT is read once per process:
    own process): hit sort and overlap key sort both go through the blocked sort in global scratch"""
    import subprocess
    import sys
    code = ("import os, sys, pathlib\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import t4check\n"
            "os.environ['T4_LIB'] = t4check.build_emulator_lib()\n"
            "import trust4_amd, test_assembler_emu as t\n"
            "t._run_shared_segment(trust4_amd.Engine(0), pathlib.Path(%r), 300, 4)\n"
            "print('identical')\n") % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, T4_AQ_CAP_LIMIT="120"), capture_output=True, text=True)
    assert p.returncode == 0 and "identical" in p.stdout, p.stderr[-1500:]
