"""Whole stage 1 through the `trust4-hip` driver (rough annotation + ordered assembly on the GPU engine) against the
reference `trust4` binary: `_raw.out` and `_assembled_reads.fa` must be byte-identical.
  * config[0] of BASELINE.json: the repo's own example (data/example_{1,2}.fq.gz), golden outputs generated here by
    oracle/_ref/trust4 --skipMateExtension and committed under tests/golden/;
  * synthetic 150 bp pairs, compared with oracle/_ref/trust4 run side by side (skipped when it did not travel)."""
import filecmp
import gzip
import os
import shutil
import subprocess

import pytest

from t4libs import REF_FA, ROOT, Synth, rows_to_strs

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "trust4")


def _gunzip(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        shutil.copyfileobj(f, g)


def _driver():
    import trust4_amd.build as b
    b.build()
    return os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip")


@pytest.mark.gpu
def test_example_matches_golden(tmp_path):
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    out = str(tmp_path / "mine")
    subprocess.run([_driver(), "--skipMateExtension", "-f", fa, "-1", os.path.join(ROOT, "data", "example_1.fq.gz"),
                    "-2", os.path.join(ROOT, "data", "example_2.fq.gz"), "-o", out], check=True)
    assert filecmp.cmp(out + "_raw.out", os.path.join(ROOT, "tests", "golden", "example_raw.out"), shallow=False)
    assert filecmp.cmp(out + "_assembled_reads.fa", os.path.join(ROOT, "tests", "golden", "example_assembled_reads.fa"), shallow=False)
    assert filecmp.cmp(out + "_final.out", out + "_raw.out", shallow=False)


def _write_fastq(path, rows, prefix="r"):
    with open(path, "w") as f:
        for i, s in enumerate(rows):
            f.write("@%s%d\n%s\n+\n%s\n" % (prefix, i, s, "I" * len(s)))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
@pytest.mark.parametrize("pairs,clones,seed", [(1000, 40, 1), (4000, 150, 2)])
def test_synthetic_matches_reference_binary(tmp_path, pairs, clones, seed):
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    r1, r2 = Synth(clones, seed).next_pairs(pairs)
    f1, f2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    _write_fastq(f1, rows_to_strs(r1))
    _write_fastq(f2, rows_to_strs(r2))
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "1", "--skipMateExtension", "-f", fa, "-1", f1, "-2", f2, "-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([_driver(), "--skipMateExtension", "-f", fa, "-1", f1, "-2", f2, "-o", my_out], check=True)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix
    assert open(ref_out + "_raw.out").read().count(">") > 5


def _barcode_case(tmp_path, driver, pairs, cells, seed, env=None):
    """10x-style input (tools/t4synth --cells: barcode + UMI FASTA files parallel to the reads) through `driver` and through
    the reference binary; all three outputs must be byte-identical (with barcodes the reference writes _final.out = raw)."""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    pre = str(tmp_path / "c5")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", str(seed), pre, "--cells", str(cells)], check=True)
    args = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"]
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    e = dict(os.environ)
    e.update(env or {})
    subprocess.run([driver] + args + ["-o", my_out], check=True, env=e)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix
    assert open(ref_out + "_raw.out").read().count(">") >= cells


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
@pytest.mark.parametrize("pairs,cells,seed,lanes,window,threads", [(3000, 60, 4, 4096, 4, 8), (3000, 60, 5, 7, 1, 1)])
def test_barcode_mode_matches_reference_binary(tmp_path, pairs, cells, seed, lanes, window, threads):
    _barcode_case(tmp_path, _driver(), pairs, cells, seed, {"T4_LANES": str(lanes), "T4_WINDOW": str(window), "T4_THREADS": str(threads), "T4_SORT_MIN": "256"})


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_barcode_mode_emulated(tmp_path):
    """the same driver source linked against the emulator build of the kernels (test infrastructure): tiny case for the CPU suite"""
    exe = _emulated_driver()
    _barcode_case(tmp_path, exe, 160, 8, 6, {"T4_LANES": "8", "T4_WINDOW": "3", "T4_THREADS": "2", "T4_SORT_MIN": "64"})   # (the read list sorted on the threads)


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
@pytest.mark.parametrize("groups,threads,cells", [(2, 2, 8), (3, 4, 8), (5, 2, 3)])
def test_barcode_mode_cell_groups_emulated(tmp_path, groups, threads, cells):
    """Cell groups (round 4): the cells in contiguous groups, each with its own t4_ctx / stream / image arena / host thread, so that one
    group's query batch runs while the others commit; contig ids of a group follow the slots of the groups before it
    (t4_cellset_output_at). Two and three groups, and more groups than cells (empty groups): the reference binary's bytes."""
    exe = _emulated_driver()
    _barcode_case(tmp_path, exe, 160, cells, 6, {"T4_LANES": "8", "T4_WINDOW": "3", "T4_THREADS": str(threads), "T4_CELL_GROUPS": str(groups), "HIPEMU_THREADS": "2"})


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
@pytest.mark.parametrize("groups", [1, 4, 7])
def test_barcode_mode_cell_groups_gpu(tmp_path, groups):
    _barcode_case(tmp_path, _driver(), 6000, 120, 7, {"T4_THREADS": "8", "T4_CELL_GROUPS": str(groups)})


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_device_kmer_counts_emulated(tmp_path):
    """T4_GPU_KMERCOUNT=1: the 21-mer counts, the count statistics and the quality trimming come from t4_kmer_count_* instead of
    the host threads; T4_GPU_MATEOVERLAP=1: ProcessRead's two IsMateOverlap tests per pair come from t4_mate_overlap (both opt-in
    this round); the outputs must not move. FASTQ qualities with low tails so that the trim happens."""
    import random
    exe = _emulated_driver()
    _barcode_case(tmp_path, exe, 160, 8, 6, {"T4_LANES": "8", "T4_WINDOW": "3", "T4_THREADS": "2", "T4_GPU_KMERCOUNT": "1", "T4_GPU_MATEOVERLAP": "1"})
    rnd = random.Random(5)
    fa = str(tmp_path / "ref2.fa")
    _gunzip(REF_FA, fa)
    r1, r2 = Synth(12, 8).next_pairs(150)
    files = []
    for name, rows in (("q_1.fq", rows_to_strs(r1)), ("q_2.fq", rows_to_strs(r2))):
        path = str(tmp_path / name)
        with open(path, "w") as f:
            for i, r in enumerate(rows):
                tail = rnd.choice([0, 0, 10, 25, 60])
                q = "".join(rnd.choice("FI:") for _ in range(len(r) - tail)) + "".join(rnd.choice("#+5") for _ in range(tail))
                f.write("@r%d\n%s\n+\n%s\n" % (i, r, q))
        files.append(path)
    outs = {}
    for tag, cmd, env in (("ref", [REF_BIN, "-t", "1"], {}), ("mine", [exe], {"T4_GPU_KMERCOUNT": "1", "T4_GPU_MATEOVERLAP": "1"}), ("host", [exe], {})):
        outs[tag] = str(tmp_path / ("bulk_" + tag))
        subprocess.run(cmd + ["--skipMateExtension", "-f", fa, "-1", files[0], "-2", files[1], "-o", outs[tag]], check=True,
                       stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(outs["ref"] + suffix, outs["mine"] + suffix, shallow=False), suffix
        assert filecmp.cmp(outs["ref"] + suffix, outs["host"] + suffix, shallow=False), suffix
    trimmed = [len(l.strip()) for l in open(outs["ref"] + "_assembled_reads.fa") if not l.startswith(">")]
    assert trimmed and min(trimmed) < 150   # the trim did happen


def _barcode_options_case(tmp_path, driver, extra, pairs, cells):
    """--contigMinCov (thin barcodes and shallow contigs dropped) and --keepNoBarcode (index not keyed by barcode: one set --
    the live set of bulk mode, here with the barcode filter on every hit)"""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    pre = str(tmp_path / "c5")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", "8", pre, "--cells", str(cells)], check=True)
    # some reads lose their barcode, one cell is thin
    lines = open(pre + "_bc.fa").read().split("\n")
    for i in range(1, len(lines), 2):
        if (i // 2) % 17 == 0 and lines[i]:
            lines[i] = "missing_barcode"
    open(pre + "_bc.fa", "w").write("\n".join(lines))
    args = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"] + extra
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([driver, "-t", "4"] + args + ["-o", my_out], check=True)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), (extra, suffix)
    assert open(ref_out + "_raw.out").read().count(">") > cells // 4


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
@pytest.mark.parametrize("extra", [["--contigMinCov", "3"], ["--keepNoBarcode"], ["--keepNoBarcode", "--contigMinCov", "2"]])
def test_barcode_mode_options(tmp_path, extra):
    _barcode_options_case(tmp_path, _driver(), extra, 1500, 40)


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_keep_no_barcode_emulated(tmp_path):
    _barcode_options_case(tmp_path, _emulated_driver(), ["--keepNoBarcode"], 200, 8)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
def test_trim_level_2_matches_reference_binary(tmp_path):
    """--trimLevel 2: reference set with k = 7 and radius 0, V gene assignments used as barcodes"""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    r1, r2 = Synth(60, 3).next_pairs(1500)
    f1, f2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    _write_fastq(f1, rows_to_strs(r1))
    _write_fastq(f2, rows_to_strs(r2))
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    args = ["--skipMateExtension", "--trimLevel", "2", "-f", fa, "-1", f1, "-2", f2]
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([_driver()] + args + ["-o", my_out], check=True)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_trim_level_2_emulated(tmp_path):
    """the same on the emulator build (small): the V / C trimming loops of the driver run on its threads"""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    r1, r2 = Synth(12, 5).next_pairs(220)
    f1, f2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    _write_fastq(f1, rows_to_strs(r1))
    _write_fastq(f2, rows_to_strs(r2))
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    args = ["--skipMateExtension", "--trimLevel", "2", "-f", fa, "-1", f1, "-2", f2]
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([_emulated_driver(), "-t", "4"] + args + ["-o", my_out], check=True, env=dict(os.environ, T4_PACK_MIN="16", T4_SORT_MIN="32"))
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix
    trimmed = [len(l.strip()) for l in open(ref_out + "_assembled_reads.fa") if not l.startswith(">")]
    assert trimmed and min(trimmed) < 150   # reads were trimmed


def _edge_reads(seed):
    """synthetic reads plus the edge cases the reference's own input handling has branches for: reads shorter than the 21-mer
    and than k, N runs that split a read into contigs, all-N and low-complexity reads, a read-through pair, exact duplicates"""
    import random
    rnd = random.Random(seed)
    rows = rows_to_strs(Synth(30, seed).next_reads(300))
    r1, r2 = rows[0::2], rows[1::2]
    def rc(s):
        return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))
    extra1, extra2 = [], []
    base = r1[0]
    extra1 += [base[:20], base[:8], base[:5], "N" * 60, "A" * 80, base[:60] + "N" * 9 + base[69:], base[:40] + "NNN" + base[43:]]
    extra2 += [r2[0][:30], r2[0][:12], "ACGT", "N" * 40, "C" * 70, r2[0], r2[1]]
    frag = r1[3][:100]                                   # read-through: the mates cover the same 100 bp fragment
    extra1.append(frag + "AGATCGGAAGAGC"); extra2.append(rc(frag) + "AGATCGGAAGAGC")
    extra1 += [r1[5]] * 3; extra2 += [r2[5]] * 3         # duplicates -> RepeatAddRead
    return r1 + extra1, r2 + extra2


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
@pytest.mark.parametrize("mode", ["paired", "single", "empty"])
def test_edge_inputs_match_reference_binary(tmp_path, mode):
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    a, b = _edge_reads(21)
    if mode == "empty":
        a, b = [], []
    f1, f2 = str(tmp_path / "e_1.fq"), str(tmp_path / "e_2.fq")
    _write_fastq(f1, a)
    _write_fastq(f2, b)
    args = ["--skipMateExtension", "-f", fa] + (["-u", f1] if mode == "single" else ["-1", f1, "-2", f2])
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([_driver()] + args + ["-o", my_out], check=True)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), (mode, suffix)


def _emulated_driver():
    """the driver source linked against the emulator build of the kernels (test infrastructure, CPU suite only)"""
    import t4check
    lib = t4check.build_emulator_lib()
    exe = os.path.join(ROOT, "tests", "hipemu", "trust4-hip-emu")
    src = os.path.join(ROOT, "trust4_amd", "host", "trust4_main.cpp")
    deps = [src, lib] + [os.path.join(ROOT, "trust4_amd", "host", h) for h in ("seq_reader.h", "process_read.h")] + [os.path.join(ROOT, "include", "trust4_hip.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-DT4_TEST_KNOBS", "-o", exe, src, "-L" + os.path.dirname(lib), "-lt4hip_emu",
                        "-Wl,-rpath," + os.path.dirname(lib), "-lz", "-lpthread"], check=True)
    return exe


def _kmer_count_file(path, reads, k=21):
    """a k-mer counter's dump (">COUNT\\nKMER" records) as `-c` reads it: counts of the k-mers as written, singletons included
    (both programs skip them), one k-mer listed twice (the later record wins)"""
    from collections import Counter
    cnt = Counter()
    for r in reads:
        for i in range(len(r) - k + 1):
            cnt[r[i:i + k]] += 1
    items = sorted(cnt.items())
    with open(path, "w") as f:
        for km, c in items:
            f.write(">%d\n%s\n" % (c, km))
        for km, c in items[:3]:
            f.write(">%d\n%s\n" % (c + 5, km))


def _driver_options_case(tmp_path, driver, pairs, clones, seed, env=None):
    """-c (k-mer counts from a file), --debug-ns (contigs preloaded from a FASTA file) and `-o -PREFIX` (contig files on stdout)"""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    r1, r2 = Synth(clones, seed).next_pairs(pairs)
    s1, s2 = rows_to_strs(r1), rows_to_strs(r2)
    f1, f2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    _write_fastq(f1, s1)
    _write_fastq(f2, s2)
    cfile = str(tmp_path / "counts.fa")
    _kmer_count_file(cfile, s1 + s2)
    ns = str(tmp_path / "ns.fa")
    with open(ns, "w") as f:
        f.write(">IGHV_seed extra\n%s\n>other\n%s\n" % (s1[0][:120], s1[1][10:140]))
    base = ["--skipMateExtension", "-f", fa, "-1", f1, "-2", f2]
    for name, extra in (("c", ["-c", cfile]), ("ns", ["--debug-ns", ns]), ("k", ["-k", "11", "--debug-ns", ns, "-c", cfile])):
        ref_out, my_out = str(tmp_path / ("ref_" + name)), str(tmp_path / ("mine_" + name))
        subprocess.run([REF_BIN, "-t", "1"] + base + extra + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([driver] + base + extra + ["-o", my_out], check=True, env=dict(os.environ, **(env or {})))
        for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
            assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), (name, suffix)
    # stdout mode: both contig files on stdout, the assembled reads still in a file named after the prefix
    outs = []
    for exe, sub in ((REF_BIN, "so_ref"), (driver, "so_mine")):
        d = tmp_path / sub
        d.mkdir()
        p = subprocess.run([exe] + (["-t", "1"] if exe == REF_BIN else []) + base + ["-o", "-x"], check=True, cwd=str(d),
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        outs.append((p.stdout, open(str(d / "-x_assembled_reads.fa"), "rb").read(), sorted(os.listdir(str(d)))))
    assert outs[0] == outs[1]
    assert outs[0][0].count(b">") > 2


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_driver_options_emulated(tmp_path):
    _driver_options_case(tmp_path, _emulated_driver(), 60, 4, 9)


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_driver_options_device_kmer_counts_emulated(tmp_path):
    """-c with the counts on the device (T4_GPU_KMERCOUNT=1: t4_kmer_count_set)"""
    _driver_options_case(tmp_path, _emulated_driver(), 60, 4, 9, {"T4_GPU_KMERCOUNT": "1"})


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
def test_driver_options_match_reference_binary(tmp_path):
    _driver_options_case(tmp_path, _driver(), 1500, 50, 9)


def _bulk_case(tmp_path, driver, pairs, clones, seed, env, threads="4"):
    """bulk paired-end input through `driver` (with the testing aids of `env`) and through the reference binary"""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    pre = str(tmp_path / "b")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), str(clones), str(seed), pre], check=True, stdout=subprocess.DEVNULL)
    args = ["--skipMateExtension", "-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    e = dict(os.environ)
    e.update(env)
    e["T4_TIMING"] = "1"
    p = subprocess.run([driver, "-t", threads] + args + ["-o", my_out], check=True, env=e, stderr=subprocess.PIPE, text=True)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), (env, suffix)
    return p.stderr


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
@pytest.mark.parametrize("env", [{"T4_AQ_CAP_LIMIT": "120"}, {"T4_AQ_CAP_LIMIT": "120", "T4_WIDE_PCAP": "512", "T4_WIDE_PARTS": "2", "T4_WIDE_GROUPS": "16"}, {"T4_AQ_CAP_LIMIT": "120", "T4_WIDE_OFF": "1"},
                                 {"T4_AQ_FORCE_GLOBAL": "1"}, {"T4_QUERY_AHEAD": "3", "T4_WINDOW": "7"}, {"T4_AQ_EXTEND_DEFER": "1"}, {"T4_AQ_EXTEND_DEFER": "0"},
                                 {"T4_AQ_POOL_CAP": "8", "T4_AQ_EXTEND_DEFER": "1"}, {"T4_AQ_POOL_CAP": "5"}, {"T4_SORT_MIN": "64"}, {"T4_GPU_PROCESSREAD": "1"},
                                 {"T4_ANNOT_CHUNK": "7"}])
def test_bulk_live_set_paths_emulated(tmp_path, env):
    """Bulk mode = the live set (device image by t4_index_apply_delta, sliding speculation window). The testing aids send a
    small input down the paths large sets take: reads that outgrow the LDS arrays and are spread over the chip by the wide query
    (T4_AQ_CAP_LIMIT; with T4_WIDE_PCAP / _PARTS / _GROUPS through many partitions and the grow-and-repeat paths of its pools; with
    T4_WIDE_OFF they go on in one workgroup's global scratch inside the launch), the global-scratch tier launched beside the LDS tier
    (T4_AQ_FORCE_GLOBAL), a window that is re-queried
    a few reads at a time (T4_QUERY_AHEAD), a result pool that overflows so that the call is repeated with a larger one
    (T4_AQ_POOL_CAP; with the extensions deferred, extendKernel runs over the pool of the failed attempt first), the rough annotation
    in chunks of seven distinct reads (T4_ANNOT_CHUNK: chunk boundaries between the copies of a read). Outputs must equal the reference binary's byte for byte."""
    log = _bulk_case(tmp_path, _emulated_driver(), 240, 5, 11, env)
    if "T4_GPU_PROCESSREAD" in env:   # ProcessRead of every pair on the device (t4_process_pairs): mates must have been merged there
        import re
        m = re.search(r"ProcessRead on the device: (\d+) pairs stay as they are, (\d+) read-through, (\d+) merged", log)
        assert m and int(m.group(3)) > 20, log[-600:]
    if ("T4_AQ_CAP_LIMIT" in env and "T4_WIDE_OFF" in env) or "T4_AQ_FORCE_GLOBAL" in env:
        import re
        m = re.search(r"global-scratch tier (\d+) launches for (\d+) reads", log)
        assert m and int(m.group(2)) > 0, log[-600:]
    elif "T4_AQ_CAP_LIMIT" in env:
        import re
        m = re.search(r"wide query served (\d+) window entries \((\d+) dependency records", log)
        assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0, log[-600:]
    if "T4_AQ_POOL_CAP" in env:
        import re
        m = re.search(r"result pool grown (\d+) times", log)
        assert m and int(m.group(1)) > 0, log[-600:]


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_extra_mate_records_are_ignored_emulated(tmp_path):
    """a mate file with more records than the first read file: the reference's `while ( reads.Next() )` (main.cpp:787-800) never reads
    the extra ones; neither does the block-wise reader of trust4-hip"""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    pre = str(tmp_path / "b")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, "150", "4", "13", pre], check=True, stdout=subprocess.DEVNULL)
    with open(pre + "_2.fq", "a") as f:
        for i in range(3):
            f.write("@extra%d\n%s\n+\n%s\n" % (i, "ACGT" * 30, "I" * 120))
    args = ["--skipMateExtension", "-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([_emulated_driver(), "-t", "2"] + args + ["-o", my_out], check=True, stderr=subprocess.DEVNULL)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
@pytest.mark.parametrize("env", [{}, {"T4_AQ_CAP_LIMIT": "2000"}, {"T4_AQ_CAP_LIMIT": "2000", "T4_WIDE_PCAP": "512", "T4_WIDE_PARTS": "8"}, {"T4_AQ_CAP_LIMIT": "2000", "T4_WIDE_OFF": "1"}, {"T4_AQ_FORCE_GLOBAL": "1"}, {"T4_AQ_EXTEND_DEFER": "2"},
                                 {"T4_AQ_POOL_CAP": "16", "T4_AQ_EXTEND_DEFER": "2"}, {"T4_GPU_PROCESSREAD": "1", "T4_SORT_MIN": "1024"}])
def test_bulk_live_set_paths_gpu(tmp_path, env):
    _bulk_case(tmp_path, _driver(), 6000, 120, 12, env, threads="8")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_tied_reads_take_the_reference_sort_emulated(tmp_path):
    """The driver sorts the read list on its threads only when no two reads tie under the reference's comparator (then the order is
    THE sorted order); pairs whose mates carry the same id and the same sequence do tie, and the list must then go through
    std::sort as in the reference -- the outputs stay byte-identical either way (T4_SORT_MIN sends this small input down the
    threaded path first)."""
    import random
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    r1, r2 = Synth(6, 31).next_pairs(260)
    a, b = rows_to_strs(r1), rows_to_strs(r2)
    rnd = random.Random(5)
    for at in (3, 40, 41, 120, 259):   # both mates of these pairs: one sequence (no overlap with its own reverse complement)
        a[at] = b[at] = "".join(rnd.choice("ACGT") for _ in range(100))
    f1, f2 = str(tmp_path / "t_1.fq"), str(tmp_path / "t_2.fq")
    _write_fastq(f1, a)
    _write_fastq(f2, b)
    args = ["--skipMateExtension", "-f", fa, "-1", f1, "-2", f2]
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    p = subprocess.run([_emulated_driver(), "-t", "4"] + args + ["-o", my_out], check=True, env=dict(os.environ, T4_SORT_MIN="64", T4_TIMING="1"), stderr=subprocess.PIPE, text=True)
    assert "two reads of the list tie" in p.stderr
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_bulk_paired_needs_skip_mate_extension_emulated(tmp_path):
    """Paired-end input without barcodes and without --skipMateExtension: the reference would run its mate-pair extension and the
    annotator reads _final.out, so the driver refuses (exit 1, before touching the input) instead of writing the raw assembly under
    that name; --allowRawFinal gives that file knowingly. Single-end input needs no flag (main.cpp:2018)."""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    pre = str(tmp_path / "b")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, "60", "3", "5", pre], check=True, stdout=subprocess.DEVNULL)
    exe = _emulated_driver()
    args = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "-o", str(tmp_path / "x")]
    p = subprocess.run([exe] + args, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "--skipMateExtension" in p.stderr and not os.path.exists(str(tmp_path / "x_final.out"))
    subprocess.run([exe, "--allowRawFinal"] + args, check=True, stderr=subprocess.DEVNULL)
    assert filecmp.cmp(str(tmp_path / "x_raw.out"), str(tmp_path / "x_final.out"), shallow=False)
    ref_out, my_out = str(tmp_path / "ref_u"), str(tmp_path / "mine_u")
    subprocess.run([REF_BIN, "-t", "1", "-f", fa, "-u", pre + "_1.fq", "-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([exe, "-f", fa, "-u", pre + "_1.fq", "-o", my_out], check=True, stderr=subprocess.DEVNULL)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix


def _verify_window_case(tmp_path, driver, pairs, clones, seed, knobs, threads="4"):
    """T4_VERIFY_WINDOW=1: every window entry is queried again, alone, at the moment it is served, and must equal its cached
    result -- the rules that keep a cached query alive across commits (t4_assembler::processEvents) tested directly, under
    window shapes drawn at random. The run stops with an error at the first difference."""
    import random
    import re
    rnd = random.Random(knobs)
    env = {"T4_VERIFY_WINDOW": "1", "T4_WINDOW": str(rnd.choice([5, 17, 48, 192, 400])), "T4_QUERY_AHEAD": str(rnd.choice([0, 2, 9, 40, 150])),
           "T4_LIVE_HARVEST_DELAY": str(rnd.choice([0, 2, 7, 25])), "T4_LIVE_MIN_BATCH": str(rnd.choice([1, 2, 4, 9])), "T4_LIVE_LANES": str(rnd.choice([1, 2, 3, 5]))}
    if env["T4_QUERY_AHEAD"] == "0":
        del env["T4_QUERY_AHEAD"]
    log = _bulk_case(tmp_path, driver, pairs, clones, seed, env, threads=threads)
    m = re.search(r"T4_VERIFY_WINDOW: (\d+) served window entries queried again at serve time, all equal to their cached results \((\d+) of them put together", log)
    assert m and int(m.group(1)) > pairs // 4, (env, log[-800:])
    assert int(m.group(2)) > pairs // 40, (env, m.groups())   # restricted re-queries: entries that kept the records of their other contigs were served, and verified as sets
    print(env, re.findall(r"timing: query lanes.*", log), re.findall(r"timing: restricted.*", log))
    return int(m.group(1))


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
@pytest.mark.parametrize("knobs", [1, 2, 3, 4, 5, 6])
def test_window_validity_rules_emulated(tmp_path, knobs):
    _verify_window_case(tmp_path, _emulated_driver(), 300, 6, 20 + knobs, knobs)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
@pytest.mark.parametrize("knobs", [11, 12, 13])
def test_window_validity_rules_gpu(tmp_path, knobs):
    """>= 50 k pairs: the k-growth step (4 096 contigs) is not reached at this size, list sizes cross 100 many times"""
    _verify_window_case(tmp_path, _driver(), 50000, 1000, 30 + knobs, knobs, threads="8")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
@pytest.mark.parametrize("pairs,clones,seed", [(3000, 1500, 3), (30000, 15000, 7)])
def test_window_entries_with_stable_group_statistics_gpu(tmp_path, pairs, clones, seed):
    """Many clones on few genes: reads inside a shared gene meet hundreds of contigs, more than 100 groups of four hits, so
    novelMinHitRequired follows the group statistics (SeqSet.hpp:813-823). The query reports when those statistics cannot move
    under index edits of small groups (T4QueryArgs::statsStable) and the window then keeps the entry across such edits without a
    budget. Round 6: an entry without that certificate is CHECKED after such an edit -- its group sizes are booked exactly (the emit mask
    says at which k-mer positions an edit reaches the read at all), the statistics loop is repeated over them on the host, and the
    entry stands while novelMinHitRequired comes out the same. T4_VERIFY_WINDOW checks every served entry against a fresh query; the
    same input with neither (T4_NO_STABLE_STATS + T4_NO_EXACT_TOLERANCE: the budget rule alone) must lose more entries to the tolerance
    rule, and with the checks alone (T4_NO_STABLE_STATS) the checks must have run."""
    import re
    logs = {}
    for name, extra in (("stable", {}), ("checks", {"T4_NO_STABLE_STATS": "1"}), ("budget", {"T4_NO_STABLE_STATS": "1", "T4_NO_EXACT_TOLERANCE": "1"})):
        d = tmp_path / name
        d.mkdir()
        env = {"T4_VERIFY_WINDOW": "1", "T4_TIMING": "1"}
        env.update(extra)
        logs[name] = _bulk_case(d, _driver(), pairs, clones, seed, env, threads="8")
    pat = r"tolerated index edits (\d+), of which (\d+) met an entry whose group statistics cannot move.*tolerance kills (\d+)"
    st, bu = re.search(pat, logs["stable"]), re.search(pat, logs["budget"])
    assert st and bu, logs["stable"][-600:]
    assert int(st.group(2)) > 0 and int(bu.group(2)) == 0
    assert int(st.group(3)) < int(bu.group(3)), (st.groups(), bu.groups())
    ck = re.search(pat, logs["checks"])
    ran = re.search(r"thresholds checked after an edit of a small group by repeating the statistics loop: (\d+) entries, (\d+) of them fell", logs["checks"])
    assert ck and ran and int(ran.group(1)) > 0 and int(ck.group(3)) < int(bu.group(3)), (ck.groups(), ran.groups() if ran else None, bu.groups())
    assert re.search(r"T4_VERIFY_WINDOW: (\d+) served window entries", logs["stable"])


CAND_PAT = (r"candidate store: (\d+) candidate records kept with whole queries, (\d+) restricted re-queries merged through the replay of the scan "
            r"\((\d+) with more than 50 candidates, (\d+) with more than 100 groups of four hits on a strand, (\d+) cut a candidate of another contig\), "
            r"fell back to the whole query: (\d+) a cut candidate of another contig passes now, (\d+) the group statistics could move the threshold, "
            r"(\d+) an overlap on the other strand, (\d+) other; (\d+) whole queries checked against the host's scan")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
@pytest.mark.parametrize("env", [{}, {"T4_AQ_CAP_LIMIT": "1500", "T4_AQ_CAND_CAP": "64", "T4_MAX_PENDING": "2"}])
def test_candidate_store_emulated(tmp_path, env):
    """The candidate store: a window entry keeps EVERY scored candidate overlap of its whole query (pre-score key, scored
    fields, cut by the pre-filters of SeqSet.hpp:1705-1794 or not) and the group statistics of 784-823; after a restricted re-query of
    one contig the host swaps that contig's candidates, repeats the scan and checks that the statistics still certify the
    novelMinHitRequired the other candidates were made with. Many clones on few genes, so that reads inside a shared gene segment meet
    dozens of contigs. T4_VERIFY_WINDOW: every whole query's cut flags and survivors are compared with the host's replay of the scan,
    and every served entry -- most of them put together from restricted re-queries -- with a fresh whole query. The second case sends
    the heavy reads through the wide query (its merge kernel writes the candidates), starts from a candidate pool of 64 records (the
    grow-and-repeat path) and lets an entry wait for two contigs at most."""
    import re
    e = {"T4_VERIFY_WINDOW": "1"}
    e.update(env)
    log = _bulk_case(tmp_path, _emulated_driver(), 800, 400, 3, e, threads="4")
    m = re.search(CAND_PAT, log)
    assert m, log[-1500:]
    recs, merged, big, stats, recut, fb_uncut, fb_stats, fb_strand, fb_other, checked = (int(x) for x in m.groups())
    assert recs > 5000 and merged > 1000 and checked > 1000, m.groups()
    assert fb_uncut + fb_stats + fb_strand + fb_other < merged // 10, m.groups()
    v = re.search(r"T4_VERIFY_WINDOW: (\d+) served window entries queried again at serve time, all equal to their cached results \((\d+) of them put together", log)
    assert v and int(v.group(2)) > 500, log[-800:]
    if "T4_AQ_CAP_LIMIT" in env:
        w = re.search(r"wide query served (\d+) window entries", log)
        assert w and int(w.group(1)) > 0, log[-800:]
    print(m.groups())


def _shared_constant_gene_fasta(path):
    """a gene set whose five chains hold the SAME genes (the IGK records under the names of all five chains): every clone t4synth draws
    then shares one constant gene, and a read inside it meets hundreds of contigs at a depth the emulator can run"""
    recs, cur = [], None
    with gzip.open(REF_FA, "rt") as f:
        for line in f:
            if line.startswith(">"):
                cur = [line, ""] if line[1:4] == "IGK" else None
                if cur:
                    recs.append(cur)
            elif cur:
                cur[1] += line
    with open(path, "w") as g:
        for chain in ("IGK", "IGH", "IGL", "TRA", "TRB"):
            for h, q in recs:
                g.write(">" + chain + h[4:] + q)


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not built")
def test_candidate_store_group_statistics_emulated(tmp_path):
    """The candidate store where novelMinHitRequired follows the group statistics (SeqSet.hpp:784-823: more than 100 groups of four
    hits on a strand): 700 pairs over 350 clones that all share one constant gene. Restricted re-queries of such entries are merged
    when the bounds of the statistics certify the threshold, when the exact replay of the statistics loop over the entry's dependency
    records gives the same one, and -- round 5 -- when it gives a HIGHER one: the candidates chained from runs shorter than the new
    threshold leave the list (run sizes ride with the candidate records). T4_VERIFY_WINDOW compares every served entry with a fresh
    whole query; the three files are the reference binary's."""
    import re
    fa = str(tmp_path / "shared.fa")
    _shared_constant_gene_fasta(fa)
    pre = str(tmp_path / "b")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, "700", "350", "7", pre], check=True, stdout=subprocess.DEVNULL)
    args = ["--skipMateExtension", "-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    p = subprocess.run([_emulated_driver(), "-t", "8"] + args + ["-o", my_out], check=True, env=dict(os.environ, T4_TIMING="1", T4_VERIFY_WINDOW="1"), stderr=subprocess.PIPE, text=True)
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix
    log = p.stderr
    m = re.search(CAND_PAT, log)
    assert m, log[-1500:]
    recs, merged, big, stats, recut, fb_uncut, fb_stats, fb_strand, fb_other, checked = (int(x) for x in m.groups())
    assert merged > 1500 and big > 300 and stats > 80 and checked > 600, m.groups()
    x = re.search(r"(\d+) thresholds settled by repeating the statistics loop over the entry's groups, (\d+) raised thresholds served", log)
    assert x and int(x.group(1)) >= 5 and int(x.group(2)) >= 2, log[-1500:]
    v = re.search(r"T4_VERIFY_WINDOW: (\d+) served window entries queried again at serve time, all equal to their cached results \((\d+) of them put together", log)
    assert v and int(v.group(2)) > 400, log[-800:]
    # round 6: the dependency records of LDS-tier reads are the host's replay of GetHitsFromRead -- held against the wide query's own
    # records for every read the wide query served --, and an entry whose threshold the query could not certify is checked by repeating
    # the statistics loop over its (exactly booked) group sizes instead of spending a budget
    g = re.search(r"the host's replay of the emitted hits equals the wide query's dependency records for all (\d+) reads", log)
    assert g and int(g.group(1)) > 100, log[-1500:]
    c = re.search(r"thresholds checked after an edit of a small group by repeating the statistics loop: (\d+) entries, (\d+) of them fell", log)
    assert c and int(c.group(1)) >= 1, log[-1500:]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
def test_candidate_store_gpu(tmp_path):
    """the same on the GPU at a depth where the regimes the store exists for are met: entries with more than 50 candidates (pre-filters
    live) and with more than 100 groups of four hits on a strand (novelMinHitRequired follows the group statistics), most heavy reads
    served by the wide query; and the round-4 rule (T4_CANDS_OFF) on the same input for comparison of the whole queries it needs"""
    import re
    logs = {}
    for name, extra in (("store", {}), ("off", {"T4_CANDS_OFF": "1"})):
        d = tmp_path / name
        d.mkdir()
        env = {"T4_VERIFY_WINDOW": "1"}
        env.update(extra)
        logs[name] = _bulk_case(d, _driver(), 30000, 15000, 7, env, threads="8")
    m = re.search(CAND_PAT, logs["store"])
    assert m, logs["store"][-1500:]
    recs, merged, big, stats, recut, fb_uncut, fb_stats, fb_strand, fb_other, checked = (int(x) for x in m.groups())
    assert merged > 10000 and big > 100 and stats > 100 and checked > 10000, m.groups()
    q = {k: int(re.search(r"GPU query rounds \d+ with (\d+) reads", v).group(1)) for k, v in logs.items()}
    r = {k: int(re.search(r"restricted re-queries: \d+ entries kept their other contigs when one contig changed, (\d+) merged", v).group(1)) for k, v in logs.items()}
    assert q["store"] - r["store"] < q["off"] - r["off"], (q, r)   # fewer whole queries
    # few clones at the same depth: hundreds of contigs share a constant-gene segment, a commit that lengthens one of them RAISES the
    # novelMinHitRequired of the reads inside it (longestHits / 4 of a larger group) -- served from the store by dropping the candidates
    # chained from runs shorter than the new threshold (run sizes ride with the candidate records), every entry verified as above
    d = tmp_path / "raise"
    d.mkdir()
    log = _bulk_case(d, _driver(), 30000, 600, 1, {"T4_VERIFY_WINDOW": "1"}, threads="8")
    raised = re.search(r"(\d+) raised thresholds served by dropping the candidates of shorter runs", log)
    assert raised and int(raised.group(1)) >= 5, log[-1500:]
    assert "all equal to their cached results" in log
    print(m.groups(), q, r)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/trust4 not shipped")
def test_rccl_gather_inside_the_engine_gpu(tmp_path):
    """`trust4-hip --cellShard 0/1 --rcclId FILE`: the shard results go through t4_comm (ncclCommInitRank, two ncclAllGather
    per payload) and rank 0's merge in C++ -- with one rank here (a box has one GPU), which still runs every call of the path;
    the files must equal the reference binary's."""
    fa = str(tmp_path / "ref.fa")
    _gunzip(REF_FA, fa)
    pre = str(tmp_path / "c5")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, "3000", "0", "4", pre, "--cells", "40"], check=True, stdout=subprocess.DEVNULL)
    args = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"]
    ref_out, my_out = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BIN, "-t", "2"] + args + ["-o", ref_out], check=True, stderr=subprocess.DEVNULL)
    p = subprocess.run([_driver(), "-t", "4"] + args + ["-o", my_out, "--cellShard", "0/1", "--rcclId", str(tmp_path / "id")], stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "over RCCL" in p.stderr, p.stderr[-800:]
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my_out + suffix, shallow=False), suffix
    # the exchange of the 21-mer tables of a run whose input is dealt out by cells (round 5), through the same communicator: one rank
    # owns every cell, exports its table from the device and all-gathers it (T4_SHARD_INPUT=2 lets a single rank take that path)
    my2 = str(tmp_path / "mine2")
    p = subprocess.run([_driver(), "-t", "4"] + args + ["-o", my2, "--cellShard", "0/1", "--rcclId", str(tmp_path / "id2")], stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, T4_SHARD_INPUT="2"))
    assert p.returncode == 0 and "over RCCL" in p.stderr and "pairs of this rank's table went to the other ranks" in p.stderr, p.stderr[-800:]
    for suffix in ("_raw.out", "_assembled_reads.fa", "_final.out"):
        assert filecmp.cmp(ref_out + suffix, my2 + suffix, shallow=False), suffix
