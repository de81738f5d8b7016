"""Stage-0 candidate filter: `fastq-extractor-hip` (HasHitInSet on the GPU) against the reference `fastq-extractor` compiled from
/root/reference (oracle/_ref/fastq-extractor) on the same FASTQ files: the kept reads must be byte-identical."""
import filecmp
import gzip
import os
import random
import shutil
import subprocess

import pytest

from t4libs import REF_FA, ROOT, Synth, rows_to_strs

REF_EXT = os.path.join(ROOT, "oracle", "_ref", "fastq-extractor")


def stage0_input(seed, n_receptor_pairs, n_other_pairs):
    """receptor pairs diluted in random 'genomic' pairs, low-complexity reads, pairs where only mate 2 is a receptor read,
    short reads; shuffled"""
    rnd = random.Random(seed)
    rows = rows_to_strs(Synth(50, seed).next_reads(n_receptor_pairs))
    pairs = [(rows[2 * i], rows[2 * i + 1]) for i in range(n_receptor_pairs)]
    rand = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    for _ in range(n_other_pairs):
        pairs.append((rand(150), rand(150)))
    for i in range(n_receptor_pairs // 4):
        pairs.append((rand(150), rows[2 * i + 1]))              # only the mate hits
        pairs.append((rows[2 * i][:rnd.randint(10, 60)], rand(rnd.randint(20, 150))))
    pairs += [("A" * 150, rows[1]), ("ACAC" * 37, "TG" * 75), (rows[0], "N" * 150), ("N" * 20 + rows[2][20:], rows[3])]
    rnd.shuffle(pairs)
    return pairs


def write_fq(path, seqs):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write("@q%d\n%s\n+\n%s\n" % (i, s, "F" * len(s)))


def run_case(tmp_path, driver, mode, n_receptor=300, n_other=1500, trim=None):
    """trim = (lo, hi): every read cut to a length in [lo, hi] -- short single-end reads, where the reference's
    hitLenRequired is 23 (FastqExtractor.cpp:436-438) instead of the 27 of paired input or the len / 5 of long reads"""
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pairs = stage0_input(11, n_receptor, n_other)
    if trim:
        rnd = random.Random(7)
        cut = lambda s: s[:rnd.randint(trim[0], trim[1])] if len(s) > trim[0] else s
        pairs = [(cut(a), cut(b)) for a, b in pairs]
    f1, f2 = str(tmp_path / "in_1.fq"), str(tmp_path / "in_2.fq")
    write_fq(f1, [p[0] for p in pairs])
    write_fq(f2, [p[1] for p in pairs])
    args = ["-f", fa] + (["-u", f1] if mode == "single" else ["-1", f1, "-2", f2])
    ref_o, my_o = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_EXT, "-t", "1"] + args + ["-o", ref_o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([driver] + args + ["-o", my_o], check=True)
    names = [".fq"] if mode == "single" else ["_1.fq", "_2.fq"]
    for s in names:
        assert filecmp.cmp(ref_o + s, my_o + s, shallow=False), s
    n_kept = open(ref_o + names[0]).read().count("\n") // 4
    assert n_receptor * (2 if trim else 3) // 3 <= n_kept < len(pairs) // 2


def barcode_inputs(tmp_path, seed, n_receptor_pairs, n_other_pairs):
    """10x-style stage-0 input: reads whose headers carry `CB:Z:<barcode> UB:Z:<umi>` comments, a barcode+UMI FASTQ file
    (16 + 10 bases with qualities), a whitelist and a translation table. Barcodes: exact, one substitution (correctable; with
    a tie between two whitelist entries decided by count / quality), two substitutions and an N (not correctable), empty."""
    rnd = random.Random(seed)
    pairs = stage0_input(seed, n_receptor_pairs, n_other_pairs)
    rand = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    cells = [rand(16) for _ in range(12)]
    cells.append(cells[0][:5] + ("A" if cells[0][5] != "A" else "C") + cells[0][6:])   # neighbour of cell 0: ties for a read one off both
    def mutate(b, k):
        b = list(b)
        for pos in rnd.sample(range(16), k):
            b[pos] = rnd.choice([x for x in "ACGT" if x != b[pos]])
        return "".join(b)
    bcs, quals = [], []
    for i in range(len(pairs)):
        b = cells[rnd.randrange(len(cells))] if i % 3 else cells[0]
        r = i % 11
        if r == 1: b = mutate(b, 1)
        elif r == 2: b = mutate(b, 2)
        elif r == 3: b = b[:7] + "N" + b[8:]
        elif r == 4 and i % 2: b = b[:5] + rnd.choice("GT") + b[6:]
        bcs.append(b + rand(10))
        quals.append("".join(rnd.choice("#5AFI") for _ in range(26)))
    f1, f2, fb = str(tmp_path / "in_1.fq"), str(tmp_path / "in_2.fq"), str(tmp_path / "in_bc.fq")
    with open(f1, "w") as g1, open(f2, "w") as g2, open(fb, "w") as gb:
        for i, (a, b) in enumerate(pairs):
            comment = "" if i % 17 == 5 else " CB:Z:%s\tUB:Z:%s" % (bcs[i][:16], bcs[i][16:]) if i % 2 else " UB:Z:%s CB:Z:%s" % (bcs[i][16:], bcs[i][:16])
            g1.write("@q%d%s\n%s\n+\n%s\n" % (i, comment, a, "F" * len(a)))
            g2.write("@q%d\n%s\n+\n%s\n" % (i, b, "F" * len(b)))
            if i % 29 == 7:
                gb.write("@q%d\n\n+\n\n" % i)                       # empty barcode record
            else:
                gb.write("@q%d\n%s\n+\n%s\n" % (i, bcs[i], quals[i]))
    wl, tr = str(tmp_path / "wl.txt"), str(tmp_path / "tr.txt")
    with open(wl, "w") as f:
        f.write("\n".join(cells + [rand(16) for _ in range(30)] + ["ACGTNNNNACGTACGT"]) + "\n")
    with open(tr, "w") as f:
        for j, cbc in enumerate(cells[:-3]):                           # three cells have no translation
            f.write("%s%s%s\n" % ("CELL%02d" % j, ",\t "[j % 3], cbc))
    return f1, f2, fb, wl, tr


BARCODE_CASES = {
    "format_whitelist": lambda fb, f1, wl, tr: ["--barcode", fb, "--UMI", fb, "--readFormat", "bc:0:15,um:16:-1", "--barcodeWhitelist", wl],
    "format_revcomp": lambda fb, f1, wl, tr: ["--barcode", fb, "--UMI", fb, "--readFormat", "bc:0:15:-,um:16:-1,r1:5:120"],
    "header_fields": lambda fb, f1, wl, tr: ["--barcode", f1, "--UMI", f1, "--readFormat", "r1:0:99,r2:10:-1:-,bc:hd:CB:5:-1,um:hd:1:5:-1", "--barcodeWhitelist", wl],
    "translate_skip": lambda fb, f1, wl, tr: ["--barcode", fb, "--readFormat", "bc:0:7,bc:8:15:+,r1:0:-1", "--barcodeWhitelist", wl, "--barcodeTranslate", tr, "--skipBarcodeErrorRead"],
    "plain_barcode": lambda fb, f1, wl, tr: ["--barcode", fb, "--UMI", fb],
}


def run_barcode_case(tmp_path, driver, case, n_receptor_pairs, n_other_pairs):
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    f1, f2, fb, wl, tr = barcode_inputs(tmp_path, 13, n_receptor_pairs, n_other_pairs)
    args = ["-f", fa, "-1", f1, "-2", f2] + BARCODE_CASES[case](fb, f1, wl, tr)
    ref_o, my_o = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_EXT, "-t", "1"] + args + ["-o", ref_o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([driver] + args + ["-o", my_o], check=True)
    names = ["_1.fq", "_2.fq", "_bc.fa"] + (["_umi.fa"] if "--UMI" in args else [])
    for s in names:
        assert filecmp.cmp(ref_o + s, my_o + s, shallow=False), (case, s)
    bc_lines = open(ref_o + "_bc.fa").read().split("\n")[1::2]
    assert len(bc_lines) >= n_receptor_pairs // 2 and max(len(x) for x in bc_lines) >= 6
    if case == "format_revcomp":
        # the --barcodeStart/End/RevComp, --umiStart/End, --read1Start/End spelling of the same request. The reference leaves two fields
        # of such segments uninitialised (ReadFormatter.hpp:241-252): its binary writes empty barcodes / reads for them or segfaults
        # (--read2Start/End), so the range options are held to their --readFormat equivalent instead
        alt = ["--barcode", fb, "--barcodeStart", "0", "--barcodeEnd", "15", "--barcodeRevComp", "--UMI", fb, "--umiStart", "16", "--umiEnd", "-1",
               "--read1Start", "5", "--read1End", "120"]
        subprocess.run([driver, "-f", fa, "-1", f1, "-2", f2] + alt + ["-o", my_o + "_alt"], check=True)
        for s in names:
            assert filecmp.cmp(ref_o + s, my_o + "_alt" + s, shallow=False), (case, "range options", s)
    if "--barcodeWhitelist" in args and "--skipBarcodeErrorRead" not in args:
        assert "missing_barcode" in bc_lines


def _emulated_extractor():
    """the driver source linked against the emulator build of the kernels (test infrastructure)"""
    import t4check
    lib = t4check.build_emulator_lib()
    exe = os.path.join(ROOT, "tests", "hipemu", "fastq-extractor-hip-emu")
    srcs = [os.path.join(ROOT, "trust4_amd", "host", f) for f in ("fastq_extractor_main.cpp", "read_format.h", "seq_reader.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max([os.path.getmtime(x) for x in srcs] + [os.path.getmtime(lib)]):
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, srcs[0], "-L" + os.path.dirname(lib), "-lt4hip_emu",
                        "-Wl,-rpath," + os.path.dirname(lib), "-lz", "-lpthread"], check=True)
    return exe


@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not built")
@pytest.mark.parametrize("case", sorted(BARCODE_CASES))
def test_extractor_barcode_options_emulated(tmp_path, case):
    run_barcode_case(tmp_path, _emulated_extractor(), case, 80, 120)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not shipped")
@pytest.mark.parametrize("case", sorted(BARCODE_CASES))
def test_extractor_barcode_options_match_reference_binary(tmp_path, case):
    import trust4_amd.build as b
    b.build()
    # header_fields stays at the emulated size: with header-comment fields the REFERENCE binary double-frees (glibc abort) on every
    # larger input of this generator that was tried (>= 320 pairs), so there is nothing to compare with beyond that
    n_rec, n_other = (80, 120) if case == "header_fields" else (300, 1500)
    run_barcode_case(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip"), case, n_rec, n_other)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not shipped")
@pytest.mark.parametrize("mode", ["paired", "single"])
def test_extractor_matches_reference_binary(tmp_path, mode):
    import trust4_amd.build as b
    b.build()
    run_case(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip"), mode)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not shipped")
def test_extractor_short_single_end_matches_reference_binary(tmp_path):
    run_case(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip"), "single", trim=(75, 100))


@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not built")
def test_extractor_short_single_end_emulated(tmp_path):
    """75-100 bp single-end reads: hitLenRequired 23 (between the 27 of paired input and what len / 5 gives for long reads)"""
    run_case(tmp_path, _emulated_extractor(), "single", n_receptor=60, n_other=150, trim=(75, 100))


@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not built")
def test_extractor_emulated(tmp_path):
    """the same driver source linked against the emulator build of the kernels (test infrastructure)"""
    exe = _emulated_extractor()
    run_case(tmp_path, exe, "paired")


@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not built")
def test_extractor_iupac_letters_emulated(tmp_path):
    """Reads with IUPAC letters (FASTA input, some aligners' output): the candidate test looks at k-mer codes only, where
    KmerCode::Append (KmerCode.hpp:99-106) takes every letter but N as a valid base (nucToNum & 3 = T), so the filter keeps and
    drops exactly what the reference does (batches uploaded with T4_READS_KMERS_ONLY) and writes the reads as they came."""
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    rnd = random.Random(3)
    pairs = stage0_input(13, 50, 120)
    def spice(s):
        s = list(s)
        for _ in range(rnd.choice([0, 1, 1, 2, 5])):
            if s:
                s[rnd.randrange(len(s))] = rnd.choice("RYKMSWBDHV")
        return "".join(s)
    pairs = [(spice(a), spice(b)) for a, b in pairs]
    f1, f2 = str(tmp_path / "in_1.fq"), str(tmp_path / "in_2.fq")
    write_fq(f1, [p[0] for p in pairs])
    write_fq(f2, [p[1] for p in pairs])
    args = ["-f", fa, "-1", f1, "-2", f2]
    ref_o, my_o = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_EXT, "-t", "1"] + args + ["-o", ref_o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([_emulated_extractor()] + args + ["-o", my_o], check=True)
    for s in ("_1.fq", "_2.fq"):
        assert filecmp.cmp(ref_o + s, my_o + s, shallow=False), s
    assert open(ref_o + "_1.fq").read().count("\n") // 4 >= 30
