"""bam-extractor-hip (stage 0 for BAM input) against the reference's bam-extractor binary (oracle/_ref, built from BamExtractor.cpp
and the BAM library of its vendored samtools): the repo's example.bam where the reference tree exists, and synthetic BAM files
written here (BGZF + BAM records by hand) that walk every branch -- reads inside V/J/C genes and their mates, unaligned templates,
reads on alternative contigs, one-end-unaligned pairs, secondary records, reverse-strand records, spliced / clipped CIGARs,
low-complexity reads, barcode and UMI tags behind tags of other types; paired and single-end. Outputs byte for byte."""
import filecmp
import gzip
import os
import random
import shutil
import struct
import subprocess
import zlib

import pytest

import t4check
from t4libs import REF_FA, ROOT, Synth, rows_to_strs

REF_BAMX = os.path.join(ROOT, "oracle", "_ref", "bam-extractor")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_BAMX), reason="oracle/_ref/bam-extractor not built")
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def bgzf_write(path, payload):
    with open(path, "wb") as f:
        for off in list(range(0, len(payload), 60000)) + [None]:
            chunk = b"" if off is None else payload[off:off + 60000]   # the empty block at the end is the BGZF EOF marker
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            data = c.compress(chunk) + c.flush()
            bsize = len(data) + 25
            f.write(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize) + data + struct.pack("<II", zlib.crc32(chunk), len(chunk)))


def bam_record(name, flag, tid, pos, cigar, seq, qual, mtid=-1, mpos=-1, tags=b""):
    """seq / qual as stored in the BAM (i.e. on the reference strand for reverse-strand records)"""
    ops = "MIDNSHP=X"
    cig = b"".join(struct.pack("<I", n << 4 | ops.index(o)) for n, o in cigar)
    codes = {"=": 0, "A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
    nib = [codes[c] for c in seq] + [0]
    packed = bytes(nib[i] << 4 | nib[i + 1] for i in range(0, len(seq), 2))
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(name) + 1, 30, 4680, len(cigar), flag, len(seq), mtid, mpos, 0)
    body += name.encode() + b"\0" + cig + packed + bytes(ord(q) - 33 for q in qual) + tags
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, records):
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for n, l in refs:
        head += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    bgzf_write(path, head + b"".join(records))


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def tags_for(rnd, i, with_barcodes):
    if not with_barcodes:
        return b""
    t = b"NHC\x01" + b"XSA+" + b"ASi" + struct.pack("<i", 77) + b"ZBBs" + struct.pack("<i", 2) + struct.pack("<hh", 3, -4)
    if i % 5 != 3:
        t += b"CBZ" + ("".join(rnd.choice("ACGT") for _ in range(16)) + "-1").encode() + b"\0"
    t += b"MDZ20A3\0"
    if i % 7 != 2:
        t += b"UBZ" + "".join(rnd.choice("ACGT") for _ in range(10)).encode() + b"\0"
    return t


def synthetic_bam(path, fa, paired, seed, with_barcodes=False, n=60):
    """A coordinate-sorted BAM over three contigs; gene coordinates are taken from the -f file so that some reads fall inside."""
    rnd = random.Random(seed)
    genes, chroms = [], []
    for line in open(fa):
        if line.startswith(">"):
            f = line[1:].split()
            if f[1] not in chroms:
                chroms.append(f[1])            # every contig the gene file names has to be in the BAM header
            if f[1] in ("chr14", "chr2") and len(genes) < 40:
                genes.append((f[1], int(f[2]), int(f[3])))
    refs = [(c, 250000000) for c in sorted(chroms)] + [("chr14_KI270726v1_random", 43739)]
    tid = {r[0]: i for i, r in enumerate(refs)}
    receptor = rows_to_strs(Synth(30, seed).next_reads(n))
    rand = lambda k: "".join(rnd.choice("ACGT") for _ in range(k))
    qual = lambda k: "".join(rnd.choice("FI:5#") for _ in range(k))
    recs = []   # (tid, pos, bytes)

    def add(name, flag, chrom, pos, cigar, seq, mchrom=None, mpos=-1, i=0):
        t = tid[chrom] if chrom else -1
        stored = revcomp(seq) if flag & 0x10 else seq
        q = qual(len(seq))
        recs.append((t if t >= 0 else 1 << 30, pos, bam_record(name, flag, t, pos, cigar, stored, q[::-1] if flag & 0x10 else q,
                                                               tid[mchrom] if mchrom else -1, mpos, tags_for(rnd, i, with_barcodes))))

    P = 0x1 if paired else 0
    for i in range(n):
        kind = i % 10
        name = "frag%d" % i + ("/1" if (paired and i % 9 == 4) else "")
        name2 = name[:-1] + "2" if name.endswith("/1") else name
        g = genes[i % len(genes)]
        inside = g[1] + rnd.randrange(0, max(1, g[2] - g[1] - 60))
        outside = g[2] + 5000 + 200 * i
        if kind in (0, 1, 2):     # inside a gene (kind 2: spliced + clipped + indel CIGAR, reverse strand)
            s = receptor[i][:100] if kind else rand(100)
            cigar = [(100, "M")] if kind < 2 else [(5, "S"), (40, "M"), (2, "I"), (300, "N"), (3, "D"), (50, "M"), (3, "S")]
            fl = (0x10 if kind == 2 else 0)
            if paired:
                add(name, P | 0x2 | 0x40 | fl | (0 if fl else 0x20), g[0], inside, cigar, s, g[0], inside + 150, i)
                add(name2, P | 0x2 | 0x80 | (0x10 if not fl else 0) | (0x20 if fl else 0), g[0], inside + 150, [(100, "M")], rand(100), g[0], inside, i)
                if kind == 1:
                    add(name, P | 0x100 | 0x40, g[0], inside + 7, [(100, "M")], s, g[0], inside + 150, i)   # a secondary record of mate 1
            else:
                add(name, fl, g[0], inside, cigar, s, None, -1, i)
                if kind == 1:
                    add(name, 0x100, g[0], inside + 9, [(100, "M")], s, None, -1, i)                          # same name again
        elif kind == 3:           # aligned outside every gene
            add(name, P | (0x42 if paired else 0), g[0], outside, [(100, "M")], receptor[i][:100], g[0] if paired else None, outside + 200 if paired else -1, i)
            if paired:
                add(name2, P | 0x82 | 0x10, g[0], outside + 200, [(100, "M")], rand(100), g[0], outside, i)
        elif kind == 4:           # on the alternative contig, receptor sequence -> found by the k-mer test
            add(name, P | (0x42 if paired else 0), "chr14_KI270726v1_random", 100 + 30 * i, [(120, "M")], receptor[i][:120], "chr2" if paired else None, 5000 + i if paired else -1, i)
            if paired:
                add(name2, P | 0x82, "chr2", 5000 + i, [(100, "M")], rand(100), "chr14_KI270726v1_random", 100 + 30 * i, i)
        elif kind == 5:           # on the alternative contig, random sequence
            add(name, P | (0x42 if paired else 0), "chr14_KI270726v1_random", 200 + 30 * i, [(100, "M")], rand(100), "chr2" if paired else None, 9000 + i if paired else -1, i)
            if paired:
                add(name2, P | 0x82, "chr2", 9000 + i, [(100, "M")], rand(100), "chr14_KI270726v1_random", 200 + 30 * i, i)
        elif kind == 6 and paired:   # one end unaligned, placed at its mate (flag 0x4, tid set): skipped by the first scan
            add(name, P | 0x40 | 0x8, "chr2", 20000 + i, [(100, "M")], rand(100), "chr2", 20000 + i, i)
            add(name2, P | 0x80 | 0x4, "chr2", 20000 + i, [], receptor[i][:100], "chr2", 20000 + i, i)
        elif kind in (7, 8):      # unaligned template (single-end: unaligned read); 8: one mate of low complexity / random
            a = receptor[i][:130]
            b = rand(110) if kind == 7 else ("A" * 70 + rand(40))
            if i % 4 == 0:
                a, b = rand(120), rand(110)
            if paired:
                first_is_mate1 = i % 3 != 0
                add(name, P | 0x4 | 0x8 | (0x40 if first_is_mate1 else 0x80), None, -1, [], a, None, -1, i)
                add(name2, P | 0x4 | 0x8 | (0x80 if first_is_mate1 else 0x40), None, -1, [], b, None, -1, i)
            else:
                add(name, 0x4, None, -1, [], a, None, -1, i)
        else:                     # receptor read with an N and lower-case-free junk quality, unaligned
            s = receptor[i][:90]
            s = s[:40] + "N" + s[41:]
            if paired:
                add(name, P | 0x4 | 0x8 | 0x40, None, -1, [], s, None, -1, i)
                add(name2, P | 0x4 | 0x8 | 0x80, None, -1, [], rand(90), None, -1, i)
            else:
                add(name, 0x4, None, -1, [], s, None, -1, i)
    order = sorted(range(len(recs)), key=lambda k: (recs[k][0], recs[k][1], k))   # stable: mates of unaligned templates stay adjacent
    write_bam(path, refs, [recs[k][2] for k in order])


def emulated_bam_extractor():
    lib = t4check.build_emulator_lib()
    exe = os.path.join(ROOT, "tests", "hipemu", "bam-extractor-hip-emu")
    srcs = [os.path.join(ROOT, "trust4_amd", "host", f) for f in ("bam_extractor_main.cpp", "bam_reader.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max([os.path.getmtime(x) for x in srcs] + [os.path.getmtime(lib)]):
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, srcs[0], "-L" + os.path.dirname(lib), "-lt4hip_emu",
                        "-Wl,-rpath," + os.path.dirname(lib), "-lz", "-lpthread"], check=True)
    return exe


def compare(tmp_path, driver, bam, fa, extra, paired, barcodes):
    ref_o, my_o = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([REF_BAMX, "-t", "1", "-b", bam, "-f", fa, "-o", ref_o] + extra, check=True, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
    subprocess.run([driver, "-b", bam, "-f", fa, "-o", my_o] + extra, check=True, stderr=subprocess.DEVNULL)
    names = (["_1.fq", "_2.fq"] if paired else [".fq"]) + (["_bc.fa", "_umi.fa"] if barcodes else [])
    for s in names:
        assert filecmp.cmp(ref_o + s, my_o + s, shallow=False), s
    return sum(1 for _ in open(ref_o + names[0])) // 4


def synthetic_cases(tmp_path, driver, n):
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    for tag, paired, barcodes in (("pe", True, False), ("pe_bc", True, True), ("se", False, False), ("se_bc", False, True)):
        d = tmp_path / tag
        d.mkdir()
        bam = str(d / "in.bam")
        synthetic_bam(bam, fa, paired, 7 + len(tag), barcodes, n)
        kept = compare(d, driver, bam, fa, (["--barcode", "CB", "--UMI", "UB"] if barcodes else []), paired, barcodes)
        assert kept >= n // 6, (tag, kept)
        if not barcodes:   # -u: unaligned templates are not expected to come in adjacent records; both scans treat them differently
            du = d / "u"
            du.mkdir()
            compare(du, driver, bam, fa, ["-u"], paired, False)


@needs_ref
def test_bam_extractor_synthetic_emulated(tmp_path):
    synthetic_cases(tmp_path, emulated_bam_extractor(), 60)


@needs_ref
@pytest.mark.skipif(not os.path.exists("/root/reference/example/example.bam"), reason="the reference tree (example.bam) is not here")
def test_bam_extractor_example_emulated(tmp_path):
    kept = compare(tmp_path, emulated_bam_extractor(), "/root/reference/example/example.bam", "/root/reference/hg38_bcrtcr.fa", [], True, False)
    assert kept == 198


def test_bam_extractor_refuses_damaged_records(tmp_path):
    """a record whose lengths do not add up, and a file cut inside a record: exit 1 with the reason, not partial output with exit 0"""
    fa = str(tmp_path / "ref.fa")
    with gzip.open(os.path.join(ROOT, "data", "hg38_bcrtcr.fa.gz"), "rb") as f, open(fa, "wb") as g:
        g.write(f.read())
    chroms = []
    for line in open(fa):
        if line.startswith(">") and line.split()[1] not in chroms:
            chroms.append(line.split()[1])
    refs = [(c, 1 << 28) for c in chroms]
    good = bam_record("r1", 0, 0, 100, [(20, "M")], "ACGT" * 5, "I" * 20)
    bad = bytearray(bam_record("r2", 0, 0, 200, [(20, "M")], "ACGT" * 5, "I" * 20))
    bad[4 + 16:4 + 20] = struct.pack("<i", 5000)          # l_seq far beyond the record
    exe = emulated_bam_extractor()
    for name, payload, what in (("lens.bam", [good, bytes(bad)], "field lengths"), ("cut.bam", [good, good[:len(good) - 7]], "cut short")):
        bam = str(tmp_path / name)
        write_bam(bam, refs, payload)
        p = subprocess.run([exe, "-b", bam, "-f", fa, "-o", str(tmp_path / "o")], capture_output=True, text=True)
        assert p.returncode == 1 and what in p.stderr, (name, p.returncode, p.stderr[-300:])
