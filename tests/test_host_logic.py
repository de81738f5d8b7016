"""Host logic of the driver next to the oracle (CPU suite): the code trust4-hip's host threads run for ProcessRead -- IsMateOverlap with
its 16-bases-at-a-time mismatch count, IsLowComplexity, the read-through clip / merge / choice of a mate -- compiled from
trust4_amd/host/process_read.h into tests/host_probe.cpp together with oracle/t4_oracle.c, and the FASTA / FASTQ reader's three ways
through a file (raw, zlib, blocks that are recycled)."""
import gzip
import os
import random
import subprocess

from t4libs import ROOT


def _probe(tmp_path, flags=(), name="host_probe"):
    exe = str(tmp_path / name)
    subprocess.run(["gcc", "-O2", "-c", os.path.join(ROOT, "oracle", "t4_oracle.c"), "-o", str(tmp_path / "t4_oracle.o")], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17"] + list(flags) + ["-o", exe, os.path.join(ROOT, "tests", "host_probe.cpp"), str(tmp_path / "t4_oracle.o"), "-lz", "-lm", "-lpthread"], check=True)
    return exe


def test_process_read_of_the_driver_against_the_oracle(tmp_path):
    exe = _probe(tmp_path)
    for seed in (1, 2):
        p = subprocess.run([exe, "25000", str(seed)], stdout=subprocess.PIPE, text=True)
        assert p.returncode == 0 and p.stdout.startswith("ok pairs 25000"), p.stdout[-600:]
        counts = [int(x) for x in p.stdout.split(":")[1].split(";")[0].replace("stay", "").replace("read-through", "").replace("merged", "").replace("one-mate", "").split()]
        assert min(counts) > 300, p.stdout   # every branch of ProcessRead was taken many times


def test_process_read_without_sse2(tmp_path):
    """the portable path of IsMateOverlap's mismatch count (a host whose compiler does not define __SSE2__)"""
    exe = _probe(tmp_path, flags=["-U__SSE2__"], name="host_probe_portable")
    p = subprocess.run([exe, "12000", "7"], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0 and p.stdout.startswith("ok pairs 12000"), p.stdout[-600:]


def test_barcode_and_umi_numbering(tmp_path):
    """order of first appearance, packed keys and the map sharing one counter; 600 k strings force the flat table to grow several times"""
    exe = _probe(tmp_path)
    p = subprocess.run([exe, "numbering", "600000", "11"], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0 and p.stdout.startswith("ok numbering: 600000 strings"), p.stdout[-400:]
    distinct = int(p.stdout.split(",")[-1].split()[0])
    assert distinct > 150000   # (the table of 65 536 slots was outgrown)


def test_reader_raw_zlib_and_recycled_blocks(tmp_path):
    exe = _probe(tmp_path)
    rnd = random.Random(3)

    def fnv(records):   # the probe's checksum over "id|comment|seq|qual-or-dash" of every record (kseq's record model, ReadFiles.hpp:180-185)
        h = 1469598103934665603
        for r in records:
            for c in r.encode():
                h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h

    # FASTQ: 40 000 records (more than two reader blocks), CRLF here and there, a comment on some headers, no newline at the end
    recs, want_fq = [], []
    for i in range(40000):
        n = rnd.randint(1, 160)
        s = "".join(rnd.choice("ACGTN") for _ in range(n))
        q = "".join(chr(rnd.randint(35, 73)) for _ in range(n))
        eol = "\r\n" if i % 97 == 0 else "\n"
        recs.append("@r%d%s%s%s%s+%s%s%s" % (i, "/1" if i % 5 == 0 else "", " extra words" if i % 7 == 0 else "", eol, s + eol, eol, q, eol))
        want_fq.append("r%d|%s|%s|%s" % (i, "extra words" if i % 7 == 0 else "", s, q))
    fq = "".join(recs)[:-1]
    # FASTA: multi-line sequences, blank lines, lower case and '.' (kept: printable), records without a sequence
    parts, want_fa = [], []
    for i in range(3000):
        lines = ["".join(rnd.choice("ACGTacgtn.") for _ in range(rnd.randint(0, 70))) for _ in range(rnd.randint(0, 4))]
        parts.append(">s%d desc\n%s\n\n" % (i, "\n".join(lines)))
        want_fa.append("s%d|desc|%s|-" % (i, "".join(lines)))
    fa = "".join(parts)
    expected = {"x.fq": fnv(want_fq), "y.fa": fnv(want_fa)}
    for name, text in (("x.fq", fq), ("y.fa", fa)):
        plain, gz = str(tmp_path / name), str(tmp_path / (name + ".gz"))
        with open(plain, "w", newline="") as f:
            f.write(text)
        with gzip.open(gz, "wt", newline="") as f:
            f.write(text)
        p = subprocess.run([exe, "reader", plain, gz], stdout=subprocess.PIPE, text=True)
        assert p.returncode == 0 and p.stdout.startswith("ok reader: %d records" % (40000 if name == "x.fq" else 3000)), p.stdout[-600:]
        assert p.stdout.strip().endswith("fnv " + expected[name]), (p.stdout, expected[name])   # ... and they are the records that were written
