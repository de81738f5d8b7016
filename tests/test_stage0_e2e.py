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


def run_case(tmp_path, driver, mode):
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pairs = stage0_input(11, 300, 1500)
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
    assert 300 <= n_kept < len(pairs) // 2


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not shipped")
@pytest.mark.parametrize("mode", ["paired", "single"])
def test_extractor_matches_reference_binary(tmp_path, mode):
    import trust4_amd.build as b
    b.build()
    run_case(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip"), mode)


@pytest.mark.skipif(not os.path.exists(REF_EXT), reason="oracle/_ref/fastq-extractor not built")
def test_extractor_emulated(tmp_path):
    """the same driver source linked against the emulator build of the kernels (test infrastructure)"""
    import t4check
    lib = t4check.build_emulator_lib()
    exe = os.path.join(ROOT, "tests", "hipemu", "fastq-extractor-hip-emu")
    src = os.path.join(ROOT, "trust4_amd", "host", "fastq_extractor_main.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src, "-L" + os.path.dirname(lib), "-lt4hip_emu",
                        "-Wl,-rpath," + os.path.dirname(lib), "-lz", "-lpthread"], check=True)
    run_case(tmp_path, exe, "paired")
