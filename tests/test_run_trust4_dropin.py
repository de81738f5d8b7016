"""Process-level drop-in: the reference's own `run-trust4` driver runs the whole pipeline twice -- once over the
reference binaries, once with `trust4` and `fastq-extractor` replaced by this repo's programs -- and every file it leaves behind must
be byte-identical (stage-0 candidates, stage-1 contigs, and everything the reference's annotator and report scripts derive from them).

`run-trust4` finds its programs next to itself, so each run gets a scratch directory holding a copy of the script (made at test
time, from /root/reference, into tmp -- nothing of the reference enters the repo), links to the reference's perl report scripts
and links named `trust4` / `fastq-extractor` / `annotator` that point at the binaries under test. Needs /root/reference and perl:
it runs in the development container (emulator build of the kernels, CPU) and, with `-m gpu`, against the hipcc build where the
reference tree exists."""
import filecmp
import gzip
import os
import shutil
import subprocess

import pytest

import t4libs
from t4libs import REF_FA, ROOT

REF_BIN = os.path.join(ROOT, "oracle", "_ref")
# the reference's driver script, report scripts, IMGT file and example reads: from the reference tree where it exists (this
# container), else from the copies integration/make_dropin.py staged next to the checker binaries (git-ignored oracle/_ref/pipeline,
# which travels to the GPU box)
REF_TREE = "/root/reference" if os.path.exists("/root/reference/run-trust4") else os.path.join(REF_BIN, "pipeline")
DROPIN = os.path.join(REF_BIN, "trust4-dropin")
needs_reference = pytest.mark.skipif(
    not (os.path.exists(os.path.join(REF_TREE, "run-trust4")) and shutil.which("perl")
         and all(os.path.exists(os.path.join(REF_BIN, b)) for b in ("trust4", "fastq-extractor", "annotator"))),
    reason="needs run-trust4 + report scripts (/root/reference or oracle/_ref/pipeline), perl and oracle/_ref/{trust4,fastq-extractor,annotator}")


def install_dir(path, trust4_bin, extractor_bin, bam_extractor_bin=None):
    os.makedirs(path)
    shutil.copy(os.path.join(REF_TREE, "run-trust4"), os.path.join(path, "run-trust4"))   # abs_path($0) must resolve to THIS directory
    for f in os.listdir(REF_TREE):
        if f.endswith(".pl"):
            os.symlink(os.path.join(REF_TREE, f), os.path.join(path, f))
    os.symlink(trust4_bin, os.path.join(path, "trust4"))
    os.symlink(extractor_bin, os.path.join(path, "fastq-extractor"))
    os.symlink(os.path.join(REF_BIN, "annotator"), os.path.join(path, "annotator"))
    if bam_extractor_bin:
        os.symlink(bam_extractor_bin, os.path.join(path, "bam-extractor"))
    return os.path.join(path, "run-trust4")


def run_both(tmp_path, my_trust4, my_extractor, args, my_bam_extractor=None):
    ref_bamx = os.path.join(REF_BIN, "bam-extractor")
    runs = {"ref": install_dir(str(tmp_path / "inst_ref"), os.path.join(REF_BIN, "trust4"), os.path.join(REF_BIN, "fastq-extractor"), ref_bamx if my_bam_extractor else None),
            "mine": install_dir(str(tmp_path / "inst_mine"), my_trust4, my_extractor, my_bam_extractor)}
    outs = {}
    for tag, script in runs.items():
        od = tmp_path / ("out_" + tag)
        od.mkdir()
        # the report scripts break count ties in perl hash order: pin the hash seed, as one would to compare two reference runs
        env = dict(os.environ, PERL_HASH_SEED="0", PERL_PERTURB_KEYS="0")
        p = subprocess.run(["perl", script] + args + ["-o", "T", "--od", str(od)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        outs[tag] = str(od)
    names = sorted(os.listdir(outs["ref"]))
    assert sorted(os.listdir(outs["mine"])) == names
    for must in ("T_raw.out", "T_final.out", "T_assembled_reads.fa", "T_annot.fa", "T_cdr3.out", "T_report.tsv", "T_airr.tsv"):
        assert must in names, (must, names)
    for f in names:
        assert filecmp.cmp(os.path.join(outs["ref"], f), os.path.join(outs["mine"], f), shallow=False), f
    return outs["ref"], names


def emulated_programs():
    from test_stage0_e2e import _emulated_extractor
    from test_stage1_e2e import _emulated_driver
    return _emulated_driver(), _emulated_extractor()


def cells_input(tmp_path, pairs, cells, seed):
    t4libs.build_checkers()
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = str(tmp_path / "c5")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", str(seed), pre, "--cells", str(cells)], check=True, stdout=subprocess.DEVNULL)
    return fa, pre


def example_args():
    return ["-f", os.path.join(REF_TREE, "hg38_bcrtcr.fa"), "--ref", os.path.join(REF_TREE, "human_IMGT+C.fa"),
            "-1", os.path.join(REF_TREE, "example", "example_1.fq"), "-2", os.path.join(REF_TREE, "example", "example_2.fq"), "-t", "2"]


def emulated_dropin():
    """the reference's main.cpp bound to the C ABI (integration/make_dropin.py), linked with the emulator build of the kernels"""
    import t4check
    lib = t4check.build_emulator_lib()
    exe = os.path.join(ROOT, "oracle", "_ref", "trust4-dropin-emu")   # reference-derived: beside the other checker binaries, never committed
    deps = [lib, os.path.join(ROOT, "integration", "t4_dropin.hpp"), os.path.join(ROOT, "integration", "make_dropin.py")]
    if not os.path.exists("/root/reference/main.cpp"):
        pytest.skip("the emulated drop-in is built from /root/reference/main.cpp (a stale binary is not run)")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["python3", os.path.join(ROOT, "integration", "make_dropin.py"), "--emu"], check=True, stdout=subprocess.DEVNULL)
    return exe


def synthetic_pe(tmp_path, pairs, clones, seed):
    t4libs.build_checkers()
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = str(tmp_path / "pe")
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), str(clones), str(seed), pre], check=True, stdout=subprocess.DEVNULL)
    return fa, pre


needs_main_cpp = pytest.mark.skipif(not os.path.exists("/root/reference/main.cpp"), reason="the emulated drop-in is built from /root/reference/main.cpp")


@needs_reference
@needs_main_cpp
def test_example_pipeline_default_options_dropin_emulated(tmp_path):
    """BASELINE config[0] under run-trust4's DEFAULT options (no --skipMateExtension): stage 1 is the reference's own main.cpp
    with its three hot loops bound to the C ABI (rough annotation, the AddRead pass, AssignRead + RecomputePosWeight) and its
    mate-pair extension tail left in place on the host; `_final.out` and everything derived from it byte-identical."""
    _, extractor = emulated_programs()
    od, names = run_both(tmp_path, emulated_dropin(), extractor, example_args())
    assert not filecmp.cmp(os.path.join(od, "T_raw.out"), os.path.join(od, "T_final.out"), shallow=False)   # the tail did run


@needs_reference
@needs_main_cpp
def test_synthetic_pe_default_options_dropin_emulated(tmp_path):
    """a synthetic paired-end set (SURVEY 8d recipe) through the whole pipeline with default options"""
    _, extractor = emulated_programs()
    fa, pre = synthetic_pe(tmp_path, 1200, 24, 21)
    od, names = run_both(tmp_path, emulated_dropin(), extractor, ["-f", fa, "--ref", os.path.join(REF_TREE, "human_IMGT+C.fa"), "-1", pre + "_1.fq", "-2", pre + "_2.fq", "-t", "3"])
    assert not filecmp.cmp(os.path.join(od, "T_raw.out"), os.path.join(od, "T_final.out"), shallow=False)


@needs_reference
@needs_main_cpp
def test_barcode_pipeline_dropin_emulated(tmp_path):
    """the same binding with 10x-style input: the one SeqSet keyed by barcode (main.cpp:1556-1559) behind t4_assembler_*"""
    _, extractor = emulated_programs()
    fa, pre = cells_input(tmp_path, 200, 5, 19)
    run_both(tmp_path, emulated_dropin(), extractor, ["-f", fa, "--ref", os.path.join(REF_TREE, "human_IMGT+C.fa"), "-1", pre + "_1.fq", "-2", pre + "_2.fq",
                                                      "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa", "-t", "2"])


@needs_reference
def test_example_pipeline_emulated(tmp_path):
    """BASELINE config[0] through run-trust4 (stage 0 -> 1 -> 2 -> 3). Bulk mode: --skipMateExtension, the mate-graph tail of
    stage 1 is not built."""
    trust4, extractor = emulated_programs()
    od, names = run_both(tmp_path, trust4, extractor, example_args() + ["--skipMateExtension"])
    assert "T_toassemble_1.fq" in names and os.path.getsize(os.path.join(od, "T_report.tsv")) > 100


@needs_reference
def test_barcode_pipeline_emulated(tmp_path):
    """10x-style input (barcode + UMI files) through run-trust4: stage 1 runs as the reference does by default (no option added),
    the barcode reports are compared as well."""
    trust4, extractor = emulated_programs()
    fa, pre = cells_input(tmp_path, 300, 6, 9)
    od, names = run_both(tmp_path, trust4, extractor, ["-f", fa, "--ref", os.path.join(REF_TREE, "human_IMGT+C.fa"), "-1", pre + "_1.fq", "-2", pre + "_2.fq",
                                                       "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa", "-t", "2"])
    assert "T_barcode_report.tsv" in names and "T_toassemble_bc.fa" in names


@needs_reference
@pytest.mark.skipif(not os.path.exists(os.path.join(REF_BIN, "bam-extractor")), reason="oracle/_ref/bam-extractor not built")
def test_example_bam_pipeline_emulated(tmp_path):
    """the same example from its BAM file: run-trust4 -b, stage 0 by bam-extractor (replaced by bam-extractor-hip)"""
    from test_bam_extractor import emulated_bam_extractor
    trust4, extractor = emulated_programs()
    od, names = run_both(tmp_path, trust4, extractor, ["-f", os.path.join(REF_TREE, "hg38_bcrtcr.fa"), "--ref", os.path.join(REF_TREE, "human_IMGT+C.fa"),
                                                       "-b", os.path.join(REF_TREE, "example", "example.bam"), "-t", "1", "--skipMateExtension"], emulated_bam_extractor())
    assert "T_toassemble_1.fq" in names and os.path.getsize(os.path.join(od, "T_report.tsv")) > 100


@needs_reference
@pytest.mark.skipif(not os.path.exists(os.path.join(REF_BIN, "bam-extractor")), reason="oracle/_ref/bam-extractor not built")
def test_barcode_bam_pipeline_emulated(tmp_path):
    """10x-style BAM input: run-trust4 -b in.bam --barcode CB --UMI UB (barcode and UMI from the BAM fields)"""
    import test_bam_extractor as tb
    fa = str(tmp_path / "ref.fa")
    with gzip.open(REF_FA, "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    bam = str(tmp_path / "in.bam")
    tb.synthetic_bam(bam, fa, True, 12, True, 80)
    trust4, extractor = emulated_programs()
    od, names = run_both(tmp_path, trust4, extractor, ["-f", fa, "--ref", os.path.join(REF_TREE, "human_IMGT+C.fa"), "-b", bam, "--barcode", "CB", "--UMI", "UB", "-t", "1"],
                         tb.emulated_bam_extractor())
    assert "T_toassemble_bc.fa" in names and "T_barcode_report.tsv" in names


@pytest.mark.gpu
@needs_reference
def test_example_pipeline_gpu(tmp_path):
    import trust4_amd.build as b
    b.build()
    run_both(tmp_path, os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip"), os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip"),
             example_args() + ["--skipMateExtension"])


needs_dropin = pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/trust4-dropin not built (integration/make_dropin.py, needs /root/reference at build time)")


@pytest.mark.gpu
@needs_reference
@needs_dropin
def test_example_pipeline_default_options_dropin_gpu(tmp_path):
    """run-trust4 with its default options on the example, stage 1 = the reference's main.cpp bound to libt4hip.so"""
    od, names = run_both(tmp_path, DROPIN, os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip"), example_args())
    assert not filecmp.cmp(os.path.join(od, "T_raw.out"), os.path.join(od, "T_final.out"), shallow=False)


@pytest.mark.gpu
@needs_reference
@needs_dropin
def test_synthetic_pe_default_options_dropin_gpu(tmp_path):
    """4 k synthetic pairs with default options: `_final.out` after the mate-pair extension identical"""
    fa, pre = synthetic_pe(tmp_path, 4000, 80, 22)
    od, names = run_both(tmp_path, DROPIN, os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip"),
                         ["-f", fa, "--ref", os.path.join(REF_TREE, "human_IMGT+C.fa"), "-1", pre + "_1.fq", "-2", pre + "_2.fq", "-t", "8"])
    assert not filecmp.cmp(os.path.join(od, "T_raw.out"), os.path.join(od, "T_final.out"), shallow=False)
