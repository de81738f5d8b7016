#!/usr/bin/env python3
"""integration/make_dropin.py -- builds `trust4-dropin`: the REFERENCE's own main.cpp with its three hot loops bound to libt4hip.so.

The north star's drop-in ("keeping the run-trust4 CLI and the stage-1 on-disk inputs/outputs so it drops in under main.cpp: host
code stays C++ calling HIP through a thin C-ABI layer") shown on the reference's driver itself: a copy of /root/reference/main.cpp
gets the seven one-line edits below (anchored on unique strings of the file, each checked to occur exactly as expected), is compiled
against integration/t4_dropin.hpp (the binding) and linked with libt4hip.so (or, for the CPU test suite, with the emulator build of
the same kernels). Everything else -- ProcessRead, counting, sorting, trimming, the loop bodies, ExtendSeqFromReads,
RemoveRedundantSeq, the writers -- is the reference's code, compiled where it lies. The patched source is a temporary file; nothing
of the reference is stored in the repository, neither source nor binary (every output path below is git-ignored).

  python integration/make_dropin.py --product --ref /path/to/TRUST4     ->  trust4_amd/bin/trust4-dropin
      what a user of run-trust4 builds: the drop-in `trust4` for their own TRUST4 tree (point run-trust4's $WD/trust4 at it)
  python integration/make_dropin.py [--ref /root/reference] [-o PATH]   ->  oracle/_ref/trust4-dropin (+ oracle/_ref/pipeline)
      the same binary among the checker artefacts (travels to the GPU box with run-trust4 & co. for the process-level tests)
  python integration/make_dropin.py --emu                               ->  oracle/_ref/trust4-dropin-emu
      linked with the emulator build of the kernels (CPU test suite)
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (description, where to start looking (regex, or None = file start), regex to replace once after that point, replacement)
EDITS = [
    ("binding header after SeqSet.hpp, which it reads the members of",
     None, r'#include "SeqSet\.hpp"', '#define private public\n#include "SeqSet.hpp"\n#undef private\n#include "t4_dropin.hpp"'),
    ("the novel-contig set lives behind the C ABI (main.cpp:642)",
     None, r'SeqSet seqSet\( indexKmerLength \) ;', 't4bind::SeqSetProxy seqSet( indexKmerLength ) ;'),
    ("rough annotation of every distinct read on the GPU (main.cpp:1084-1120)",
     r'// Quickly annoate the reads\.', r'if \( threadCnt <= 1 \)', 'if ( t4bind::RoughAnnotate( refSet, sortedReads, readCnt ) ) ; else if ( threadCnt <= 1 )'),
    ("the assembly loop tells the binding where it stands (main.cpp:1585)",
     r'int prevAddRet = -1 ;', r'static struct _overlap geneOverlap\[4\] ;',
     'static struct _overlap geneOverlap[4] ; t4bind::Step( seqSet, sortedReads, i, readCnt, constantGeneEnd, threadCnt ) ;'),
    ("the host-side tail takes the engine's contigs over (main.cpp:2048)",
     None, r'extendedSeq\.InputSeqSet\( seqSet, false \) ;', 'extendedSeq.InputSeqSet( seqSet.Materialize(), false ) ;'),
    ("AssignRead of every assembled read on the GPU (main.cpp:2075-2116)",
     r'extendedSeq\.SetNovelSeqSimilarity\( 0\.95 \) ;', r'if \( threadCnt <= 1 \)',
     'if ( t4bind::AssignReads( extendedSeq, assembledReads, assembledReadCnt ) ) ; else if ( threadCnt <= 1 )'),
    ("RecomputePosWeight on the GPU (main.cpp:2118)",
     None, r'extendedSeq\.RecomputePosWeight\( assembledReads \) ;', 't4bind::RecomputePosWeight( extendedSeq, assembledReads ) ;'),
]


def patch(src):
    for what, start, pat, rep in EDITS:
        at = 0
        if start is not None:
            m = re.search(start, src)
            if not m:
                raise SystemExit("make_dropin: anchor %r not found (%s)" % (start, what))
            if len(re.findall(start, src)) != 1:
                raise SystemExit("make_dropin: anchor %r is not unique (%s)" % (start, what))
            at = m.end()
        elif len(re.findall(pat, src)) != 1:
            raise SystemExit("make_dropin: %r occurs %d times, expected once (%s)" % (pat, len(re.findall(pat, src)), what))
        m = re.compile(pat).search(src, at)
        if not m:
            raise SystemExit("make_dropin: %r not found (%s)" % (pat, what))
        src = src[:m.start()] + rep + src[m.end():]
    return src


def stage_pipeline(ref, dst):
    """run-trust4, its perl report scripts, the annotator's IMGT reference and the example reads next to the checker binaries, so
    that the process-level tests also run where /root/reference does not exist (the GPU box)."""
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(ref):
        if f == "run-trust4" or f.endswith(".pl") or f in ("human_IMGT+C.fa", "hg38_bcrtcr.fa"):
            if os.path.exists(os.path.join(dst, f)):
                os.remove(os.path.join(dst, f))
            shutil.copy(os.path.join(ref, f), os.path.join(dst, f))
            os.chmod(os.path.join(dst, f), 0o755 if f == "run-trust4" else 0o644)
    ex = os.path.join(ref, "example")
    os.makedirs(os.path.join(dst, "example"), exist_ok=True)
    for f in ("example_1.fq", "example_2.fq", "example.bam"):
        if os.path.exists(os.path.join(ex, f)):
            if os.path.exists(os.path.join(dst, "example", f)):
                os.remove(os.path.join(dst, "example", f))
            shutil.copy(os.path.join(ex, f), os.path.join(dst, "example", f))
            os.chmod(os.path.join(dst, "example", f), 0o644)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--emu", action="store_true", help="link the emulator build of the kernels (tests/hipemu/libt4hip_emu.so; CPU test suite)")
    ap.add_argument("--product", action="store_true", help="write trust4_amd/bin/trust4-dropin (the user-facing build, beside trust4-hip)")
    ap.add_argument("-o", default="")
    args = ap.parse_args()
    if not os.path.exists(os.path.join(args.ref, "main.cpp")):
        print("make_dropin: %s absent, keeping the prebuilt binary (if any)" % args.ref, file=sys.stderr)
        return
    libdir = os.path.join(ROOT, "tests", "hipemu") if args.emu else os.path.join(ROOT, "trust4_amd")
    lib = "t4hip_emu" if args.emu else "t4hip"
    out = args.o or (os.path.join(ROOT, "oracle", "_ref", "trust4-dropin-emu") if args.emu else
                     os.path.join(ROOT, "trust4_amd", "bin", "trust4-dropin") if args.product else os.path.join(ROOT, "oracle", "_ref", "trust4-dropin"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(os.path.join(args.ref, "main.cpp")) as f:
        src = patch(f.read())
    tmp = tempfile.mkdtemp(prefix="t4dropin_")
    try:
        cpp = os.path.join(tmp, "main_dropin.cpp")
        with open(cpp, "w") as f:
            f.write(src)
        cmd = ["g++", "-O3", "-w", "-std=c++11", "-I" + args.ref, "-I" + os.path.join(ROOT, "integration"), "-I" + os.path.join(ROOT, "include"), "-o", out, cpp,
               "-L" + libdir, "-l" + lib, "-Wl,-rpath," + libdir, "-Wl,-rpath,$ORIGIN/../../trust4_amd", "-Wl,-rpath,$ORIGIN/..", "-lpthread", "-lz"]
        subprocess.run(cmd, check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not args.emu and not args.product:
        stage_pipeline(args.ref, os.path.join(ROOT, "oracle", "_ref", "pipeline"))
    print(out)


if __name__ == "__main__":
    main()
