#!/usr/bin/env python3
"""tools/c2_digests.py -- digests of the REFERENCE's stage-1 outputs on a BASELINE config, for bench.py's `c2` leg.

The reference binary (oracle/_ref/trust4, built by oracle/Makefile from /root/reference as it lies) takes 10-20 minutes on
config C2 (1 M pairs), more than a default bench run can spend next to the GPU run, so its outputs are digested once, here,
and the md5 sums are committed (tests/golden/c2_digests.json). bench.py regenerates the same input files with
tools/t4synth (the SURVEY 8d recipe is deterministic: the input md5 sums are checked too), runs trust4-hip on them and compares.

  python tools/c2_digests.py [--config c2|c3p5] [--threads N]

Test infrastructure (a checker of the product), never on the product path.
"""
import argparse
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "trust4")
OUT = os.path.join(ROOT, "tests", "golden", "c2_digests.json")
SUFFIXES = ("_raw.out", "_assembled_reads.fa", "_final.out")
# name -> (pairs, clones, seed, prefix pairs or 0): SURVEY.md 8(d)
# (c5m5: the C5 recipe -- barcode + UMI files, 50 k cells x 2 clones, seed 4 -- at 5 M pairs; barcode mode, no --skipMateExtension)
CELL_CONFIGS = {"c5m5": (5000000, 50000, 4), "c5m20": (20000000, 50000, 4)}   # (c5m20: 40 % of config C5's pairs over all of its cells, VERDICT r5 #6c)
CONFIGS = {"c2": (1000000, 20000, 1, 0), "c2mini": (20000, 400, 1, 0), "c2micro": (400, 8, 1, 0),   # (c2mini / c2micro: no digests -- stand-ins that let bench.py's C2-as-the-workload path be exercised in seconds, T4_BENCH_C2_STANDIN)
           "c3p5": (20000000, 200000, 2, 5000000), "c3p2": (20000000, 200000, 2, 2000000), "c3p05": (20000000, 200000, 2, 500000)}


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def make_inputs(tmp, name):
    """the config's FASTQ files (a prefix of the config's read stream when the config says so) + the plain gene FASTA"""
    pairs, clones, seed, prefix = CONFIGS[name]
    fa = os.path.join(tmp, "ref.fa")
    with gzip.open(os.path.join(ROOT, "data", "hg38_bcrtcr.fa.gz"), "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = os.path.join(tmp, name)
    n = prefix if prefix else pairs
    # t4synth draws the clone table from (clones, seed) and then the pairs one after the other: asking for n pairs of the
    # config's clone table IS the config's first n pairs
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(n), str(clones), str(seed), pre], check=True, stdout=subprocess.DEVNULL)
    return fa, pre + "_1.fq", pre + "_2.fq", n


def make_cell_inputs(tmp, name):
    """the C5-recipe files of a CELL_CONFIGS entry: reads, mates, barcodes, UMIs + the plain gene FASTA"""
    pairs, cells, seed = CELL_CONFIGS[name]
    fa = os.path.join(tmp, "ref.fa")
    with gzip.open(os.path.join(ROOT, "data", "hg38_bcrtcr.fa.gz"), "rb") as f, open(fa, "wb") as g:
        shutil.copyfileobj(f, g)
    pre = os.path.join(tmp, name)
    subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", str(seed), pre, "--cells", str(cells)], check=True, stdout=subprocess.DEVNULL)
    return fa, pre + "_1.fq", pre + "_2.fq", pre + "_bc.fa", pre + "_umi.fa", pairs


def digest_cells(name, threads, keep):
    """barcode-mode digest: oracle/_ref/trust4 --barcode --UMI on the C5-recipe sample (38 minutes at -t 8 for c5m5)"""
    tmp = keep or tempfile.mkdtemp(prefix="t4c5_")
    os.makedirs(tmp, exist_ok=True)
    try:
        fa, f1, f2, bc, umi, n = make_cell_inputs(tmp, name)
        out = os.path.join(tmp, "ref")
        t0 = time.time()
        subprocess.run([REF_BIN, "-t", str(threads), "-f", fa, "-1", f1, "-2", f2, "--barcode", bc, "--UMI", umi, "-o", out], check=True, stderr=subprocess.DEVNULL)
        rec = {"pairs": n, "cells": CELL_CONFIGS[name][1], "seed": CELL_CONFIGS[name][2], "inputs_md5": [md5(x) for x in (f1, f2, bc, umi)],
               "modes": {"barcode": {"md5": {s: md5(out + s) for s in SUFFIXES}, "reference_seconds": round(time.time() - t0, 1), "reference_threads": threads}}}
        allrec = json.load(open(OUT)) if os.path.exists(OUT) else {}
        allrec[name] = rec
        with open(OUT, "w") as f:
            json.dump(allrec, f, indent=1, sort_keys=True)
            f.write("\n")
    finally:
        if not keep:
            shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + sorted(CELL_CONFIGS))
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--keep", default="", help="directory to keep the files in (default: a temporary one)")
    ap.add_argument("--modes", default="skipMateExtension,default", help="which option sets to digest")
    args = ap.parse_args()
    if args.config in CELL_CONFIGS:
        return digest_cells(args.config, args.threads, args.keep)
    tmp = args.keep or tempfile.mkdtemp(prefix="t4c2_")
    os.makedirs(tmp, exist_ok=True)
    try:
        fa, f1, f2, n = make_inputs(tmp, args.config)
        rec = {"pairs": n, "clones": CONFIGS[args.config][1], "seed": CONFIGS[args.config][2], "inputs_md5": [md5(f1), md5(f2)], "modes": {}}
        for mode, extra in (("skipMateExtension", ["--skipMateExtension"]), ("default", [])):
            if mode not in args.modes.split(","):
                continue
            out = os.path.join(tmp, "ref_" + mode)
            t0 = time.time()
            subprocess.run([REF_BIN, "-t", str(args.threads)] + extra + ["-f", fa, "-1", f1, "-2", f2, "-o", out], check=True, stderr=subprocess.DEVNULL)
            rec["modes"][mode] = {"md5": {s: md5(out + s) for s in SUFFIXES}, "bytes": {s: os.path.getsize(out + s) for s in SUFFIXES},
                                  "reference_seconds": round(time.time() - t0, 1), "reference_threads": args.threads}
            print(mode, rec["modes"][mode], file=sys.stderr)
        allrec = json.load(open(OUT)) if os.path.exists(OUT) else {}
        if args.config in allrec and allrec[args.config].get("inputs_md5") == rec["inputs_md5"]:   # keep the modes digested earlier
            for m, v in allrec[args.config].get("modes", {}).items():
                rec["modes"].setdefault(m, v)
        allrec[args.config] = rec
        with open(OUT, "w") as f:
            json.dump(allrec, f, indent=1, sort_keys=True)
            f.write("\n")
    finally:
        if not args.keep:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
