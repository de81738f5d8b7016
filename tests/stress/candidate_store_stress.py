#!/usr/bin/env python3
"""Stress runs of the ordered builder's window on the EMULATED engine (test infrastructure, run by hand; not collected by pytest).

Input: tools/t4synth over a gene set whose five chains hold the same genes (tests/test_stage1_e2e.py::_shared_constant_gene_fasta), so
that every clone shares one constant gene and a read inside it meets hundreds of contigs at a depth the emulator can run -- the
regimes of the candidate store (more than 50 candidates: order-dependent pre-filters; more than 100 groups of four hits: the
threshold follows the group statistics, exact replays, raised thresholds). Every run is compared with the reference binary
(oracle/_ref/trust4, the checker) byte for byte, and T4_VERIFY_WINDOW compares every served window entry with a fresh whole query.

    python tests/stress/candidate_store_stress.py SEED RUNS        # random sizes / seeds / policy switches
    python tests/stress/candidate_store_stress.py one PAIRS CLONES SEED [NAME=VALUE ...]

Round 5: 16 + 40 + 30 runs, all identical; round 6: see DESIGN 3g."""
import filecmp
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from test_stage1_e2e import REF_BIN, ROOT, _emulated_driver, _shared_constant_gene_fasta   # noqa: E402

KNOBS = [("T4_MAX_PENDING", ["1", "2", "4", "8"]), ("T4_LIGHT_AHEAD", ["0", "1", "3"]), ("T4_LIVE_LANES", ["1", "2", "4"]), ("T4_LIVE_HARVEST_DELAY", ["0", "3"]),
         ("T4_WINDOW", ["9", "33", "192"]), ("T4_QUERY_AHEAD", ["4", "30"]), ("T4_AQ_CAP_LIMIT", ["400", "1500"]), ("T4_AQ_CAND_CAP", ["64"]), ("T4_AQ_POOL_CAP", ["8", "64"]),
         ("T4_AQ_EXTEND_DEFER", ["0", "1", "16"]), ("T4_WIDE_MIN_HITS", ["300", "2000"]), ("T4_NO_MARKS", ["1"]), ("T4_NO_PREDICT", ["1"]), ("T4_NO_STABLE_STATS", ["1"]),
         ("T4_WIDE_PCAP", ["512"]), ("T4_WIDE_OFF", ["1"]), ("T4_LIVE_MIN_BATCH", ["1", "6"]), ("T4_CANDS_OFF", ["1"]),
         # round 6: the wide kernels behind every round, the sample of the partition plan, merges that end entries, the budget rule, restricted re-queries near the head only
         ("T4_WIDE_EAGER", ["1"]), ("T4_WIDE_SAMPLE", ["0", "64"]), ("T4_CONTIG_KILLS", ["1"]), ("T4_NO_EXACT_TOLERANCE", ["1"]), ("T4_RESTRICT_AHEAD", ["2", "6", "-2"])]


def one(pairs, clones, seed, env):
    """-> (identical, log tail)"""
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "shared.fa")
        _shared_constant_gene_fasta(fa)
        pre = os.path.join(d, "b")
        subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), str(clones), str(seed), pre], check=True, stdout=subprocess.DEVNULL)
        args = ["--skipMateExtension", "-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq"]
        subprocess.run([REF_BIN, "-t", "1"] + args + ["-o", os.path.join(d, "ref")], check=True, stderr=subprocess.DEVNULL)
        e = dict(os.environ, T4_TIMING="1", T4_VERIFY_WINDOW="1")
        e.update(env)
        p = subprocess.run([_emulated_driver(), "-t", "8"] + args + ["-o", os.path.join(d, "mine")], env=e, stderr=subprocess.PIPE, text=True)
        same = p.returncode == 0 and all(filecmp.cmp(os.path.join(d, "ref") + s, os.path.join(d, "mine") + s, shallow=False) for s in ("_raw.out", "_assembled_reads.fa", "_final.out"))
        same = same and "all equal to their cached results" in p.stderr
        lines = [l for l in p.stderr.split("\n") if "candidate store" in l or "T4_VERIFY_WINDOW" in l]
        return same, "\n".join(lines) if same else p.stderr[-3000:]


def main():
    if sys.argv[1] == "one":
        ok, log = one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), dict(kv.split("=", 1) for kv in sys.argv[5:]))
        print("identical" if ok else "DIFFERENT", log, sep="\n")
        sys.exit(0 if ok else 1)
    rnd = random.Random(int(sys.argv[1]))
    bad = 0
    for it in range(int(sys.argv[2])):
        pairs, clones, seed = rnd.choice([500, 800, 1100]), rnd.choice([150, 300, 500]), rnd.randrange(100, 100000)
        env = {k: rnd.choice(v) for k, v in KNOBS if rnd.random() < 0.2}
        ok, log = one(pairs, clones, seed, env)
        print("ok" if ok else "FAILED", it, pairs, clones, seed, env, flush=True)
        if not ok:
            bad += 1
            print(log, flush=True)
    print("runs that differed:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
