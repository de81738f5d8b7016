"""21-mer counting + per-read count statistics of n synthetic reads on cuda:0 (t4_kmer_count_*), next to the compiled reference's
KmerCount on one host thread over a sample. usage: kmer_count_bench.py [n_reads] [cpu_sample]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, t4libs, trust4_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
eng = trust4_amd.Engine(0)
arr = t4libs.Synth(20000, 1).next_reads(n // 2)
b = eng.upload(arr)
kc = eng.kmer_counter(21, max_kmers=130 * n // 4)
eng.check(eng.lib.t4_sync(eng.h))
t0 = time.perf_counter(); kc.add(b); t1 = time.perf_counter()
mn, md, av, ln = kc.stats(b); t2 = time.perf_counter()
print("GPU: %d reads, 21-mers counted in %.3f s (%.1f M reads/s), stats in %.3f s (%.1f M reads/s, incl. D2H), %d distinct k-mers, mean of min counts %.1f" % (
    n, t1 - t0, n / (t1 - t0) / 1e6, t2 - t1, n / (t2 - t1) / 1e6, kc.distinct(), float(mn.mean())))
if t4libs.Ref.available():
    reads = t4libs.rows_to_strs(arr[:sample])
    r = t4libs.KmerCountChecker(21, True)
    t0 = time.perf_counter()
    for x in reads: r.add(x)
    t1 = time.perf_counter()
    for x in reads[:20000]: r.stats(x, None)
    t2 = time.perf_counter()
    print("reference KmerCount, 1 thread (through ctypes): AddCount %.0f k reads/s, GetCountStatsAndTrim %.0f k reads/s" % (len(reads) / (t1 - t0) / 1e3, 20000 / (t2 - t1) / 1e3))
