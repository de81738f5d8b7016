"""Throughput of t4_has_hit (stage-0 candidate test) on one MI355X vs the reference on one host thread.
usage: has_hit_bench.py [n_reads]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, t4libs, trust4_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
eng = trust4_amd.Engine(0)
ix = eng.index(9).set_params(27, 10, 0.9).load_ref_fasta(t4libs.REF_FA).commit()
rec = t4libs.Synth(20000, 1).next_reads(n // 2)                          # receptor reads (all candidates)
rnd = np.random.RandomState(3)
gen = np.frombuffer(b"ACGT", dtype=np.uint8)[rnd.randint(0, 4, size=(n, 150))]
gen = np.concatenate([gen, np.zeros((n, 1), np.uint8)], axis=1)         # random 150-mers (no candidates): the bulk of a stage-0 input
for name, arr in (("receptor reads", rec), ("random 150-mers", gen)):
    b = eng.upload(arr)
    ix.has_hit(b)
    t0 = time.perf_counter(); out = ix.has_hit(b); dt = time.perf_counter() - t0
    st = eng.stats()
    print("%-16s %8d reads: %.3f s (%.2f M reads/s; kernels %.0f ms), candidates %.1f %%" % (name, arr.shape[0], dt, arr.shape[0] / dt / 1e6, st["kernel_ms"], 100.0 * (out != 0).mean()))
    if t4libs.Ref.available():
        r = t4libs.Ref(9, t4libs.REF_FA, 27)
        m = min(20000, arr.shape[0])
        rows = t4libs.rows_to_strs(arr[:m])
        t0 = time.perf_counter(); exp = [r.has_hit_in_set(x, 0) for x in rows]; dc = time.perf_counter() - t0
        print("   reference, 1 thread: %.1f k reads/s; identical on the sample: %s" % (m / dc / 1e3, bool((np.array(exp) == out[:m]).all())))
