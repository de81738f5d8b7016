"""Per-phase cycle breakdown of the query kernel (development aid; needs the -DT4_PHASE_TIMING build)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["T4_LIB"] = sys.argv[1]
import numpy as np, t4libs, trust4_amd
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
eng = trust4_amd.Engine(0)
ref = eng.index(9).set_params(17, 10, 0.9).load_ref_fasta(t4libs.REF_FA).commit()
arr = t4libs.Synth(20000, 1).next_reads(n // 2)
b = eng.upload(arr)
ref.annotate_rough(b, fetch=False)
buf = (C.c_ulonglong * 32)()
dbg = (C.c_ulonglong * 8)()
eng.lib.t4_debug_phase_cycles(buf)
eng.lib.t4_debug_counters(dbg)
ref.annotate_rough(b, fetch=False)
eng.lib.t4_debug_phase_cycles(buf)
eng.lib.t4_debug_counters(dbg)
print("gap jobs %d (%.1f/read), banded %d (%.2f/read), wave-DP steps %d (%.0f/DP), scratch fallbacks %d, fallback cells %d, fallback cycles %.3e" % (
    dbg[0], dbg[0] / n, dbg[1], dbg[1] / n, dbg[2], dbg[2] / max(1, dbg[1] - dbg[3]), dbg[3], dbg[4], dbg[5]))
names = ["other", "seed", "expand", "sort", "stats", "runs", "bigsort", "chain", "ovsort", "score", "prefilter", "final", "annotate", "score:quick", "score:banded", "score:finish"]
tot = sum(buf[:16])
print(eng.stats())
for i, nm in enumerate(names):
    print("%-10s %6.2f%%  %.3e cycles" % (nm, 100.0 * buf[i] / tot, buf[i]))
