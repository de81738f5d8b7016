"""Quick GPU sanity + timing probe (used during development via gpurun)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import t4libs, t4check, trust4_amd
t4libs.build_checkers()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
eng = trust4_amd.Engine(0)
print("CUs", eng.cus())
ref = eng.index(9).set_params(17, 10, 0.9).load_ref_fasta(t4libs.REF_FA).commit()
arr = t4libs.Synth(2000, 1).next_reads(n // 2)
t0 = time.time(); b = eng.upload(arr); print("upload s", time.time() - t0)
for it in range(3):
    t0 = time.time(); ann = ref.annotate_rough(b, fetch=(it == 2)); dt = time.time() - t0
    print("annotate %d reads: %.3f s wall, stats %s" % (arr.shape[0], dt, eng.stats()))
o = t4libs.Oracle(9, t4libs.REF_FA, 17)
m = min(2000, arr.shape[0])
exp, hp, tot = o.annotate_batch(arr[:m], arr.shape[1], m)
bad = int((ann["seqIdx"][:m] != exp["seqIdx"]).sum())
mask = exp["seqIdx"] != -1
for f in ("readStart", "readEnd", "seqStart", "seqEnd", "strand", "matchCnt", "indelCnt", "similarity"):
    bad += int((ann[f][:m][mask] != exp[f][mask]).sum())
print("parity mismatches on first %d reads: %d" % (m, bad))
