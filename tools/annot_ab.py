"""Development aid: kernel time of the rough-annotation pass for one build of the library (T4_LIB), reads from a saved array."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import t4libs
import trust4_amd
reads = np.load(sys.argv[1])
eng = trust4_amd.Engine(0)
ref = eng.index(9).set_params(17, 10, 0.9).load_ref_fasta(t4libs.REF_FA).commit()
batch = eng.upload(reads)
ref.annotate_rough(batch, fetch=False)
eng.check(eng.lib.t4_sync(eng.h))
kms = []
for _ in range(3):
    ref.annotate_rough(batch, fetch=False)
    kms.append(eng.stats()["chain_kernel_ms"])
print(sys.argv[2], "kernel ms %.2f %.2f %.2f" % tuple(kms), "tier_reads", eng.stats()["tier_reads"], flush=True)
