"""Two rough-annotation passes over n synthetic reads on cuda:0 with the library T4_LIB points at (development aid: the
process rocprofv3 wraps for PMC passes and the unit of the A/B scripts). usage: gpu_pass.py [n_reads] [passes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import t4libs, trust4_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = trust4_amd.Engine(0)
ref = eng.index(9).set_params(17, 10, 0.9).load_ref_fasta(t4libs.REF_FA).commit()
b = eng.upload(t4libs.Synth(20000, 1).next_reads(n // 2))
ms = []
for _ in range(passes):
    ref.annotate_rough(b, fetch=False)
    ms.append(eng.stats()["kernel_ms"])
print("reads %d kernel_ms %s best %.2f (%.2f M reads/s)" % (n, ["%.2f" % x for x in ms], min(ms), n / min(ms) / 1e3))
