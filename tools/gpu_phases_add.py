"""Per-phase cycles of the AddRead query kernel (mode 4) over a bulk-mode assembly (development aid; -DT4_PHASE_TIMING build)."""
import os, sys, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["T4_LIB"] = sys.argv[1]
import t4libs, trust4_amd
from test_assembler_emu import make_reads
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
eng = trust4_amd.Engine(0)
reads = make_reads(1, n_pairs, 200)
asm = trust4_amd.Assembler(eng, 9)
buf = (C.c_ulonglong * 20)()
eng.lib.t4_debug_phase_cycles(buf)
t0 = time.time(); nq = 0
for i, rd in enumerate(reads):
    if i > 0 and rd == reads[i - 1]:
        continue
    ret, st = asm.add_read(rd, "", 0, -1, 1, 0, 0.9); nq += 1
    if ret < 0:
        asm.input_novel_read("Novel", rd, 1, -1)
dt = time.time() - t0
eng.lib.t4_debug_phase_cycles(buf)
names = ["other", "seed", "expand", "sort", "stats", "runs", "bigsort", "chain", "ovsort", "score", "prefilter", "final", "annotate", "score:quick", "score:banded", "score:finish", "extend", "after-extend", "-", "-"]
tot = sum(buf[:20])
print("queries %d in %.2f s (%.2f ms each), contigs %d, counters %s" % (nq, dt, 1e3 * dt / nq, asm.size(), asm.counters()))
for i, nm in enumerate(names):
    if buf[i]:
        print("%-13s %6.2f%%  %.3e cycles  (%.0f per query)" % (nm, 100.0 * buf[i] / tot, buf[i], buf[i] / nq))
