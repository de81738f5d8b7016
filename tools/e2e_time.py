"""Whole stage 1: trust4-hip vs the reference binary on the same synthetic pairs (parity + wall time)."""
import filecmp, gzip, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import t4libs, trust4_amd.build
pairs, clones, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1
threads = sys.argv[4] if len(sys.argv) > 4 else "8"
t4libs.build_checkers(); trust4_amd.build.build()
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "ref.fa")
with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g: shutil.copyfileobj(f, g)
r1, r2 = t4libs.Synth(clones, seed).next_pairs(pairs)
for name, arr in (("s_1.fq", r1), ("s_2.fq", r2)):
    with open(os.path.join(tmp, name), "w") as f:
        for i, s in enumerate(t4libs.rows_to_strs(arr)): f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
f1, f2 = os.path.join(tmp, "s_1.fq"), os.path.join(tmp, "s_2.fq")
ref_bin = os.path.join(ROOT, "oracle", "_ref", "trust4")
t0 = time.time(); subprocess.run([ref_bin, "-t", threads, "--skipMateExtension", "-f", fa, "-1", f1, "-2", f2, "-o", os.path.join(tmp, "ref")], check=True, stderr=subprocess.DEVNULL); t_ref = time.time() - t0
t0 = time.time(); p = subprocess.run([os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip"), "-t", threads, "--skipMateExtension", "-f", fa, "-1", f1, "-2", f2, "-o", os.path.join(tmp, "mine")], stderr=subprocess.PIPE, text=True); t_mine = time.time() - t0
if p.returncode:
    print("trust4-hip failed (%d):\n%s" % (p.returncode, "\n".join(p.stderr.strip().split("\n")[-12:]))); sys.exit(1)
same = all(filecmp.cmp(os.path.join(tmp, "ref" + s), os.path.join(tmp, "mine" + s), shallow=False) for s in ("_raw.out", "_assembled_reads.fa", "_final.out"))
print("pairs %d clones %d: reference -t %s %.1f s (%.0f pairs/s) | trust4-hip %.1f s (%.0f pairs/s) | identical=%s | contigs %d" % (
    pairs, clones, threads, t_ref, pairs / t_ref, t_mine, pairs / t_mine, same, open(os.path.join(tmp, "ref_raw.out")).read().count(">")))
print("\n".join(p.stderr.strip().split("\n")[-8:]))
