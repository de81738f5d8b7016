"""Stage 0: fastq-extractor-hip vs the reference fastq-extractor on the same synthetic FASTQ (receptor pairs diluted in
random pairs) -- parity + wall time. usage: stage0_time.py n_pairs receptor_fraction [ref_threads]"""
import filecmp, gzip, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, t4libs, trust4_amd.build
n, frac = int(sys.argv[1]), float(sys.argv[2])
threads = sys.argv[3] if len(sys.argv) > 3 else "8"
t4libs.build_checkers(); trust4_amd.build.build()
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "ref.fa")
with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g: shutil.copyfileobj(f, g)
nrec = int(n * frac)
r1, r2 = t4libs.Synth(2000, 1).next_pairs(nrec)
rnd = np.random.RandomState(5)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
g1, g2 = acgt[rnd.randint(0, 4, size=(n - nrec, 150))], acgt[rnd.randint(0, 4, size=(n - nrec, 150))]
order = rnd.permutation(n)
for name, rec, gen in (("in_1.fq", r1, g1), ("in_2.fq", r2, g2)):
    rows = [bytes(x[:150]) for x in rec] + [x.tobytes() for x in gen]
    with open(os.path.join(tmp, name), "wb") as f:
        q = b"F" * 150
        for i in order:
            f.write(b"@q%d\n%s\n+\n%s\n" % (i, rows[i], q))
f1, f2 = os.path.join(tmp, "in_1.fq"), os.path.join(tmp, "in_2.fq")
args = ["-f", fa, "-1", f1, "-2", f2]
t0 = time.time(); subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fastq-extractor"), "-t", threads] + args + ["-o", os.path.join(tmp, "ref")], check=True, stderr=subprocess.DEVNULL); t_ref = time.time() - t0
t0 = time.time(); p = subprocess.run([os.path.join(ROOT, "trust4_amd", "bin", "fastq-extractor-hip")] + args + ["-o", os.path.join(tmp, "mine")], check=True, stderr=subprocess.PIPE, text=True); t_mine = time.time() - t0
same = all(filecmp.cmp(os.path.join(tmp, "ref" + s), os.path.join(tmp, "mine" + s), shallow=False) for s in ("_1.fq", "_2.fq"))
print("pairs %d (%.0f %% receptor): reference -t %s %.1f s (%.0f k pairs/s) | fastq-extractor-hip %.1f s (%.0f k pairs/s) | identical=%s" % (
    n, 100 * frac, threads, t_ref, n / t_ref / 1e3, t_mine, n / t_mine / 1e3, same))
print(p.stderr.strip().split("\n")[-1])
shutil.rmtree(tmp, ignore_errors=True)
