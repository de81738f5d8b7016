"""Whole stage 1 in barcode mode: trust4-hip vs the reference binary on the same synthetic 10x-style input
(tools/t4synth --cells; SURVEY.md 8(d) config C5 recipe) -- parity + wall time + per-phase log stamps.
usage: e2e_cells_time.py n_pairs n_cells [seed] [ref_threads]"""
import filecmp, gzip, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import t4libs, trust4_amd.build
pairs, cells, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 4
threads = sys.argv[4] if len(sys.argv) > 4 else "8"
t4libs.build_checkers(); trust4_amd.build.build()
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "ref.fa")
with gzip.open(t4libs.REF_FA, "rb") as f, open(fa, "wb") as g: shutil.copyfileobj(f, g)
pre = os.path.join(tmp, "c5")
subprocess.run([os.path.join(ROOT, "tools", "t4synth"), fa, str(pairs), "0", str(seed), pre, "--cells", str(cells)], check=True)
args = ["-f", fa, "-1", pre + "_1.fq", "-2", pre + "_2.fq", "--barcode", pre + "_bc.fa", "--UMI", pre + "_umi.fa"]
ref_bin = os.path.join(ROOT, "oracle", "_ref", "trust4")
t0 = time.time(); pr = subprocess.run([ref_bin, "-t", threads] + args + ["-o", os.path.join(tmp, "ref")], check=True, stderr=subprocess.PIPE, text=True); t_ref = time.time() - t0
t0 = time.time(); p = subprocess.run([os.path.join(ROOT, "trust4_amd", "bin", "trust4-hip"), "-t", threads] + args + ["-o", os.path.join(tmp, "mine")], check=True, stderr=subprocess.PIPE, text=True); t_mine = time.time() - t0
same = all(filecmp.cmp(os.path.join(tmp, "ref" + s), os.path.join(tmp, "mine" + s), shallow=False) for s in ("_raw.out", "_assembled_reads.fa", "_final.out"))
print("pairs %d cells %d: reference -t %s %.1f s (%.0f pairs/s) | trust4-hip %.1f s (%.0f pairs/s) | identical=%s | contigs %d" % (
    pairs, cells, threads, t_ref, pairs / t_ref, t_mine, pairs / t_mine, same, open(os.path.join(tmp, "ref_raw.out")).read().count(">")))
print("--- reference log"); print("\n".join(pr.stderr.strip().split("\n")[-9:]))
print("--- trust4-hip log"); print("\n".join([l for l in p.stderr.strip().split("\n") if "timing" in l or "Start" in l or "Found" in l] + p.stderr.strip().split("\n")[-10:]))
shutil.rmtree(tmp, ignore_errors=True)
