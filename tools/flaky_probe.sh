set -e
mkdir -p /tmp/fl && cd /tmp/fl
python - <<'PY'
import sys, os, gzip, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tests"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import test_stage1_e2e as t
from t4libs import REF_FA
with gzip.open(REF_FA, "rb") as f, open("/tmp/fl/ref.fa", "wb") as g: shutil.copyfileobj(f, g)
a, b = t._edge_reads(21)
t._write_fastq("/tmp/fl/e_1.fq", a); t._write_fastq("/tmp/fl/e_2.fq", b)
PY
R=$GRAFT_REPO_ROOT
$R/oracle/_ref/trust4 -t 1 --skipMateExtension -f ref.fa -1 e_1.fq -2 e_2.fq -o ref 2>/dev/null
for i in 1 2 3 4 5 6 7 8; do
  T4_WINDOW=${W:-4} $R/trust4_amd/bin/trust4-hip --skipMateExtension -f ref.fa -1 e_1.fq -2 e_2.fq -o m$i 2>/dev/null
  if cmp -s ref_raw.out m${i}_raw.out; then echo "run $i same"; else echo "run $i DIFF"; continue; echo "contigs ref $(grep -c '>' ref_raw.out) mine $(grep -c '>' m${i}_raw.out); assembled reads ref $(grep -c '>' ref_assembled_reads.fa) mine $(grep -c '>' m${i}_assembled_reads.fa)"; python3 - ref_raw.out m${i}_raw.out <<'PY'
import sys
a=open(sys.argv[1]).read().split("\n"); b=open(sys.argv[2]).read().split("\n")
nd=0
for k,(x,y) in enumerate(zip(a,b)):
    if x!=y:
        nd+=1
        if nd<=3:
            xs,ys=x.split(" "),y.split(" ")
            pos=[i for i,(u,v) in enumerate(zip(xs,ys)) if u!=v][:8]
            print(" line",k,"kind",k%6,"len",len(xs),len(ys),"diffpos",pos,[(xs[i],ys[i]) for i in pos[:4]], a[k-(k%6)][:50])
print(" differing lines",nd)
PY
fi
done
