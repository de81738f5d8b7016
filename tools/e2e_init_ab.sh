mkdir -p /tmp/e && cd /tmp/e && zcat $GRAFT_REPO_ROOT/data/hg38_bcrtcr.fa.gz > ref.fa && $GRAFT_REPO_ROOT/tools/t4synth ref.fa 100000 0 4 c5 --cells 1000 > /dev/null
A="-t 8 -f ref.fa -1 c5_1.fq -2 c5_2.fq --barcode c5_bc.fa --UMI c5_umi.fa"
for k in 1 2; do
for M in async sync; do
  if [ $M = sync ]; then export T4_SYNC_INIT=1; else unset T4_SYNC_INIT; fi
  S=$(date +%s.%N); $GRAFT_REPO_ROOT/trust4_amd/bin/trust4-hip $A -o out_$M 2> log_$M; E=$(date +%s.%N)
  echo "$M: $(echo "$E - $S" | bc) s; $(grep 'Finish assembly' log_$M | sed 's/.*(//')"
done; done
cmp out_async_raw.out out_sync_raw.out && echo same
