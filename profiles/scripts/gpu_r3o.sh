#!/bin/bash
# round 3, session o: where a slow read's time goes (phases by overlap count), deferral threshold of the extensions, records per block of extendKernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3o; mkdir -p $O
W=/tmp/w3o; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
run() {  # name, env...
  local name=$1; shift
  ( time env T4_TIMING=1 "$@" timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  md5sum $W/m_${name}_raw.out >> $O/log_$name.txt
  echo "== $name: $(grep -h 'real' $O/log_$name.txt | tr '\n' ' ') $(grep -o 'first launch to sync [0-9.]*' $O/log_$name.txt) $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads in [0-9.]* s' $O/log_$name.txt) $(tail -1 $O/log_$name.txt | cut -c1-12)"
}
run base T4_ROUND_LOG=$O/rounds_base.txt
run base2
run phases T4_PHASE_DUMP=1 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/phases
for d in 32 16 8; do run defer$d T4_AQ_EXTEND_DEFER=$d; done
run nrec4 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/nrec4
run nrec4_defer16 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/nrec4 T4_AQ_EXTEND_DEFER=16 T4_ROUND_LOG=$O/rounds_nrec4_defer16.txt
run nrec4_defer8 LD_LIBRARY_PATH=$PWD/trust4_amd/variants/nrec4 T4_AQ_EXTEND_DEFER=8
grep -A12 "AddRead queries with" $O/log_phases.txt | head -80
gzip -f $O/rounds_*.txt
timeout 200 python tools/gpu_pass.py 2000000 4 > $O/annotate_pass.txt 2>&1; tail -4 $O/annotate_pass.txt
