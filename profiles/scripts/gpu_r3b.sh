#!/bin/bash
# round 3, session b: asynchronous query lanes -- 100 k pairs under several lane / batch / look-ahead settings (md5 of _raw.out each), then C2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b; mkdir -p $O
W=/tmp/w3b; mkdir -p $W; zcat data/hg38_bcrtcr.fa.gz > $W/ref.fa
tools/t4synth $W/ref.fa 100000 2000 1 $W/b > /dev/null
run() {  # name, env...
  local name=$1; shift
  ( time env T4_TIMING=1 "$@" timeout 300 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m_$name ) > $O/log_$name.txt 2>&1
  echo "== $name $* : $(grep real $O/log_$name.txt) $(md5sum < $W/m_${name}_raw.out | cut -c1-12) $(grep -o 'GPU query rounds [0-9]* with [0-9]* reads in [0-9.]* s' $O/log_$name.txt) | $(grep -o 'waits for the head.*' $O/log_$name.txt)" | tee -a $O/summary.txt
}
run warm T4_NOP=1
run default T4_NOP=1
run sync T4_LIVE_SYNC=1
run lanes1 T4_LIVE_LANES=1
run lanes2 T4_LIVE_LANES=2
run lanes4 T4_LIVE_LANES=4
run mb1 T4_LIVE_MIN_BATCH=1
run mb2 T4_LIVE_MIN_BATCH=2
run mb8 T4_LIVE_MIN_BATCH=8
run ah8 T4_QUERY_AHEAD=8
run ah12mb2 T4_QUERY_AHEAD=12 T4_LIVE_MIN_BATCH=2
run ah48 T4_QUERY_AHEAD=48
run t16 T4_NOP=1
T4_ROUND_LOG=$O/rounds_default.txt T4_TIMING=1 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/b_1.fq -2 $W/b_2.fq -o $W/m_rl > $O/log_roundlog.txt 2>&1
tools/t4synth $W/ref.fa 1000000 20000 1 $W/c2 > /dev/null
( time T4_TIMING=1 T4_STATS_JSON=$O/stats_c2.json timeout 900 trust4_amd/bin/trust4-hip -t 8 --skipMateExtension -f $W/ref.fa -1 $W/c2_1.fq -2 $W/c2_2.fq -o $W/mc2 ) > $O/log_c2.txt 2>&1
md5sum $W/mc2_raw.out $W/mc2_assembled_reads.fa $W/mc2_final.out > $O/c2_md5.txt
tail -7 $O/log_c2.txt; cat $O/c2_md5.txt
