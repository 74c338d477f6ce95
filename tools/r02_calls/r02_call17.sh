#!/bin/bash
# (a) A/B of nt probe / nt frontier on the headline; (b) true DRAM bytes of randbench's write / atomic modes; (c) kernel trace of
# the P = 2 loopback headline after the send-filter counter fix
cd "$(dirname "$0")/.."
REPO=$PWD
OUT=$REPO/gpurun_out/r02c17
mkdir -p $OUT
rm -f gpurun_out/sweep.log
bash tools/sweep.sh "base||" "nt_probe|-DKMC_NT_PROBE=1|" "nt_frontier|-DKMC_NT_FRONTIER=1|" "nt_both|-DKMC_NT_PROBE=1 -DKMC_NT_FRONTIER=1|" "base_again||" "nt_probe_again|-DKMC_NT_PROBE=1|"
cp gpurun_out/sweep.log $OUT/sweep.log
cd /tmp && export TMPDIR=/tmp
BIN=$REPO/tools/membench/randbench
i=0
for ctr in "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  timeout 60 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/rb$i -o pmc -- $BIN 0 > $OUT/rb$i.log 2>&1
  echo "randbench pass $i ($ctr) rc=$?" >> $OUT/passes.log
done
cat $OUT/passes.log
python3 $REPO/tools/calibrate_fetch.py $OUT | tee $OUT/randbench_dram_bytes.txt
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/tools/loopback_headline.py 2 > $OUT/loopback_trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cut -c1-160 $f | head -8
grep shards $OUT/loopback_trace.log
