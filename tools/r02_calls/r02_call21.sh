#!/bin/bash
# sharded k_expand with one atomic round trip per batch: sharded / trace / AsyncIsr loopback tests, then the loopback headline
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c21
mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_sharded_and_traces.py tests/test_gpu_async_isr.py -m gpu -x -q -k "shard or loopback or trace or filter or checkpoint or config4" > $OUT/tests.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/tests.txt
grep -n "passed\|failed\|rror" $OUT/tests.txt | tail -5
export KMC_LOOPBACK_EXCHANGES=rccl
timeout 200 python tools/loopback_headline.py 2 4 8 > $OUT/loopback.jsonl 2> $OUT/err.txt
KMC_SEND_FILTER=1 timeout 100 python tools/loopback_headline.py 8 >> $OUT/loopback.jsonl 2>> $OUT/err.txt
cat $OUT/loopback.jsonl
