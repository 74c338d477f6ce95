#!/bin/bash
# sharded headline on one GPU (P logical shards, exchange under the C ABI): with and without the sender-side duplicate filter
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c18
mkdir -p $OUT
export KMC_LOOPBACK_EXCHANGES=rccl
for nf in 0 1; do
  KMC_NO_SEND_FILTER=$nf timeout 200 python tools/loopback_headline.py 2 4 8 >> $OUT/loopback_filter.jsonl 2>> $OUT/err.txt
done
cat $OUT/loopback_filter.jsonl; tail -n 3 $OUT/err.txt
