#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02c14
mkdir -p $OUT
timeout 600 python tools/loopback_headline.py 2 4 8 > $OUT/loopback.jsonl 2> $OUT/loopback.err
cat $OUT/loopback.jsonl; tail -3 $OUT/loopback.err
