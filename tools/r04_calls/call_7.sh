#!/bin/bash
# round 4, call 7: traces under orbit counting across shards (logical shards and thread-ranks), then the whole -m gpu suite
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_symmetry.py -q -n 4 -k "trace_across_shards_under_orbit" > $O/t_sym_trace.log 2>&1; tail -15 $O/t_sym_trace.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_native_exchange_threads.py -q -k "with_traces_across" > $O/t_sym_trace_threads.log 2>&1; tail -8 $O/t_sym_trace_threads.log | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q -n 4 > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
