#!/bin/bash
# round 6, call 25: (a) the chip's random-access ceilings re-measured in memory mapped from 8 MiB chunks (what the seen-set lies in
# since call 23) against one hipMalloc, same box, interleaved: tools/membench/randbench, every mode at 8 GiB, the stretch's regime
# (128 GiB, modes 1 / 7 / 13) once each way; (b) the predecessor table of a run that keeps traces (the CLI's default) through the
# same allocator against hipMalloc (KMC_PRED_CHUNKS=0), the headline with KMC_BENCH_TRACE=1, fresh processes, interleaved.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r06_calls/call_25.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_25; mkdir -p $O
( cd tools/membench && hipcc --offload-arch=gfx950 -O3 randbench.hip -o randbench 2>/dev/null ) || echo "randbench build failed"
R=tools/membench/randbench
for rep in 1 2; do
  echo "## rep $rep: one hipMalloc" | tee -a $O/randbench.txt
  timeout 300 $R 0 2>&1 | tee -a $O/randbench.txt
  echo "## rep $rep: 8 MiB chunks" | tee -a $O/randbench.txt
  RANDBENCH_CHUNK_LOG2=23 timeout 300 $R 0 2>&1 | tee -a $O/randbench.txt
done
echo "## 2 MiB chunks" | tee -a $O/randbench.txt
RANDBENCH_CHUNK_LOG2=21 RANDBENCH_MODES=1,3,7 timeout 300 $R 0 2>&1 | tee -a $O/randbench.txt
for lg in 0 23; do
  echo "## 2^34 slots (128 GiB), chunk log2 $lg" | tee -a $O/randbench_128g.txt
  RANDBENCH_CHUNK_LOG2=$lg RANDBENCH_MAX_LOG2=34 RANDBENCH_MODES=1,7,13 timeout 600 $R 0 34 2>&1 | tee -a $O/randbench_128g.txt
done
export KMC_NO_TORCH=1
B="python bench.py --no-cpu-baseline --no-orbit-counting --no-cold-start --no-baseline-configs --no-stretch"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); c = j['config']; b = c.get('step_breakdown') or {}
        print('$1', 'ms/step %.2f' % j['ms_per_step'], 'k_expand %.2f k_inv %.2f clear %.2f' % (b.get('k_expand_ms', 0), b.get('k_inv_ms', 0), b.get('clear_seen_set_ms', 0)), 'golden', c['matches_oracle_golden'])
"; }
for rep in 1 2 3; do
  KMC_BENCH_TRACE=1 KMC_PRED_CHUNKS=0 timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, traces kept, pred = one hipMalloc]" | tee -a $O/pred.txt
  KMC_BENCH_TRACE=1 timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, traces kept, pred from chunks]" | tee -a $O/pred.txt
  timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, no traces]" | tee -a $O/pred.txt
done
tail -5 $O/err.txt
