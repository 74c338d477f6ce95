#!/bin/bash
# round 6, call 26: what call 25 pointed at - chunks change the random WRITES (stores +30 %, claims +17 %), not the loads, and a
# 128 GiB hipMalloc is as fast as a chunked 8 GiB one: is it how much of the HBM the table's pages lie over?
# (a) can the box show the physical placement (debugfs)?  (b) randbench with the chunks spread over 4 / 16 times the extent
# (RANDBENCH_SPREAD), and one hipMalloc of 16 / 32 / 64 GiB probed over its first 8 GiB vs over all of it; (c) the engine's seen-set
# spread the same way (KMC_SEEN_SET_SPREAD); (d) a run that keeps traces: the predecessor table as one hipMalloc / chunks after the
# seen-set's / chunks alternating with the seen-set's (KMC_PRED_CHUNKS=0 / 1 / 2), five fresh processes each, interleaved.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/r06_calls/call_26.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_26; mkdir -p $O
{ ls /sys/kernel/debug/ 2>&1 | head -5; ls /sys/kernel/debug/dri/ 2>&1 | head; for f in /sys/kernel/debug/dri/*/amdgpu_vram_mm; do echo $f; head -40 $f; done; } > $O/debugfs.txt 2>&1
( cd tools/membench && hipcc --offload-arch=gfx950 -O3 randbench.hip -o randbench 2>/dev/null ) || echo "randbench build failed"
R=tools/membench/randbench
for sp in 1 4 16 1 4 16; do
  echo "## 8 GiB from 8 MiB chunks, spread $sp" | tee -a $O/spread.txt
  RANDBENCH_CHUNK_LOG2=23 RANDBENCH_SPREAD=$sp RANDBENCH_MODES=2,3,4,7 timeout 300 $R 0 2>&1 | tee -a $O/spread.txt
done
for lg in 31 32 33; do
  echo "## one hipMalloc of 2^$lg slots: random accesses over its first 2^30 slots, then over all of it" | tee -a $O/spread.txt
  RANDBENCH_MAX_LOG2=$lg RANDBENCH_MODES=3,4,7 timeout 600 $R 0 30 $lg 2>&1 | tee -a $O/spread.txt
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
for rep in 1 2 3; do for sp in 1 4 16; do
  KMC_SEEN_SET_SPREAD=$sp timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, no traces, spread $sp]" | tee -a $O/engine_spread.txt
done; done
for rep in 1 2 3 4 5; do for pm in 0 1 2; do
  KMC_BENCH_TRACE=1 KMC_PRED_CHUNKS=$pm timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, traces kept, KMC_PRED_CHUNKS=$pm]" | tee -a $O/pred.txt
done
KMC_BENCH_TRACE=1 KMC_PRED_CHUNKS=2 KMC_SEEN_SET_SPREAD=4 timeout 300 $B --steps 5 --warmup 1 2>>$O/err.txt | pick "[headline, traces kept, KMC_PRED_CHUNKS=2 spread 4]" | tee -a $O/pred.txt
done
tail -5 $O/err.txt
