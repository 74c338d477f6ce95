#!/bin/bash
# round 5, call 13b: the deferred probe forced onto the headline and config 4 (3-word states, 6 waves per SIMD)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_13b; mkdir -p $O
export KMC_NO_TORCH=1
for d in "" "-DKMC_DEFER_MIN_WORDS=1"; do
  tag=default; [ -n "$d" ] && tag=deferred
  KMC_JIT_DEFINES=$d timeout 200 python bench.py --no-cpu-baseline --no-cold-start --no-baseline-configs --steps 10 --warmup 2 > $O/head_$tag.json 2> $O/head_$tag.err
  KMC_JIT_DEFINES=$d timeout 200 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline --steps 5 --warmup 1 > $O/c4_$tag.json 2> $O/c4_$tag.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_13b/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    oc = j.get("orbit_counting") or {}
    print(os.path.basename(f), "ms/step %.2f kernel %.2f golden %s" % (j["ms_per_step"], 1e3 * j["roofline"]["kernel_seconds_per_step"], j["config"]["matches_oracle_golden"]),
          ("| sym kernel %.2f same %s" % (1e3 * oc["kernel_seconds_per_step"], oc["every_count_equals_the_plain_run"])) if oc else "")
PY
