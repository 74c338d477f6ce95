#!/bin/bash
# round 5, call 2b: full leaves below seven brokers (forced: -DKMC_FULL_LEAVES_MIN_INSTANCES=0) on the headline and on BASELINE
# config 4, plain and orbit counting; the cold start's breakdown (tlc -v)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r05_2b; mkdir -p $O
export KMC_NO_TORCH=1
for d in "" "-DKMC_FULL_LEAVES_MIN_INSTANCES=0"; do
  tag=default; [ -n "$d" ] && tag=full_leaves
  KMC_JIT_DEFINES=$d timeout 200 python bench.py --no-cpu-baseline --no-cold-start --no-baseline-configs --steps 10 --warmup 2 > $O/head_$tag.json 2> $O/head_$tag.err
  KMC_JIT_DEFINES=$d timeout 200 python bench.py --workload Kip279,5,2,2,1 --no-cpu-baseline --steps 5 --warmup 1 > $O/c4_$tag.json 2> $O/c4_$tag.err
  KMC_JIT_DEFINES=$d timeout 200 python bench.py --workload Kip279,5,2,2,1 --symmetry --no-cpu-baseline --steps 5 --warmup 1 > $O/c4_sym_$tag.json 2> $O/c4_sym_$tag.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_2b/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r, c = j.get("roofline", {}), j.get("config", {})
    oc = j.get("orbit_counting") or {}
    print(os.path.basename(f), "ms/step %.2f" % j["ms_per_step"], "kernel ms %.2f" % (1e3 * r.get("kernel_seconds_per_step", 0)),
          "golden", c.get("matches_oracle_golden"), "distinct", c.get("distinct_states"), "generated", c.get("states_generated"),
          ("| sym ms/step %.2f kernel %.2f same %s" % (oc["ms_per_step"], 1e3 * oc["kernel_seconds_per_step"], oc["every_count_equals_the_plain_run"])) if oc else "")
PY
for extra in "" "-notrace"; do
  for i in 1 2; do
    s=$(date +%s.%N)
    kafka_specification_amd/tlc models/Kip320.tla -table $((1<<30)) -frontier $((1<<26)) -v $extra > $O/tlc$extra.$i.log 2>&1
    e=$(date +%s.%N)
    echo "tlc $extra run $i: wall $(echo "$e - $s" | bc) s"; grep -E "Wall time|distinct states found" $O/tlc$extra.$i.log | tail -2
  done
done
