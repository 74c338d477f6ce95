#!/bin/bash
# round 6, call 32: does a box's first search run slower than its later ones (calls 29 / 30: the default bench line read 31.7 ms of
# k_expand and a 1.35 ms clear as the first long process, 28.4 / 1.22 in twelve processes afterwards)?  One handle, 200 searches back
# to back as the box's FIRST GPU work, k_expand and clear per search; 40 s idle; 100 more; the clocks rocm-smi shows before / after.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/r06_calls/call_32.sh'
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06_32; mkdir -p $O
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "clk|Power|Temp" | head -12 > $O/smi_before.txt
KMC_NO_TORCH=1 python - <<'PY' 2>&1 | tee $O/series.txt
import time, subprocess
import kafka_specification_amd as kmc
c = dict(model="Kip320", n_replicas=3, log_size=6, max_records=6, max_leader_epoch=2, invariants=("TypeOk", "WeakIsr", "StrongIsr"))
cfg = kmc.CheckerConfig(**c, device=0, table_capacity=1 << 30, frontier_capacity=1 << 26)
t0 = time.time()
def smi():
    out = subprocess.run("rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|mclk|fclk|Power|Temp' | tr -s ' ' | tr '\n' ';'", shell=True, capture_output=True, text=True).stdout
    return out[:600]
with kmc.ModelChecker(cfg) as mc:
    for phase, n in (("first work on the box", 200), ("after 40 s idle", 100)):
        xs = []
        for k in range(n):
            r = mc.run()
            xs.append((time.time() - t0, 1e3 * r.seconds_expand, 1e3 * r.seconds_clear))
            assert r.distinct == 279753922
        print(phase)
        for i in list(range(0, 10)) + list(range(10, n, 10)):
            print("  search %3d at %6.2f s: k_expand %.2f ms, clear %.2f ms" % (i, *xs[i]))
        print("  ", smi())
        time.sleep(40)
PY
