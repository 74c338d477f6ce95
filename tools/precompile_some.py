"""Specialise a hand-picked set of code objects into the in-tree cache (no GPU needed), in parallel: what an A/B call to the GPU
box needs, without waiting for build()'s whole list.

    python tools/precompile_some.py 'Kip320,3,6,6,2' 'Kip320,3,6,6,2,sym' 'Kip320,7,8,8,3' --modes 0 --defines=-DKMC_FULL_LEAVES_MIN_INSTANCES=1000000

A job = model,N,L,R,E[,sym]; --modes: which of a configuration's three code objects (0 search, 1 level-step, 2 enumerator);
--defines: KMC_JIT_DEFINES of the build (part of the cache key, as at run time)."""
import argparse
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(job):
    spec, mode, defines, verify = job
    if defines:
        os.environ["KMC_JIT_DEFINES"] = defines
    else:
        os.environ.pop("KMC_JIT_DEFINES", None)
    if verify:
        os.environ["KMC_VERIFY"] = "1"
    import kafka_specification_amd as kmc
    f = spec.split(",")
    sym = len(f) > 5 and f[5] == "sym"
    c = kmc.CheckerConfig(model=f[0], n_replicas=int(f[1]), log_size=int(f[2]), max_records=int(f[3]), max_leader_epoch=int(f[4]),
                          symmetry=sym)
    t = time.time()
    kmc.precompile(c, "gfx950", mode)
    return f"{spec} mode {mode} [{defines}] {time.time() - t:.1f} s"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("specs", nargs="+")
    ap.add_argument("--modes", default="0,1,2")
    ap.add_argument("--defines", action="append", default=None, help="one build per occurrence ('' = the default build)")
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    jobs = [(s, int(m), d, a.verify) for d in (a.defines or [""]) for s in a.specs for m in a.modes.split(",")]
    with ProcessPoolExecutor(max_workers=min(a.jobs, len(jobs))) as ex:
        for line in ex.map(one, jobs):
            print(line, flush=True)


if __name__ == "__main__":
    main()
