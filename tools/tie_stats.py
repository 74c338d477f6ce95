"""How often do replicas with EQUAL KEYS turn out not to be interchangeable?  (KmcSymm::canon_sorted, kmc_device.h: the
representative of an orbit from four replicas on is chosen among the images whose replica keys ascend; where neighbours'
keys tie, exchanging them is almost always the identity, and only otherwise does the wave walk through all the images.)
CPU only: a prefix of the plain search through the host-compiled device templates (tests/host_emu), every successor's keys
restated on its canonical bytes at three strengths:
    level 0: log, end, hw, epoch       level 1: + what the replica / quorumState / the requests say about ITSELF
    level 2 (the key in use): + how many OTHER replicas hold it in their ISR / name it as leader
usage: python tools/tie_stats.py LEVEL   ->  per configuration: share of successors with any tie, with a tie that is told apart
Round 3: level 2 -> 0 told apart in 1,187,185 successors of Kip279 5/2/2/1 (45.7 % with a tie) and 645,315 of Kip320 3/6/6/2."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_emu
from kafka_specification_amd import CheckerConfig, ModelChecker
from test_symmetry_cpu import permute_bytes


def stats(cfg6, name, model, N, L, R, E, limit, level):
    mc = ModelChecker(CheckerConfig(model=name, n_replicas=N, log_size=L, max_records=R, max_leader_epoch=E, device=-1))
    blk = 5 + L
    g = N * blk

    def keys(b):
        ks = []
        for r in range(N):
            x = b[r * blk:(r + 1) * blk]
            k = [bytes(x[5:]), x[0], x[1], x[2]]
            if level >= 1:
                k += [x[3] == 0, x[3] == r + 1, x[4] >> r & 1, bin(x[4]).count("1"), b[g + 3] == r + 1, b[g + 4] >> r & 1]
                for e in range(E + 1):
                    k += [b[g + 5 + 2 * e] == r + 1, b[g + 6 + 2 * e] >> r & 1]
            if level >= 2:
                k += [sum(1 for o in range(N) if o != r and b[o * blk + 4] >> r & 1),
                      sum(1 for o in range(N) if o != r and b[o * blk + 3] == r + 1)]
            ks.append(tuple(k))
        return ks

    with host_emu.layout(cfg6):
        init = tuple(host_emu.init(cfg6))
        seen, frontier = {init}, [init]
        n = told_apart = any_tie = 0
        while frontier and len(seen) < limit:
            nxt = []
            for s in frontier:
                for _, t in host_emu.successors(cfg6, list(s)):
                    t = tuple(t)
                    n += 1
                    b = mc.unpack(list(t))
                    ks = keys(b)
                    order = sorted(range(N), key=lambda r: ks[r])
                    img = [0] * N
                    for d, r in enumerate(order):
                        img[r] = d
                    bs = permute_bytes(model, N, L, E, b, img)
                    ks2 = [ks[r] for r in order]
                    bad = tie = False
                    for a in range(N - 1):
                        if ks2[a] == ks2[a + 1]:
                            tie = True
                            sw = list(range(N))
                            sw[a], sw[a + 1] = a + 1, a
                            bad = bad or permute_bytes(model, N, L, E, bs, sw) != bs
                    any_tie += tie
                    told_apart += bad
                    if t not in seen:
                        seen.add(t)
                        nxt.append(t)
            frontier = nxt
    print(f"{name} {N}/{L}/{R}/{E} key level {level}: {len(seen)} states, {n} successors, any tie {any_tie / n:.4f}, "
          f"told apart {told_apart / n:.5f}")


if __name__ == "__main__":
    lv = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    stats((4, 5, 2, 2, 1, 0, 0), "Kip279", 4, 5, 2, 2, 1, 150000, lv)
    stats((5, 3, 6, 6, 2, 0, 0), "Kip320", 5, 3, 6, 6, 2, 200000, lv)
