#!/bin/sh
# Regenerates the large-configuration golden fixtures with the C oracle (several minutes and
# ~30 GB of RAM for 3/6/6/2).  The reference repository holds no golden vectors at all; these
# pin the GPU engine at the headline size to the exact-state CPU oracle.
set -e
cd "$(dirname "$0")/../.."
make -s -C oracle
./oracle/kmc_oracle --model Kip320 --N 3 --L 5 --R 5 --E 2 --threads 8 --inv 7 > tests/golden/oracle_kip320_3_5_5_2.json
./oracle/kmc_oracle --model Kip320 --N 3 --L 6 --R 6 --E 2 --threads 8 --inv 7 > tests/golden/oracle_kip320_3_6_6_2.json
./oracle/kmc_oracle --model Kip279 --N 5 --L 2 --R 2 --E 1 --threads 8 --inv 1 > tests/golden/oracle_kip279_5_2_2_1.json   # BASELINE config 4 (TypeOk: exhaustive)
# SURVEY §8d names two bindings for the headline: Kip320, and KafkaTruncateToHighWatermark with TypeOk only (its Next is built
# purely from KafkaReplication.tla actions).  At LogSize 6 the latter outgrows the oracle's RAM; LogSize 5 (221 M states) is pinned.
./oracle/kmc_oracle --model KafkaTruncateToHighWatermark --N 3 --L 5 --R 5 --E 2 --threads 8 --inv 1 > tests/golden/oracle_thw_3_5_5_2.json
# ... and at the headline's LogSize 6 (810,380,080 states) in the oracle's fingerprint-only mode: 17 GB and ~10 minutes on 8
# cores where the exact mode would need ~90 GB.  Not exact (n^2/2^65 = 0.02 expected collisions), but a hash, an encoding, a
# table and a BFS that share nothing with the GPU's.
./oracle/kmc_oracle --model KafkaTruncateToHighWatermark --N 3 --L 6 --R 6 --E 2 --threads 8 --inv 1 --fp-only --table-log2 31 > tests/golden/oracle_fp_thw_3_6_6_2.json
# the three other Kafka models at the same near-headline constants (TypeOk only: they violate StrongIsr by design), 160-177 M states
for m in Kip101 Kip279 Kip320FirstTry; do
  ./oracle/kmc_oracle --model $m --N 3 --L 5 --R 5 --E 2 --threads 8 --inv 1 > tests/golden/oracle_$(echo $m | tr A-Z a-z)_3_5_5_2.json
done
# ... and at the headline's own constants, 607-655 M states each, in the fingerprint-only mode (2-3 minutes each on 8 cores)
for m in Kip101 Kip279 Kip320FirstTry; do
  ./oracle/kmc_oracle --model $m --N 3 --L 6 --R 6 --E 2 --threads 8 --inv 1 --fp-only --table-log2 31 > tests/golden/oracle_fp_$(echo $m | tr A-Z a-z)_3_6_6_2.json
done
# ... and EXACTLY (round 4): Oracle-O stores one full state per orbit of the permutations of Replicas (a sixth of the states at
# three replicas: 101-135 M stored states, ~11 GB, 3-5 minutes on 4 threads) and weighs every count, so the four models whose plain
# exact search outgrows this box are pinned without a hash in sight; the fp-only files above stay as a third witness.  Kip320 itself
# too (46,636,681 stored states = what the GPU's orbit-counting search stores).  --inv 7: TypeOk, WeakIsr, StrongIsr in continue mode.
for m in KafkaTruncateToHighWatermark:thw Kip101:kip101 Kip279:kip279 Kip320FirstTry:kip320firsttry Kip320:kip320; do
  ./oracle/orbit_oracle --model ${m%%:*} --N 3 --L 6 --R 6 --E 2 --threads 4 --inv 7 --table-log2 29 --max-stored 170000000 > tests/golden/orbit_${m##*:}_3_6_6_2.json
done
# ... and one step beyond the headline: Kip320 3/7/7/2, 973,929,178 states from 162,341,877 stored ones (~6 minutes, ~17 GB)
./oracle/orbit_oracle --model Kip320 --N 3 --L 7 --R 7 --E 2 --threads 4 --inv 7 --table-log2 30 --max-stored 450000000 > tests/golden/orbit_kip320_3_7_7_2.json
# ... and the stretch configuration, Kip320 3/6/6/3: 6,452,700,520 states from 1,075,491,542 stored ones — EXACT, with the arena
# bit-packed (--compact: 22 bytes per stored state instead of 48, a table of 32-bit indices: 23.7 + 8.6 GB; 20 minutes on 6 threads)
./oracle/orbit_oracle --model Kip320 --N 3 --L 6 --R 6 --E 3 --threads 6 --inv 7 --compact --table-log2 31 --max-stored 1100000000 > tests/golden/orbit_kip320_3_6_6_3.json
# BASELINE config 5 (7 brokers, LogSize 8) cannot be exhausted: the exact oracle's PREFIX of ten levels (197,561,008 states,
# 2.5 minutes, ~25 GB) — what the plain and the orbit-counting GPU searches are held to over a level budget of 10
# (written through tests/kmo.py: kmo.Run(make_config("Kip320", N=7, L=8, R=8, E=3, threads=8, max_states=41002348)) -> levels, generated,
#  action_generated; the stand-alone binary's equivalent:)
# ./oracle/kmc_oracle --model Kip320 --N 7 --L 8 --R 8 --E 3 --threads 8 --inv 1 --max-states 41002348 > tests/golden/oracle_kip320_7_8_8_3_levels10.json
# ... and beyond what the plain search fits anywhere: the orbit-counting oracle (oracle/orbit_oracle.c), 14 levels in one minute
# (50,390,682,994 states from 18.9 M stored ones), 17 levels with the last level kept as fingerprints only (~40 minutes, ~35 GB)
./oracle/orbit_oracle --model Kip320 --N 7 --L 8 --R 8 --E 3 --levels 14 --threads 8 --table-log2 26 > tests/golden/orbit_kip320_7_8_8_3_levels14.json
./oracle/orbit_oracle --model Kip320 --N 7 --L 8 --R 8 --E 3 --levels 17 --last-level-fp --threads 8 --table-log2 29 --max-stored 230000000 --fp-table-log2 30 > tests/golden/orbit_kip320_7_8_8_3_levels17.json
# The reference's own text, executed (Oracle-R, oracle/tlar; needs /root/reference): two and three replicas (25 CPU-minutes) ...
# python tests/golden/make_oracle_r_golden.py            # -> tests/golden/oracle_r_ladder.json
# ... and four to eight (two CPU-hours, one of them the whole of Kip320 at seven replicas on one core; incremental: entries already
# in the file are kept)
# python tests/golden/make_oracle_r_golden.py --wide     # -> tests/golden/oracle_r_wide.json
# BASELINE config 4 at SURVEY 8(a.0)'s own sizing (Kip279 5/4/4/3: logs four deep, four epochs; round 6): not exhaustible — the exact
# oracle's PREFIX of twelve levels (318,475,476 states, 2.4 minutes on 7 threads, ~24 GB).  --max-states = the first eleven levels + 1,
# so the search stops after the twelfth (the eleven-level sum comes from a first run with --max-states 38031472: 112,310,901)
./oracle/kmc_oracle --model Kip279 --N 5 --L 4 --R 4 --E 3 --threads 7 --inv 1 --max-states 112310902 > tests/golden/oracle_kip279_5_4_4_3_levels12.json
