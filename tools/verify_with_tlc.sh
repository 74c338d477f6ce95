#!/bin/bash
# Pin this checker against the real engine: run stock TLC on every models/*.cfg twin and compare its verdict and its
# three summary numbers with models/EXPECTED.json (written by tools/make_expected.py from the C oracle).
#
#   tools/verify_with_tlc.sh [-reference DIR] [-jar tla2tools.jar] [-workers N] [cfg ...]
#
# Needs a JVM and tla2tools.jar (TLA2TOOLS_JAR or -jar); neither exists in the build image, where this prints
# "SKIP: no java" and exits 0 — parity with TLC stays unpinned until somebody runs it.
# The reference's .tla files are NOT part of this repository: -reference (default /root/reference, or $KMC_REFERENCE)
# must point at a checkout of hachikuji/kafka-specification; they are symlinked into a scratch directory beside the
# .cfg twins and models/MCAsyncIsr.tla.
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
REF="${KMC_REFERENCE:-/root/reference}"
JAR="${TLA2TOOLS_JAR:-}"
WORKERS="$(nproc 2>/dev/null || echo 4)"
CFGS=()
while [ $# -gt 0 ]; do
  case "$1" in
    -reference) REF="$2"; shift 2;;
    -jar) JAR="$2"; shift 2;;
    -workers) WORKERS="$2"; shift 2;;
    *) CFGS+=("$1"); shift;;
  esac
done
if ! command -v java >/dev/null 2>&1; then
  echo "SKIP: no java on PATH (TLC cannot run here); expectations are in models/EXPECTED.json"
  exit 0
fi
if [ -z "$JAR" ] || [ ! -f "$JAR" ]; then
  echo "SKIP: tla2tools.jar not found (set TLA2TOOLS_JAR or pass -jar)"
  exit 0
fi
if [ ! -f "$REF/KafkaReplication.tla" ]; then
  echo "ERROR: $REF does not hold the reference's .tla files (pass -reference DIR)" >&2
  exit 2
fi
WORK="$(mktemp -d /tmp/kmc_tlc.XXXXXX)"
for f in "$REF"/*.tla "$REPO"/models/*.tla; do ln -sf "$f" "$WORK/$(basename "$f")"; done
cp "$REPO"/models/*.cfg "$WORK"/
[ ${#CFGS[@]} -eq 0 ] && CFGS=($(cd "$REPO/models" && ls *.cfg))
FAIL=0
for cfg in "${CFGS[@]}"; do
  cfg="$(basename "$cfg")"
  mod="$(python3 - "$REPO" "$cfg" <<'PY'
import json, sys
e = json.load(open(sys.argv[1] + "/models/EXPECTED.json")).get(sys.argv[2])
print(e["module"] if e else "")
PY
)"
  if [ -z "$mod" ]; then echo "?? $cfg: no entry in models/EXPECTED.json"; FAIL=1; continue; fi
  for mode in stop exhaustive; do
    extra=""; [ "$mode" = exhaustive ] && extra="-continue"
    log="$WORK/${cfg%.cfg}.$mode.log"
    (cd "$WORK" && java -XX:+UseParallelGC -cp "$JAR" tlc2.TLC -deadlock $extra -workers "$WORKERS" -config "$cfg" "$mod.tla" > "$log" 2>&1)
    python3 "$REPO/tools/tlc_log_diff.py" "$REPO/models/EXPECTED.json" "$cfg" "$mode" "$log" || FAIL=1
  done
done
echo "logs: $WORK"
exit $FAIL
