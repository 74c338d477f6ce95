#!/bin/bash
# PC sampling of the headline bench (beta feature of rocprofv3).  usage: tools/pcsample.sh <tag> [method]
TAG=${1:-pcs}; METHOD=${2:-stochastic}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/pcs_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp KMC_NO_TORCH=1 ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
UNIT=cycles; INT=1048576
if [ "$METHOD" = "host_trap" ]; then UNIT=time; INT=100; fi
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INT \
   --kernel-trace --output-format csv -d "$OUT" -o pcs -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/log.txt" 2>&1
echo "rc=$?"; tail -3 "$OUT/log.txt" | cut -c1-300; ls -la "$OUT" | head; 
