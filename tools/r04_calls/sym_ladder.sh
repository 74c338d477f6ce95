cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r04_13; O=gpurun_out/r04_13/sym_ladder.jsonl; : > $O
export KMC_NO_TORCH=1
for w in "Kip320 3 6 6 2 3 30" "Kip279 5 2 2 1 3 29" "Kip320 5 1 1 1 3 26" "KafkaTruncateToHighWatermark 6 1 1 1 3 28" "KafkaTruncateToHighWatermark 3 6 6 2 3 31" "Kip101 3 6 6 2 3 31"; do
  timeout 300 python tools/sym_ab.py $w >> $O 2>&1
done
KMC_AB_FP128=1 timeout 400 python tools/sym_ab.py Kip320 3 6 6 3 2 33 >> $O 2>&1
tail -30 $O | cut -c1-220
