#!/bin/bash
# round 3, final call 3: from how many replicas on should the representative be chosen among the sorted images?
# (KMC_SYMM_UNROLLED_MAX as a JIT define: the headline with 3 replicas sorted instead of 6 unrolled images, two
# four-replica configurations sorted instead of 24 unrolled images)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/f3; mkdir -p $O
export KMC_NO_TORCH=1
out=$O/threshold.log; : > $out
echo "== headline, unrolled (default)" >> $out; timeout 200 python tools/sym_headline.py 4 sym 2>&1 | grep ms_step | tail -3 >> $out
echo "== headline, sorted (KMC_SYMM_UNROLLED_MAX=2)" >> $out; KMC_JIT_DEFINES=-DKMC_SYMM_UNROLLED_MAX=2 timeout 200 python tools/sym_headline.py 4 sym 2>&1 | grep ms_step | tail -3 >> $out
for w in "Kip320 4 2 2 1" "Kip101 4 2 1 2"; do
  echo "== $w, unrolled (default)" >> $out; timeout 200 python tools/sym_ab.py $w 4 26 2>&1 | tail -7 >> $out
  echo "== $w, sorted (KMC_SYMM_UNROLLED_MAX=3)" >> $out; KMC_JIT_DEFINES=-DKMC_SYMM_UNROLLED_MAX=3 timeout 200 python tools/sym_ab.py $w 4 26 2>&1 | tail -7 >> $out
done
cat $out
