#!/bin/bash
# round 4, call 6: the whole -m gpu suite once more (with the sharded 128-bit collision test), and which code objects the box
# had to specialise itself (they belong in configs.precompile_list / precompile_variants).
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04_6; mkdir -p $O
ls kafka_specification_amd/kmc_cache > $O/cache_before.txt
timeout 2400 python -m pytest tests -m gpu -q -n 4 > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
ls kafka_specification_amd/kmc_cache > $O/cache_after.txt
diff $O/cache_before.txt $O/cache_after.txt | grep '^>' > $O/jit_compiled_on_the_box.txt; cat $O/jit_compiled_on_the_box.txt
