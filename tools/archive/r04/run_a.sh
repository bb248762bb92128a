#!/bin/bash
# round 4, first GPU call: LDS-DMA semantics / rate micro-benchmarks + the round-3 tree's baseline on this box
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 300 tools/ubench/_bin/lds_dma abc > $O/r04_a_lds_dma.txt 2>&1
cat $O/r04_a_lds_dma.txt
timeout 400 python bench.py --no-cpu-baseline > $O/r04_a_bench.json 2> $O/r04_a_bench.err
tail -c 600 $O/r04_a_bench.json
timeout 300 python tools/conv_bench.py bf16 10 > $O/r04_a_conv_bench.txt 2>&1
cat $O/r04_a_conv_bench.txt
