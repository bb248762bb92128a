#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in acdc/unet++_3d.yaml amos_ct/attention_unet_3d.yaml acdc/vnet_3d.yaml acdc/medformer_3d.yaml; do
t=$(echo $c | tr '/+' '__' | sed 's/_3d.yaml//')
rm -rf /tmp/pf_x
rocprofv3 --kernel-trace --stats -d /tmp/pf_x -o p -- python $R/tools/bench_shipped_config.py $c --steps 5 --warmup 2 > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pf_x/p_results.db 7 > $O/r06_f3_${t}_kernels.txt 2>&1
echo "== $c"; head -16 $O/r06_f3_${t}_kernels.txt | cut -c1-150
done
