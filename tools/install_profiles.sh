#!/bin/bash
# Copy one tools/collect_profiles.sh collection into profiles/ under this round's names:
#     bash tools/install_profiles.sh gpurun_out/prof_r05 r05
set -e
SRC=$1; R=$2
python3 tools/aggregate_profiles.py $SRC profiles/${R}_final > /dev/null
cp $SRC/bench_wkv6.json profiles/${R}_bench_final.json
for w in bi infctx prefill; do cp $SRC/bench_$w.json profiles/${R}_bench_$w.json; done
[ -s $SRC/bench_dp_lora.json ] && grep "^{" $SRC/bench_dp_lora.json | tail -1 > profiles/${R}_bench_dp_lora_24l.json
for w in bi infctx; do
    f=$(ls -t $SRC/stats_$w/*/*kernel_stats.csv | head -1)
    { head -1 $f; grep "wkv6\|mask_to_lens\|length_order" $f || true; } > profiles/${R}_${w}_kernel_stats.csv
done
ls -la profiles/${R}_*
