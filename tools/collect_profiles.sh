#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to.  Run on the GPU box from the repo root:
#     bash tools/collect_profiles.sh gpurun_out/prof
# then, back in the container:  python tools/aggregate_profiles.py gpurun_out/prof profiles/r06_final
# (kernel-trace/stats and every --pmc group are separate runs; no sys/hip trace is combined with --pmc).
set -e -o pipefail
OUT=${1:-gpurun_out/prof}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 50 --warmup 10 --no-cpu --traffic none"   # no nested profiler under the profiler
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -- $CMD > "$ROOT/$OUT/stats.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$ROOT/$OUT/pmc$i" -- $CMD > "$ROOT/$OUT/pmc$i.log" 2>&1 || echo "pmc group $i failed: $grp"
done
# kernel-trace statistics and bench lines of the other single-GPU workloads (BASELINE configs[2], [4])
for wl in bi infctx; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_$wl" -- python3 $ROOT/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu > "$ROOT/$OUT/stats_$wl.log" 2>&1 || echo "stats $wl failed"
    python3 $ROOT/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu > "$ROOT/$OUT/bench_$wl.json" 2> /dev/null
done
python3 $ROOT/bench.py --workload prefill --steps 20 --warmup 5 --no-cpu > "$ROOT/$OUT/bench_prefill.json" 2> /dev/null
python3 $ROOT/bench.py --workload dp_lora --steps 10 --warmup 3 > "$ROOT/$OUT/bench_dp_lora.json" 2> "$ROOT/$OUT/bench_dp_lora.err" || echo "dp_lora failed"   # configs[3], 24 layers, one GPU (DDP over a one-rank RCCL group)
python3 $ROOT/bench.py --steps 100 --warmup 20 > "$ROOT/$OUT/bench_wkv6.json" 2> "$ROOT/$OUT/bench_wkv6.err"           # traffic measured live
echo done
