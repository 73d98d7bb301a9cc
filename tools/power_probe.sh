#!/bin/bash
# Sample GPU clock and package power while a command runs (GPU box):
#   bash tools/power_probe.sh <seconds before first sample> <samples> <command...>
DELAY=$1; N=$2; shift 2
"$@" > gpurun_out/power_probe_cmd.log 2>&1 &
BP=$!
sleep $DELAY
for i in $(seq $N); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/GPU\[0\]\t\t: //; s/Current Socket Graphics Package Power (W)/W/' | tr '\n' ' '; echo
    sleep 0.5
done
wait $BP
grep -v amdgpu.ids gpurun_out/power_probe_cmd.log | tail -3
