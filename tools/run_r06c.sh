mkdir -p gpurun_out/r06c
export RWKV_AMD_NO_SELFTEST=1
echo "== full-stage experiment (past-the-end selects compiled out) vs head, alternating" > gpurun_out/r06c/full_ab.txt
for i in 1 2 3; do for v in head full; do ABL_NAME=$v RWKV_AMD_LIB=build_ab/$v/lib.so WKV6_CLOCKS=1 python tools/time_ops.py --iters 60 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06c/full_ab.txt; done; done
unset RWKV_AMD_NO_SELFTEST
python tools/time_bi.py > gpurun_out/r06c/time_bi.txt 2>&1
bash tools/collect_profiles.sh gpurun_out/r06c/prof > gpurun_out/r06c/collect.log 2>&1
tail -3 gpurun_out/r06c/collect.log
