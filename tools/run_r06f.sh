# end-of-round evidence run (GPU box): bash tools/run_r06f.sh
mkdir -p gpurun_out/r06f
python tools/time_bi.py > gpurun_out/r06f/time_bi.txt 2>&1
RWKV_AMD_LIB=build_ab/head/lib.so python tools/time_bi.py > gpurun_out/r06f/time_bi_head.txt 2>&1
bash tools/collect_profiles.sh gpurun_out/r06f/prof > gpurun_out/r06f/collect.log 2>&1
python tools/determinism_check.py > gpurun_out/r06f/determinism.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06f/bench_driver_style.json 2> gpurun_out/r06f/bench_driver_style.err
tail -2 gpurun_out/r06f/collect.log; tail -3 gpurun_out/r06f/determinism.txt
