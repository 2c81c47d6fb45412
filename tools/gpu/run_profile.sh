cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=r03 bash tools/profile_round.sh gpurun_out/r03prof
cp profiles/r03_final_pmc_traffic_per_launch.jsonl gpurun_out/r03prof/r03_final_pmc_traffic_per_launch.jsonl
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03prof/bench_fp32_ops.json 2> gpurun_out/r03prof/bench_fp32_ops.err
