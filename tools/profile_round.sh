#!/bin/bash
# Collects every artefact profiles/README.md lists for one build, on the GPU box:  tools/profile_round.sh <out dir>
# (kernel trace + stats, FETCH_SIZE / WRITE_SIZE / MFMA-busy / SQ counter passes -- each --pmc pass on its own, without
# tracing domains -- and the bench lines).  Copy the summaries from <out dir> into profiles/ afterwards.
set -u
O=${1:-gpurun_out/profile}
mkdir -p "$O"
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-ops --no-roofline --no-reference-loop"
rocprofv3 --kernel-trace --stats -d "$O/trace" -o run --output-format csv -- $B --steps 20 --warmup 5 > "$O/trace.log" 2>&1
cp "$O"/trace/run_kernel_stats.csv "$O/kernel_stats.csv" 2>/dev/null || cp "$O"/trace/*/run_kernel_stats.csv "$O/kernel_stats.csv"
python tools/prof_summary.py "$O/trace" 25 60 > "$O/kernel_summary.txt" 2>&1 || python - "$O" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows); steps = 25
out = ["kernel time total %.2f ms, per step %.2f ms over %d steps; %d distinct kernels, %d launches/step" % (
    tot / 1e6, tot / 1e6 / steps, steps, len(rows), sum(int(r['Calls']) for r in rows) / steps)]
for r in rows[:60]:
    out.append("%-92s n/step=%6.1f ms/step=%7.3f avg_us=%8.1f %5.1f%%" % (r['Name'][:92], int(r['Calls']) / steps,
               float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
open(sys.argv[1] + "/kernel_summary.txt", "w").write("\n".join(out) + "\n")
PY
python tools/trace_gaps.py "$O/trace" > "$O/trace_gaps.txt" 2>&1
python tools/trace_grids.py "$O/trace" 25 > "$O/trace_grids.txt" 2>&1
python tools/trace_step.py "$O/trace" 12 > "$O/step_sequence.txt" 2>&1
# one S2 Block's backward (192 channels): the launches around a mid-run attention_bwd_tile_kernel<24, 192, 8> (4 before: the tail's
# BatchNorm + fc3; 9 behind: gv ... the fc1 input gradient)
python tools/trace_block.py "$O/trace" "attention_bwd_tile_(mixed_)?kernel<24, 192" 60 4 9 same > "$O/blockS2.txt" 2>&1
# one S2 Block's forward: around a mid-run attention_fwd_tile_kernel<24, 192, 12> (7 before: fc1 ... the logit sums; 3 behind)
python tools/trace_block.py "$O/trace" "attention_fwd_tile_kernel<24, 192, 12" 60 7 4 same > "$O/blockS2_fwd.txt" 2>&1
# one S3 Block's forward (384 channels, ~1 074 points): the launches around a mid-run logits_fwd_mfma_kernel<48, 384, 4>
python tools/trace_block.py "$O/trace" "logits_fwd_mfma_kernel<48, 384, 4" 30 5 8 same > "$O/blockS3_fwd.txt" 2>&1
rm -rf "$O/trace"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_f" -- $B --steps 5 --warmup 2 > "$O/pmc_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_w" -- $B --steps 5 --warmup 2 > "$O/pmc_w.log" 2>&1
python tools/pmc_traffic.py "$O/pmc_f" "$O/pmc_w" > "$O/pmc_traffic_per_launch.jsonl" 2> "$O/pmc_traffic.err"
rm -rf "$O/pmc_f" "$O/pmc_w"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_m" -- $B --steps 5 --warmup 2 > "$O/pmc_m.log" 2>&1
python tools/pmc_mfma.py "$O/pmc_m" > "$O/mfma_busy.txt" 2>&1
rm -rf "$O/pmc_m"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$O/pmc_sq" -- $B --steps 5 --warmup 2 > "$O/pmc_sq.log" 2>&1
python tools/pmc_sq.py "$O/pmc_sq" > "$O/sq_counters.jsonl" 2> "$O/sq.err"
rm -rf "$O/pmc_sq"
# the bench line takes roofline.traffic from the PMC profile of THIS build under profiles/ (matched by source digest)
cp "$O/pmc_traffic_per_launch.jsonl" profiles/${ROUND:-r06}_final_pmc_traffic_per_launch.jsonl
python bench.py --steps 30 --warmup 5 > "$O/bench_fp32.json" 2> "$O/bench_fp32.err"
python bench.py --steps 30 --warmup 5 --dtype bf16 --no-cpu-baseline --no-ops > "$O/bench_bf16.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --cfg scannet --scenes 2 --points 100000 --no-cpu-baseline --no-ops > "$O/bench_scannet.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --cfg scannet --scenes 2 --points 100000 --dtype bf16 --no-cpu-baseline --no-ops > "$O/bench_scannet_bf16.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --segmentor sam_image --no-cpu-baseline --no-ops > "$O/bench_sam.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --scenes 4 --points 80000 --no-cpu-baseline --no-ops > "$O/bench_4x80k.json" 2>/dev/null
# the same loop without the prefetcher: the forward builds its own geometry, pipelined with the level-0 prefix
AO_AMD_PREFETCH=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops > "$O/bench_fp32_no_prefetcher.json" 2>/dev/null
for f in fp32 fp32_no_prefetcher bf16 scannet scannet_bf16 sam 4x80k; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["ms_per_step"], d["value"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
