# usage: bash tools/gpu/bench_lines.sh OUT   -- the six bench lines of profiles/<round>_final_bench_*.json (tools/profile_round.sh's
# bench leg alone: for a change on the python side that leaves the library build, and with it the profiles, as they are)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; mkdir -p $O
python bench.py --steps 30 --warmup 5 > "$O/bench_fp32.json" 2> "$O/bench_fp32.err"
python bench.py --steps 30 --warmup 5 --dtype bf16 --no-cpu-baseline --no-ops > "$O/bench_bf16.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --cfg scannet --scenes 2 --points 100000 --no-cpu-baseline --no-ops > "$O/bench_scannet.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --cfg scannet --scenes 2 --points 100000 --dtype bf16 --no-cpu-baseline --no-ops > "$O/bench_scannet_bf16.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --segmentor sam_image --no-cpu-baseline --no-ops > "$O/bench_sam.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --scenes 4 --points 80000 --no-cpu-baseline --no-ops > "$O/bench_4x80k.json" 2>/dev/null
for f in fp32 bf16 scannet scannet_bf16 sam 4x80k; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d["host"]
print(sys.argv[2], round(d["ms_per_step"], 3), round(d["value"] / 1e6, 2), {k: round(v, 2) for k, v in h["step_ms"].items()},
      h["allocator_reserved_growth_MB"], d["roofline"]["kernel"] if d.get("roofline") else None, d["roofline"].get("traffic") if d.get("roofline") else None)
PY
done
