# usage: bash tools/gpu/final_round.sh [round]   -- the whole GPU suite, then every profiles/<round>_final_* artefact, in one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r06}
mkdir -p gpurun_out/${R}final
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/${R}final/gpu_tests.txt
bash tools/gpu/run_profile.sh $R
