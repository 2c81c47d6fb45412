# one call on the GPU box: every profiles/<round>_final_* artefact of the current build.  usage: bash tools/gpu/run_profile.sh [round]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r06}
ROUND=$R bash tools/profile_round.sh gpurun_out/${R}prof
cp profiles/${R}_final_pmc_traffic_per_launch.jsonl gpurun_out/${R}prof/${R}_final_pmc_traffic_per_launch.jsonl
