cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
uptime
bash tools/gpu/ab.sh gpurun_out/r03an/z AO_AMD_PREFETCH=thread AO_AMD_PREFETCH=1 6
uptime
