cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
bash tools/gpu/ab.sh $O/t512 AO_AMD_WG_TARGET=512 AO_AMD_WG_TARGET=768 2
bash tools/gpu/ab.sh $O/t1024 AO_AMD_WG_TARGET=1024 AO_AMD_WG_TARGET=1536 2
bash tools/gpu/ab.sh $O/t2304 AO_AMD_WG_TARGET=2304 AO_AMD_WG_TARGET=384 2
