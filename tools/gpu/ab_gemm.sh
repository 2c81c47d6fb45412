cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ag; mkdir -p $O
bash tools/gpu/ab.sh $O/a AO_AMD_GEMM_MID=512 AO_AMD_GEMM_MID=0 4
bash tools/gpu/ab.sh $O/b AO_AMD_GEMM_MID=256 AO_AMD_GEMM_MID=0 4
bash tools/gpu/ab.sh $O/c AO_AMD_GEMM_MID=128 AO_AMD_GEMM_MID=512 3
