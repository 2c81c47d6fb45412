cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ai; mkdir -p $O
bash tools/gpu/ab.sh $O/a AO_AMD_GEMM_WIDE=512 X=1 3
bash tools/gpu/ab.sh $O/b AO_AMD_GEMM_WIDE=256 AO_AMD_GEMM_WIDE=1024 3
