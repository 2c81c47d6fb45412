# usage: bash tools/gpu/r05_ab.sh OUT "test files" reps name1 name2 ...  -- the named parity tests on ao_amd/lib/libptv2_<name1>.so, then alternating
# bench runs of the named library builds (tools/gpu/ab_multi.sh); the library named first is left in place
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; T="$2"; R=$3; shift 3; mkdir -p $O
cp ao_amd/lib/libptv2_$1.so ao_amd/lib/libptv2_hip.so
if [ -n "$T" ]; then
  timeout 1500 python -m pytest $T -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
fi
bash tools/gpu/ab_multi.sh $O $R "$@" 2>&1 | tee $O/ab.txt
