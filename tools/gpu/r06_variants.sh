# usage: bash tools/gpu/r06_variants.sh "python tools/bench_fwd_tile.py 100" name1 name2 ...  -- runs the command once per ao_amd/lib/libptv2_<name>.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="$1"; shift
cp ao_amd/lib/libptv2_hip.so /tmp/libptv2_keep.so
for t in "$@"; do
  echo "== $t"
  cp ao_amd/lib/libptv2_$t.so ao_amd/lib/libptv2_hip.so
  $CMD
done
cp /tmp/libptv2_keep.so ao_amd/lib/libptv2_hip.so
