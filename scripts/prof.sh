# usage: bash scripts/prof.sh <tag> <bench args...>  -- rocprofv3 kernel trace + stats of bench.py
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT.log 2>&1
cat $OUT/*/*kernel_stats.csv | cut -c1-200
grep '"value"' $OUT.log | cut -c1-1600
