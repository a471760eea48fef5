# kernel stats of the GloVe C3 loop and the big-batch triplet step: r2 tree against the current tree, same box
export TMPDIR=/tmp
for tree in _r2tree .; do for cfg in "glove --steps 100 --warmup 10" "triplet --batch 262144 --steps 20 --warmup 3"; do
  set -- $cfg; name=$(echo $tree$1 | tr -d './_')
  rm -rf /tmp/ab_$name
  (cd $tree && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$name -o t -- python bench.py --workload $cfg --no-cpu-baseline --no-kernel-timing --no-secondary $( [ $tree = . ] && echo --no-steady ) > /tmp/ab_$name.log 2>&1)
  echo "== $tree $cfg: $(grep '^{' /tmp/ab_$name.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"
  python3 scripts/prof_stats.py /tmp/ab_$name 9 | cut -c1-150
done; done
