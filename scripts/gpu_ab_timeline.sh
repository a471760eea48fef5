export TMPDIR=/tmp
for tree in _r2tree .; do
  name=$(echo tl$tree | tr -d './_'); rm -rf /tmp/$name
  (cd $tree && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/$name -o t -- python bench.py --workload glove --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary $( [ $tree = . ] && echo --no-steady ) > /tmp/$name.log 2>&1)
  echo "== $tree"
  python3 scripts/trace_gaps.py /tmp/$name glove_step 90 26 | grep -v "at::native" | cut -c1-110
done
