# A/B of two builds of the library on one box: gpurun_ab_old.so (the previous build, copied there by hand) against the tree's
python -m pytest tests/test_gpu_retrieve.py tests/test_gpu_ivf.py -x -q -p no:cacheprovider 2>&1 | tail -2
cp esrecsys_amd/libesr_hip.so /tmp/new.so
for rep in 1 2; do for which in old new; do
  if [ $which = old ]; then cp gpurun_ab_old.so esrecsys_amd/libesr_hip.so; else cp /tmp/new.so esrecsys_amd/libesr_hip.so; fi
  for mode in f16x2 exact; do echo "== $which $mode: $(python scripts/retr_ktime.py $mode 2>/dev/null | grep -E "^op|score_gemm|topk_select" | tr '\n' ' ' | tr -s ' ')"; done
done; done
cp /tmp/new.so esrecsys_amd/libesr_hip.so
