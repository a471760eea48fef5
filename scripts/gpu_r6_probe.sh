# round 6 probes of inbatch2h_pct_kernel: prebuilt scripts/libib2h_<variant>.so (scripts/build_ib2h_variant.sh, in the
# container), per-kernel averages over 110 calls, the variants in turn, IB2H_ROUNDS times
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for r in $(seq ${IB2H_ROUNDS:-1}); do
for v in ${IB2H_VARIANTS:-BASE}; do
  rm -rf gpurun_out/prof/pr
  IB2H_LIB=libib2h_$v.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/pr -o t -- python scripts/ib2h_probe.py 2>&1 | grep "op "
  python scripts/prof_stats.py gpurun_out/prof/pr | grep -E "${IB2H_GREP:-2h|merge}" | cut -c1-44,100-140
done
done
