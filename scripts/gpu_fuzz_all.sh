# every fuzzer once with the seed given (SEED=n bash scripts/gpu_fuzz_all.sh): a few minutes on one GPU
S=${SEED:-101}
for f in "fuzz_ops.py 80" "fuzz_loops.py 45" "fuzz_steps.py 60" "fuzz_routing.py 100" "fuzz_retrieve.py 30" "fuzz_towers.py 60" "fuzz_optim.py 60" "fuzz_bf16_steps.py 50" "fuzz_ivf.py 20" "inbatch_stress.py 100"; do
  set -- $f
  echo "$1: $(SEED=$S CASES=$2 timeout 900 python scripts/$1 2>&1 | grep -E "MISMATCH|BEYOND|cases|Error|error|fault" | tail -3)"
done
echo "fuzz_sharded_w1.py: $(SEED=$S CASES=16 timeout 900 python scripts/fuzz_sharded_w1.py 2>&1 | grep -E "MISMATCH|cases|Error|fault" | tail -3)"
echo "fuzz_sharded_gloo.py WIRE=1 (world 2-4 on this GPU, real kernels, loopback wire): $(WIRE=1 SEED=$S CASES=14 PYTHONPATH=$PWD timeout 900 python scripts/fuzz_sharded_gloo.py 2>&1 | grep -E "MISMATCH|cases|Error|fault" | tail -3)"
