b() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-steady 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
echo "default: $(b) $(b)"
for q in 1 3 4; do echo "Q_PER_CU=$q: $(ESR_IB2H_Q_PER_CU=$q b)"; done
echo "Q=64: $(ESR_IB2H_Q=64 b)"
echo "PC=dma: $(ESR_IB2H_PC=dma b)"
