# round 6: the driver's --gpus N command shapes once more over the loopback wire on one GPU (functional dry run)
mkdir -p gpurun_out/r6
export PYTHONPATH=$PWD ESR_WIRE_ONE_GPU=1 ESR_RCCL_LIB=$PWD/tests/wire/libesr_loopback_wire.so
run() {
  name=$1; n=$2; shift 2
  (timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29578 \
     bench.py --gpus $n --steps 20 --warmup 5 "$@" 2> gpurun_out/r6/dry_$name.err | tail -1) > gpurun_out/r6/dry_$name.json
  echo "== $name $(cut -c1-260 gpurun_out/r6/dry_$name.json)"; grep -v "amdgpu.ids\|socket.cpp\|^$" gpurun_out/r6/dry_$name.err | tail -3
}
run w2_default 2
run w4_default 4
run w8_default 8
run w2_triplet 2 --workload triplet --no-cpu-baseline
