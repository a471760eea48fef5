# DRY RUNS of the driver's N > 1 commands on ONE GPU over tests/wire's loopback wire (functional: the code path of an
# N-GPU run, not a scaling measurement): --gpus 4 / 8 in-batch, the overlapped loop, and BASELINE config 4 at FULL size
# (8 ranks x 12.5 M-row shards of two 100 M-row bf16 towers = 154 GB of the one GPU's 288 GB, bf16 gradient exchange)
mkdir -p gpurun_out
export PYTHONPATH=$PWD ESR_WIRE_ONE_GPU=1 ESR_RCCL_LIB=$PWD/tests/wire/libesr_loopback_wire.so
run() {  # name, nproc, extra env..., -- bench args
  name=$1; n=$2; shift 2
  (timeout 900 env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29578 \
     bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline $BARGS 2> gpurun_out/dry_$name.err | tail -1) > gpurun_out/dry_$name.json
  echo "== $name rc=$? $(cut -c1-330 gpurun_out/dry_$name.json)"; grep -v "amdgpu.ids\|socket.cpp\|^$" gpurun_out/dry_$name.err | tail -4
}
BARGS="--workload inbatch" run w4_inbatch 4 A=1
BARGS="--workload inbatch" run w8_inbatch 8 A=1
BARGS="--workload triplet" run w8_triplet 8 A=1
BARGS="--workload inbatch" run w2_inbatch_overlap 2 ESR_SHARDED_OVERLAP=1
BARGS="--workload triplet" run w4_triplet_overlap 4 ESR_SHARDED_OVERLAP=1
BARGS="--workload glove" run w2_glove_overlap 2 ESR_SHARDED_OVERLAP=1
BARGS="--workload inbatch --rows 100000000 --table-dtype bf16" run w8_config4_full 8 ESR_SHARDED_GRAD_DTYPE=bf16
BARGS="--workload retrieve --steps 2 --warmup 1" run w8_config5_retrieve 8 A=1
