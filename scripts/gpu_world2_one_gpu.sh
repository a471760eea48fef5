# world 2 on ONE GPU: the world-2 parity tests over the loopback wire / gloo, then the driver's --gpus 2 command as a dry run
mkdir -p gpurun_out
export PYTHONPATH=$PWD
(timeout 900 python -m pytest tests/test_gpu_rccl_world2.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -40) > gpurun_out/t_world2.log 2>&1
tail -25 gpurun_out/t_world2.log
export ESR_WIRE_ONE_GPU=1 ESR_RCCL_LIB=$PWD/tests/wire/libesr_loopback_wire.so
for wl in inbatch triplet glove; do
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
     bench.py --gpus 2 --steps 20 --warmup 5 --workload $wl 2> gpurun_out/dry_w2_$wl.err | tail -1) > gpurun_out/dry_w2_$wl.json
  echo "== $wl rc=$?"; cut -c1-900 gpurun_out/dry_w2_$wl.json; tail -5 gpurun_out/dry_w2_$wl.err | grep -v amdgpu.ids
done
