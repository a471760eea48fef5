# round 5: the config-size oracle trajectories (printed errors kept), the new regression tests, then the whole GPU suite
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_config_size_oracle.py tests/test_gpu_triplet_step.py::test_direct_plan_generation_is_32_bits_and_a_plan_feeds_repeated_steps tests/test_gpu_stl_loop.py::test_planned_batch_refuses_another_state -m gpu -q -s -p no:cacheprovider --durations=12 2>&1 | grep -v "^$" | tail -80) > gpurun_out/t_r5_parity.log 2>&1
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 2>&1 | grep -v "^$" | tail -60) > gpurun_out/t_all.log 2>&1
grep -E "passed|failed" gpurun_out/t_r5_parity.log gpurun_out/t_all.log
