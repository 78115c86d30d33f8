python -m pytest tests/test_smp_gpu.py tests/test_physics_gpu.py tests/test_level_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/kernel_ms.sh 32 10
