python -m pytest tests/test_distributed_gpu.py tests/test_optimizer_gpu.py tests/test_graph_gpu.py -q 2>&1 | tail -6
