python -m pytest tests/test_gpu_env_api.py -x -q -k "test_batched_info_matches_the_host_classes" 2>&1 | tail -40
