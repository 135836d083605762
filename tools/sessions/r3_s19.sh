timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "two_launch or full_size" > gpurun_out/r3_s19.log 2>&1; tail -30 gpurun_out/r3_s19.log
