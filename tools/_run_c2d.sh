python -m pytest tests/test_conv2d_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -4
python tools/e2e_layers.py 8 2>&1 | head -30
DECNET_CONV2D=torch python tools/e2e_layers.py 8 2>&1 | head -3
