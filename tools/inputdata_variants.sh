for v in "A=1" "DECNET_CONV_ALGO=direct" "DECNET_CONV_ALGO=winograd" "DECNET_CONV2D=torch" "DECNET_FOLD_BN=0" "DECNET_CONV2D=torch DECNET_FOLD_BN=0 DECNET_CONV_ALGO=direct" "DECNET_FEAT_BATCH=0"; do
  echo "=== $v"; env $v python -m pytest tests/test_inputdata_gpu.py -q -s -m gpu -k "Sceneflow-0006-init17 or KITTI-000009_10-fill" 2>&1 | grep "final:\|png:\|passed\|failed"
done
