python -m pytest tests/test_stage0_gpu.py -m gpu -x -q -k "winograd_layer or golden" 2>&1 | tail -3
for v in 1 0; do
echo "== variant $v"
DECNET_WINO_GEMM=lds python tools/bench_wino_gemm.py --variant $v
DECNET_WINO_GEMM=lds DECNET_WINO_SWZ=0 python tools/bench_wino_gemm.py --variant $v
python tools/bench_wino_gemm.py --variant $v
DECNET_WINO_SWZ=0 python tools/bench_wino_gemm.py --variant $v
DECNET_WINO_TILE=96 python tools/bench_wino_gemm.py --variant $v
DECNET_WINO_XG=2 python tools/bench_wino_gemm.py --variant $v
DECNET_WINO_XG=4 python tools/bench_wino_gemm.py --variant $v
DECNET_WINO_XG=8 python tools/bench_wino_gemm.py --variant $v
DECNET_WINO_TILE=96 DECNET_WINO_XG=2 python tools/bench_wino_gemm.py --variant $v
done
for a in winograd winograd4; do python tools/bench_conv3d.py --algo $a; done
