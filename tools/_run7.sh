python -m pytest tests/test_stage0_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0; do
python tools/bench_wino_gemm.py --variant $v --iters 100
DECNET_WINO_TILE=192 python tools/bench_wino_gemm.py --variant $v --iters 100
DECNET_WINO_XG=2 python tools/bench_wino_gemm.py --variant $v --iters 100
done
python tools/bench_wino_gemm.py --variant 1 --nt 6144 --iters 100
for a in winograd winograd4; do python tools/bench_conv3d.py --algo $a; done
DECNET_CONV_ALGO=winograd4 python bench.py --steps 10 --warmup 3 | tail -1 | cut -c1-1300
