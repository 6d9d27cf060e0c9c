for d in randn zeros relu small; do
DECNET_WINO_TILE=96 python tools/bench_wino_gemm.py --variant 1 --nt 6144 --data $d
done
for d in randn zeros; do
DECNET_WINO_GEMM=lds python tools/bench_wino_gemm.py --variant 1 --nt 6144 --data $d
DECNET_WINO_TILE=96 python tools/bench_wino_gemm.py --variant 1 --data $d
DECNET_WINO_TILE=96 python tools/bench_wino_gemm.py --variant 1 --data $d --iters 300
done
