DECNET_WINO_TILE=96 DECNET_HIP_LIB=$PWD/tools/ubench/libdecnet_wabl6.so python tools/bench_wino_gemm.py --variant 1 --nt 6144 --iters 100
DECNET_HIP_LIB=$PWD/tools/ubench/libdecnet_wabl6.so python tools/bench_wino_gemm.py --variant 1 --nt 6144 --iters 100
DECNET_WINO_TILE=96 python tools/bench_wino_gemm.py --variant 1 --nt 6144 --iters 100
python tools/bench_wino_gemm.py --variant 1 --nt 6144 --iters 100
