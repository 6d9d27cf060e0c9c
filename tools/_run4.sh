for t in 0 96; do
echo "== reg kernel tile env $t, variant 1"
DECNET_WINO_TILE=$t python tools/bench_wino_gemm.py --variant 1
for a in 1 2 3 5; do DECNET_WINO_TILE=$t DECNET_HIP_LIB=$PWD/tools/ubench/libdecnet_wabl$a.so python tools/bench_wino_gemm.py --variant 1; done
done
DECNET_WINO_TILE=96 python tools/bench_wino_gemm.py --variant 1 --nt 6144
DECNET_WINO_TILE=96 DECNET_HIP_LIB=$PWD/tools/ubench/libdecnet_wabl5.so python tools/bench_wino_gemm.py --variant 1 --nt 6144
