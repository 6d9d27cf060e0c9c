for v in 1 0; do
python tools/bench_wino_gemm.py --variant $v
for a in 1 2 3 5; do DECNET_HIP_LIB=$PWD/tools/ubench/libdecnet_wabl$a.so python tools/bench_wino_gemm.py --variant $v; done
done
# M-block quantisation: exact multiples of 192
python tools/bench_wino_gemm.py --variant 1 --nt 1536
python tools/bench_wino_gemm.py --variant 1 --nt 1344
python tools/bench_wino_gemm.py --variant 1 --nt 3072
python tools/bench_wino_gemm.py --variant 1 --nt 6144
