mkdir -p gpurun_out
python -m pytest tests/test_stage0_gpu.py -m gpu -x -q > gpurun_out/t_stage0.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/t_stage0.log
for a in direct winograd winograd4; do python tools/bench_conv3d.py --algo $a; done
for xg in 1 2 3 4 6 8; do echo xg=$xg; DECNET_WINO_XG=$xg python tools/bench_conv3d.py --algo winograd4; done
DECNET_WINO_TILE=96 python tools/bench_conv3d.py --algo winograd4
DECNET_CONV_ALGO=winograd4 python bench.py --steps 10 --warmup 3 | tail -1 > gpurun_out/bench_w4.json; cat gpurun_out/bench_w4.json | cut -c1-1500
