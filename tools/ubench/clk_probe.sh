for d in randn zeros; do
  python tools/bench_wino_gemm.py --variant 2 --iters 40000 --data $d > /tmp/o_$d.txt 2>&1 &
  PID=$!
  sleep 4
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power\|fclk\|mclk" | head -6; sleep 0.7; done
  wait $PID
  tail -1 /tmp/o_$d.txt
done
