for f in valu mfma; do
echo "== $f"
for args in "--cin 8 --cout 8 --h 180 --w 324 --batch 72" "--cin 8 --cout 8 --h 180 --w 324 --batch 8" "--cin 8 --cout 8 --h 180 --w 324 --batch 16" "--cin 8 --cout 1 --h 180 --w 324 --batch 8" "--cin 8 --cout 8 --h 60 --w 108 --batch 8" "--cin 8 --cout 8 --h 540 --w 972 --batch 8"; do DECNET_CONV2D_SMALL=$f python tools/bench_conv2d.py $args 2>&1 | grep conv; done
done
