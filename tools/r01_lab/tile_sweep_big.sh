for shape in "448 56 56 256 256 3 1 1" "448 28 28 256 256 3 1 1" "448 14 14 256 256 3 1 1" "448 7 7 512 512 3 1 1" "448 14 14 1024 256 1 1 0" "448 14 14 256 1024 1 1 0" "448 7 7 512 2048 1 1 0"; do
for t in ${TILES:-3 12 13 14 9}; do MCG_TILE=$t python tools/conv_bench.py $shape 30 2>&1 | tail -1; done; done
