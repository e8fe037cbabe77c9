for w in 2 3 4 6; do echo "== MVG_AB_WG=$w"; MVG_AB_WG=$w python scripts/conv_bench.py 50 128 4 10 split 2>&1 | grep -E "^wgrad|^all"; done > gpurun_out/r03n_ab.txt 2>&1
