#!/bin/bash
# usage (on the GPU box, from the repo root): tools/r3_profiles.sh  -> gpurun_out/r03_*  (copy into profiles/)
# Everything the headline line cites, from ONE build: HBM traffic (two --pmc passes), the bench line that embeds it,
# kernel stats, the per-launch table, the clock / matrix-pipe-busy table and the in-kernel timelines.
O=gpurun_out
tools/traffic.sh r3 > /dev/null 2>&1 && cp $O/traffic_r3.json profiles/r03_traffic.json && cp $O/traffic_r3.json $O/r03_traffic.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r03_bench.json 2> $O/r03_bench.err
tools/prof_cfg.sh r3cfg2 cfg2 5 > $O/r03_cfg2_kernel_table.txt 2>&1
cp $(ls $O/ks_r3cfg2/*/*kernel_stats.csv | head -1) $O/r03_bench_kernel_stats.csv
timeout -k 10 300 python tools/launch_table.py > $O/r03_cfg2_launch_table.txt 2>&1
tools/clock_table.sh r3 > /dev/null 2>&1; cp $O/clock_r3.txt $O/r03_clock_table.txt
tail -c 700 $O/r03_bench.json; tail -5 $O/r03_cfg2_launch_table.txt
