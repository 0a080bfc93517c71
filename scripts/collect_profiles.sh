#!/bin/bash
# copies what one `bash scripts/gpu_r4.sh bench prof pmc sq b4 prof4 sweep hires variants` session left under gpurun_out/ into profiles/r04_*
set -e
tail -1 gpurun_out/bench.log > profiles/r04_bench_line.json
cp "$(ls -t gpurun_out/prof/*/*kernel_stats.csv | head -1)" profiles/r04_kernel_stats.csv
cp "$(ls -t gpurun_out/prof4/*/*kernel_stats.csv | head -1)" profiles/r04_local_batch4_kernel_stats.csv
cp gpurun_out/launch_table.csv profiles/r04_launch_table.csv
cp gpurun_out/launch_table.txt profiles/r04_launch_table.txt
cp gpurun_out/launch_table_b4.txt profiles/r04_launch_table_b4.txt
cp gpurun_out/pmc_traffic.json profiles/r04_pmc_traffic.json
cp gpurun_out/pmc_traffic.txt profiles/r04_pmc_traffic.txt
cp gpurun_out/sq_counters.txt profiles/r04_sq_counters.txt
cp gpurun_out/parity_achieved.json profiles/r04_parity_achieved.json
(for f in v_odd v_even v_fp8 v_f32 v_c3rank v_c4rank b16 b8 b4 b4cycle b512 b1024; do echo "# $f"; tail -1 gpurun_out/$f.log; done) > profiles/r04_bench_variants.txt
