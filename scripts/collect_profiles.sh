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
if [ -f gpurun_out/h2d_on.json ]; then      # the PCIe-inclusive A/B (bench.py --h2d), when that call was made
  (echo; echo "# PCIe-inclusive variant (bench.py --h2d: the three views of every iteration copied from pinned host memory one iteration ahead on a side stream,"
   echo "# worker.SyntheticHostTriples = the reference's three .to(device) of worker.py:141-143), A/B on ONE box; never the headline value"
   for f in off on off_b4 on_b4; do echo -n "h2d_$f: "; tail -1 gpurun_out/h2d_$f.json; done) >> profiles/r04_bench_variants.txt
fi
if [ -f gpurun_out/cpucycle.log ]; then      # the CPU oracle's full warmed 8-iteration cycle (bench.py --cpu-baseline-cycle), when that step ran
  (echo; echo "# cpu_baseline as SURVEY 8(d) words it: one warm-up cycle, then one timed 8-iteration cycle of the CPU oracle at 256x256, batch 4 (bench.py --cpu-baseline-cycle)"
   tail -1 gpurun_out/cpucycle.log) >> profiles/r04_bench_variants.txt
fi
