#!/bin/bash
# round-4 GPU call: steps named on the command line; a KILLED step stops the chain
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a gpurun_out/summary.log
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/summary.log
  tail -${TAILN:-6} "gpurun_out/$name.log" | tee -a gpurun_out/summary.log
  if [ $rc -ge 124 ]; then echo "killed: stopping" | tee -a gpurun_out/summary.log; exit $rc; fi
}
: > gpurun_out/summary.log
for s in "$@"; do
  case $s in
    kernels) step kernels 600 python -m pytest tests/test_kernels_gpu.py tests/test_data.py -m gpu -q --timeout 300 ;;
    parity)  step parity 1100 python -m pytest tests/test_parity_gpu.py tests/test_ddp_gpu.py -m gpu -q --timeout 900 ;;
    alltests) step alltests 1100 python -m pytest tests -m gpu -q --timeout 900 ;;
    smoke)   step smoke 300 python __graft_entry__.py --smoke ;;
    bench)   step bench 900 python bench.py --steps 10 --warmup 3 --launch-table gpurun_out/launch_table.csv
             python scripts/launch_table.py gpurun_out/launch_table.csv > gpurun_out/launch_table.txt 2>&1 ;;
    benchq)  step benchq 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cycle --launch-table gpurun_out/launch_table.csv
             python scripts/launch_table.py gpurun_out/launch_table.csv > gpurun_out/launch_table.txt 2>&1 ;;
    prof)    step prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-cycle ;;
    pmc)     step pmc_fetch 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-cycle
             step pmc_write 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-cycle
             step pmc_table 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cycle --launch-table gpurun_out/pmc_launch_table.csv
             python3 scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic.json gpurun_out/pmc_launch_table.csv 2 2>&1 | tee gpurun_out/pmc_traffic.txt | tee -a gpurun_out/summary.log ;;
    sq)      step pmc_sq 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-cycle
             python3 scripts/pmc_sq.py gpurun_out/pmc_sq > gpurun_out/sq_counters.txt 2>&1 ;;
    prof4)   step prof4 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 bench.py --steps 3 --warmup 1 --batch 4 --no-cpu-baseline --no-roofline --no-cycle ;;
    sweep)   for b in 16 8 4; do step b$b 400 python bench.py --steps 10 --warmup 3 --batch $b --no-cpu-baseline --no-cycle --no-roofline; done ;;
    b4)      step b4 400 python bench.py --steps 10 --warmup 3 --batch 4 --no-cpu-baseline --no-cycle --launch-table gpurun_out/launch_table_b4.csv
             python scripts/launch_table.py gpurun_out/launch_table_b4.csv > gpurun_out/launch_table_b4.txt 2>&1
             step b4cycle 400 python bench.py --steps 8 --warmup 8 --batch 4 --epoch-type cycle --no-cpu-baseline --no-roofline ;;
    variants) step v_odd 400 python bench.py --steps 10 --warmup 3 --epoch-type odd --no-cpu-baseline --no-roofline
             step v_even 400 python bench.py --steps 6 --warmup 2 --epoch-type even --no-cpu-baseline --no-roofline
             step v_fp8 400 python bench.py --steps 10 --warmup 3 --dtype fp8 --no-cpu-baseline --no-cycle --no-roofline
             step v_f32 600 python bench.py --steps 3 --warmup 1 --dtype f32 --no-cpu-baseline --no-cycle --no-roofline
             step v_c3rank 400 python bench.py --steps 5 --warmup 2 --res 512 --batch 8 --no-cpu-baseline --no-cycle --no-roofline
             step v_c4rank 400 python bench.py --steps 5 --warmup 2 --res 1024 --batch 4 --freezeD-layer 5 --no-cpu-baseline --no-cycle --no-roofline ;;
    hires)   step b512 400 python bench.py --steps 5 --warmup 2 --res 512 --no-cpu-baseline --no-cycle
             step b1024 400 python bench.py --steps 5 --warmup 2 --res 1024 --freezeD-layer 5 --no-cpu-baseline --no-cycle ;;
    cpucycle) step cpucycle 700 python bench.py --steps 5 --warmup 2 --no-cycle --no-roofline --cpu-baseline-cycle ;;
    h2d)     for v in off on; do
               f=""; [ $v = on ] && f="--h2d"
               step h2d_$v 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline $f; tail -1 gpurun_out/h2d_$v.log > gpurun_out/h2d_$v.json
               step h2d_${v}_b4 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-cycle --batch 4 $f; tail -1 gpurun_out/h2d_${v}_b4.log > gpurun_out/h2d_${v}_b4.json
             done ;;
    *::*)    step "${s%%::*}" 900 bash -c "${s#*::}" ;;
    *)       step custom 900 bash -c "$s" ;;
  esac
done
