#!/bin/bash
# One GPU-box call: parity tests, smoke, a short bench.  A step that is KILLED (timeout) stops the chain; an ordinary
# test failure does not (its log is what we want back).
set -u
mkdir -p gpurun_out
step() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a gpurun_out/summary.log
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/summary.log
  tail -5 "gpurun_out/$name.log" | tee -a gpurun_out/summary.log
  if [ $rc -ge 124 ]; then echo "killed: stopping" | tee -a gpurun_out/summary.log; exit $rc; fi
}
: > gpurun_out/summary.log
for s in "$@"; do
  case $s in
    kernels) step kernels 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 ;;
    parity)  step parity 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 600 -x ;;
    smoke)   step smoke 300 python __graft_entry__.py --smoke ;;
    bench)   step bench 600 python bench.py --steps 5 --warmup 2 ;;
    benchq)  step benchq 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ;;
    prof)    export TMPDIR=/tmp
             step prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline ;;
    pmc)     export TMPDIR=/tmp
             step pmc_fetch 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
             step pmc_write 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
             python3 scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic.json | tee -a gpurun_out/summary.log ;;
    prof4)   export TMPDIR=/tmp
             step prof4 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 bench.py --steps 2 --warmup 1 --batch 4 --no-cpu-baseline --no-roofline ;;
  esac
done
