#!/bin/bash
# Runs GPU steps one after another on the gpurun box, each under its own time limit, logging to gpurun_out/<name>.log.
# A step that ends with anything but 0 (ok) or 1 (test / assertion failure) -- a fault, abort, segfault, time limit --
# ends the whole call: nothing more is started on the GPU after it.
# usage: source benchmarks/gpu_steps.sh; step <name> <seconds> <command...>
mkdir -p gpurun_out
step() {
  local name=$1 secs=$2
  shift 2
  echo "=== $name: $*"
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
  local rc=$?
  echo "=== $name rc=$rc"
  tail -n 3 "gpurun_out/$name.log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then
    tail -n 20 "gpurun_out/$name.err"
    exit $rc
  fi
}
