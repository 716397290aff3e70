#!/bin/bash
# Run GPU steps one after the other, each under its own timeout, logging to gpurun_out/<name>.log; a step that times out
# (exit 124 / 137) ends the whole call (no further GPU step after a hang), a step that merely fails does not.
#   tools/gpu_steps.sh name1 secs1 'cmd1' name2 secs2 'cmd2' ...
mkdir -p gpurun_out
while [ $# -ge 3 ]; do
  name=$1; secs=$2; cmd=$3; shift 3
  echo "=== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc"; tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out: stopping"; exit $rc; fi
done
exit 0
