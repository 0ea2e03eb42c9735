#!/bin/bash
# usage on the GPU box: bash tools/g.sh <name> <timeout-seconds> <command...>
# runs one GPU step under its own time limit, output to gpurun_out/r02/<name>.log, prints the tail
mkdir -p gpurun_out/r02
name=$1; lim=$2; shift 2
timeout -k 10 "$lim" "$@" > gpurun_out/r02/$name.log 2> gpurun_out/r02/$name.err
rc=$?
tail -${TAILN:-15} gpurun_out/r02/$name.log
[ $rc -ne 0 ] && tail -15 gpurun_out/r02/$name.err
exit $rc
