#!/bin/bash
# GPU box: the tests named on the command line
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest "$@" -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -25 gpurun_out/t.log; exit $rc
