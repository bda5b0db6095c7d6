#!/bin/bash
# gpurun helper: kernel trace of the fp32 step with absmax hints on / off
PROF_OUT=prof_h1 bash scratch/prof_cmd.sh --dtype fp32 --switch amax_hints=1 2>&1 | tail -42 | cut -c1-150
echo ===== hints off
PROF_OUT=prof_h0 bash scratch/prof_cmd.sh --dtype fp32 --switch amax_hints=0 2>&1 | tail -42 | cut -c1-150
