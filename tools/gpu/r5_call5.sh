#!/bin/bash
# round 5, call 5: whole-tile timeline of the igemm8 kernel with the transposed epilogue (experiments build on the box)
export TMPDIR=/tmp
mkdir -p gpurun_out/r5c5
timeout 900 python tools/gpu/timeline8.py > gpurun_out/r5c5/timeline.txt 2>&1
tail -100 gpurun_out/r5c5/timeline.txt
