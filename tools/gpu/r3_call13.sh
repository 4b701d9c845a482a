#!/bin/bash
out=gpurun_out/r3c13
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/gpu/timeline8.py --prebuilt > $out/timeline8.txt 2>&1; cat $out/timeline8.txt
