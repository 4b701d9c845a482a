#!/bin/bash
out=gpurun_out/r3c8
mkdir -p $out
timeout 600 python tools/gpu/fgfa_debug.py > $out/fgfa_debug.txt 2>&1; tail -40 $out/fgfa_debug.txt
