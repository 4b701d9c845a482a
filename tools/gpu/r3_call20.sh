#!/bin/bash
bash tools/gpu/ab_bench.sh r3c20 3
