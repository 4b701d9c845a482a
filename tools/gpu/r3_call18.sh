#!/bin/bash
bash tools/gpu/trace_cli.sh r3c18
