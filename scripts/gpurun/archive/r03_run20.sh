#!/bin/bash
set -u
mkdir -p gpurun_out/r03_run20
timeout 600 python scripts/two_stream_probe.py > gpurun_out/r03_run20/two_stream_probe.log 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/r03_run20/two_stream_probe.log
