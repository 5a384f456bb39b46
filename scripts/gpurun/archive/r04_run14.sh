#!/bin/bash
timeout 300 python scripts/short_region_launch_ab.py 2>&1 | grep -v amdgpu.ids
