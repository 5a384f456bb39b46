#!/bin/bash
timeout 600 python scripts/c4_defer_sweep.py 2>&1 | grep -v amdgpu.ids
