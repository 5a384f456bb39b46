#!/bin/bash
CL_BENCH_OVERSUBSCRIBE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --reps 2 --no-streaming 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('rank_affinity'), d['rank_ms_per_step'], d['rank_launch_us'], d['control_backend'])"
python -c "
import torch
p=torch.cuda.get_device_properties(0); print([a for a in dir(p) if 'pci' in a.lower()], getattr(p,'pci_bus_id',None), getattr(p,'pci_device_id',None), getattr(p,'pci_domain_id',None))"
ls /sys/bus/pci/devices | head -5; cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c | head
