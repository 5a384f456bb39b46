#!/bin/bash
# flakiness check: the whole GPU suite twice on the final build
set -u
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
