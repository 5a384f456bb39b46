#!/bin/bash
# Round 6, twenty-sixth GPU call: the round's profile recipe on the tree as it stands (replaces the r06_* set of call 5).
bash scripts/profile_round.sh r06 2>&1 | tail -70
