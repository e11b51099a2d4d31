#!/bin/bash
# usage (GPU box): bash scripts/cli_e2e.sh -- wall time of the find_mems CLI, text in / text out, 1 M reads
for wl in x synth; do python3 -u scripts/cli_e2e.py $wl 4000000; done
