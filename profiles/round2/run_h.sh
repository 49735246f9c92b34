#!/bin/bash
set -u
mkdir -p gpurun_out/r2h; rm -f gpurun_out/r2h/*
timeout 600 python profiles/sweep.py --format esvo --depth 12 --configs "h=0,f=2" "h=2,f=2" "h=4,f=2" "h=6,f=2" "h=7,f=2" --rounds 4 --steps 20 > gpurun_out/r2h/sweep_hot_esvo.txt 2>&1
tail -n 6 gpurun_out/r2h/sweep_hot_esvo.txt
