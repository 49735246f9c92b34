#!/bin/bash
set -u
export TMPDIR=/tmp
for hot in 0 1; do
  VX_SERVICE_MIN=64 VX_TIMELINE=1 timeout 200 python3 profiles/timeline.py --format csvo --hot $hot 2>/dev/null | tail -n 1
done
