#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/bf16x3_stress.py 2>&1 | grep -v Warning | tail -30
