#!/bin/bash
# round 4 visit W2: priority of the serial coder wavefronts (s_setprio) at 1000 chains -- is the bulk work held up by them?
export TMPDIR=/tmp
run() { BITSWAP_HIP_LIB=$1 python bench.py --no-extra --no-cpu-baseline --no-roofline --no-timeline --steps 8 --warmup 2 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['value'], d['lossless'])"; }
for p in 0 1; do BITSWAP_HIPCC_EXTRA="-DBS_SERIAL_PRIO=$p" BITSWAP_HIP_LIB=/tmp/lib_prio$p.so python -m bitswap_amd.build > /tmp/build$p.log 2>&1; tail -1 /tmp/build$p.log | cut -c1-60; done
run "" "prio 3 (default)"; run /tmp/lib_prio0.so "prio 0"; run /tmp/lib_prio1.so "prio 1"; run "" "prio 3 (default)"; run /tmp/lib_prio0.so "prio 0"
