#!/bin/bash
# Samples rocm-smi (shader clock, average socket power) twice a second while a command runs:   tests/probe/power_trace.sh <out.txt> <command...>
out=$1; shift
( while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|Socket" | tr '\n' ' '; echo; sleep 0.5; done ) > "$out" &
mon=$!
"$@"
rc=$?
kill $mon
exit $rc
