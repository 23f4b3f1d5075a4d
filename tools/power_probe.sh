#!/bin/bash
# usage: tools/power_probe.sh OUT -- command ...   : samples socket power / clocks (rocm-smi) every ~0.25 s while the command runs
out=$1; shift; shift
( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Power|sclk|mclk|Temp' | tr -s ' ' | tr '\n' ';')"; sleep 0.25; done ) > $out 2>&1 &
probe=$!
"$@"
rc=$?
kill $probe 2>/dev/null
exit $rc
