#!/bin/bash
# round 4, call 23: full GPU suite on the build that is being committed, smoke(), then the profile collection (tools/collect_profiles.sh r04)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04c23_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04c23_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
