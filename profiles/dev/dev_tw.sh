#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
B=nudge_amd/compat/_build/replay_hip
cp gpurun_out/s_pile256.bin /tmp/s.bin 2>/dev/null || python3 -c "
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import callsite_util as CU, make_golden as G
CU.write_scene('/tmp/s.bin', G.build('pile256')[0])"
echo "--- plain"; $B /tmp/s.bin /tmp/o.bin 60; echo rc=$?
echo "--- two_worlds classic sort"; NH_SORT_CLASSIC=1 $B /tmp/s.bin /tmp/o.bin 60 two_worlds; echo rc=$?
echo "--- two_worlds default"; $B /tmp/s.bin /tmp/o.bin 60 two_worlds; echo rc=$?
if which rocgdb >/dev/null 2>&1; then rocgdb -batch -ex run -ex bt -ex "thread apply all bt 8" --args $B /tmp/s.bin /tmp/o.bin 60 two_worlds 2>&1 | tail -60; fi
