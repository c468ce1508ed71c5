#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pile; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/profiles/dev/dev_pile_tl.py 2>/dev/null
rm -rf /tmp/tlp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlp -- python $R/profiles/dev/dev_pile_tl.py > $O/run.log 2>&1
python $R/profiles/export_timeline.py /tmp/tlp $O/pile_step_timeline.log
