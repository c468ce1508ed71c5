#!/bin/bash
# one test selection: dev_one.sh <pytest args>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1200 python -m pytest "$@" -x -q -m gpu -s 2>&1 | tail -25
