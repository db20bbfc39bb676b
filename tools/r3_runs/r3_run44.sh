#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  echo -n "blocking: "; OZK_HOST_BLOCKING_WAITS=1 python tools/host_path.py 20 2>&1 | grep "double" | cut -c48-200
  echo -n "polling:  "; python tools/host_path.py 20 2>&1 | grep "double" | cut -c48-200
done
