#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "
import __graft_entry__ as g, time
t=time.time(); g.smoke(); print('smoke ok %.1fs' % (time.time()-t))" 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-1500
