#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -k "fft or qap or witness or groth or jni" 2>&1 | tail -3
