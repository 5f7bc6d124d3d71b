#!/bin/bash
python -m pytest tests/test_gpu_comm.py -x -q 2>&1 | tail -15
echo "== full suite"
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15
echo "exit: ${PIPESTATUS[0]}"
