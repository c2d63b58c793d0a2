#!/bin/bash
# round 5, GPU call I: the full GPU suite with every test starting from NaN-filled free blocks (SDFHIP_TEST_POISON=1): nothing may read
# memory it did not write (this round carved smaller inference workspaces and re-routed every no_grad render onto them)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
SDFHIP_TEST_POISON=1 timeout 1300 python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest_poison.log 2>&1
echo "pytest (poison) rc $?"; grep -E "passed|failed" $O/pytest_poison.log | tail -3; grep "^FAILED" $O/pytest_poison.log | head -30
