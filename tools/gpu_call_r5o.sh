#!/bin/bash
# round 5, GPU call O: the full GPU suite and smoke() on the tree as committed at the end of the round
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; mkdir -p $O
timeout 800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep "^FAILED" $O/pytest.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
