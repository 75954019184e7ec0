cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert |timed out|error" | head -15
echo "pytest wall $(( $(date +%s) - t0 )) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
