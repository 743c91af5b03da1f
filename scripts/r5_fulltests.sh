cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_full
timeout 3300 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r5_full/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5_full/pytest.log
tail -30 gpurun_out/r5_full/pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
