cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/pytest_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> gpurun_out/pytest_final.txt
cat gpurun_out/pytest_final.txt
bash tools/gpu_r4_final.sh 2>&1 | grep -v "^+" | tail -40
