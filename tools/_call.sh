cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c12; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_unreduced_inputs.py -m gpu -q -x -p no:cacheprovider > $O/pytest_new.txt 2>&1; tail -3 $O/pytest_new.txt
timeout 900 python tools/table_sweep.py 20 > $O/table_sweep.txt 2> $O/table_sweep.err; cat $O/table_sweep.txt; tail -3 $O/table_sweep.err
