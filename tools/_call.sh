cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c11; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1200 -p no:cacheprovider --durations=10 > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt
