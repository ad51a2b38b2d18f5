cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c10; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "msm_golden or msm_vs_oracle or all_same_point or grouping_variants or proofs_golden or adversarial or prove_small or degenerate or toxic_waste or msm_2p20" > $O/pytest_subset.txt 2>&1
tail -3 $O/pytest_subset.txt
C="COMBINE_FUSED=0;COMBINE_FUSED=1;COMBINE_FUSED=1,PROVE_CALCH_FIRST=0"
timeout 600 python tools/sched_ab.py 6 "$C" 2> $O/err.txt | tee $O/sched_ab.jsonl
