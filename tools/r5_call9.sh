cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c9; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
WSNARK_NTT_NP=2 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "ntt_vs_oracle or fft_golden or calc_h or ntt_extreme" > $O/pytest_np2.txt 2>&1; tail -3 $O/pytest_np2.txt
for NP in 0 2 0 2; do
  WSNARK_NTT_NP=$NP timeout 300 python tools/ntt_probe.py 2>&1 | grep "2^20" | sed "s/^/np=$NP /" | tee -a $O/ntt_probe.txt
done
for NP in 0 2 0 2; do
  WSNARK_NTT_NP=$NP timeout 600 python tools/sched_ab.py 3 "PROVE_ORDER=1" 2> $O/err.txt | sed "s/^/np=$NP /" | tee -a $O/sched_ab.txt
done
