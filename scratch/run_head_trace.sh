PCTS="0 auto 12" bash scratch/head_sweep.sh 1600000 2>&1 | tee gpurun_out/r04_head_sweep_d.txt
PCTS="0 auto" bash scratch/head_sweep.sh 800000 3200000 2>&1 | tee -a gpurun_out/r04_head_sweep_d.txt
for wl in aadhaar_standin keyless_standin; do for p in 0 auto 0 auto; do echo "-- $wl head $p: $( ( [ $p = auto ] || export ICICLE_SNARK_HEAD_PCT=$p; LOOP_WORKLOAD=$wl python scratch/standin_loop.py 20 2>/dev/null | head -2 | tr '\n' ' ') )"; done; done 2>&1 | tee -a gpurun_out/r04_head_sweep_d.txt
ICICLE_SNARK_TRACE_HOST=1 python scratch/file_trace.py 2>&1 | grep "head .* of the witness" | head -12
