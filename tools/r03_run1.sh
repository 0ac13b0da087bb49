#!/bin/bash
# round 3, GPU call 1: new geometry / file tests, C5 timing with the synthetic 1 B-code database,
# PMC evidence for the VLQ scan kernels and the IMI 2x14 coarse kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03a; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_geometry.py tests/test_index_io.py -x -q -m gpu -k "deep1b_own or reference_written" > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.txt
tail -3 $OUT/pytest_new.txt
export SYNTH=1 NLIST=65536 NEDGE=64 NB=1000000000 CHECK=4
timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/c5_synth_fp32.log 2>&1; tail -4 $OUT/c5_synth_fp32.log
FP16=1 timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/c5_synth_fp16.log 2>&1; tail -4 $OUT/c5_synth_fp16.log
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_trace -- python $GRAFT_REPO_ROOT/tools/time_vlq.py 2000 5 > $OUT/c5_trace.log 2>&1)
PMC_TIMEOUT=400 bash profiles/pmc_cmd.sh $OUT/pmc_c5_fp32 "line" tools/time_vlq.py 2000 3 || echo "pmc fp32 failed"
export FP16=1
PMC_TIMEOUT=400 bash profiles/pmc_cmd.sh $OUT/pmc_c5_fp16 "line" tools/time_vlq.py 2000 3 || echo "pmc fp16 failed"
unset FP16 SYNTH NLIST NEDGE NB CHECK
export NBITS=14 NB=20000000
PMC_TIMEOUT=400 bash profiles/pmc_cmd.sh $OUT/pmc_imi14 "vlq::" tools/time_imi.py || echo "pmc imi failed"
echo done
