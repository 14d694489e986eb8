mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_sq -o s -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > /tmp/pmc_sq.log 2>&1
cd $R && python tools/pmc_sq.py /tmp/pmc_sq 40 > gpurun_out/r04_pmc_sq.txt 2>&1 || tail -5 /tmp/pmc_sq.log > gpurun_out/r04_pmc_sq.txt
